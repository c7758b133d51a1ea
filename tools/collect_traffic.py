#!/usr/bin/env python3
"""Measure HBM-side traffic of the fused DCT kernels with rocprofv3 PMC counters (run on the GPU box).

Separate --pmc passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2); both
counters are in KiB.  The guide's gfx950 note (FETCH_SIZE reports half of the bytes of a 16 B/lane streaming
read) is calibrated for 16-byte accesses; these kernels issue 8 B/lane, so the factor is CALIBRATED here on
a kernel with the same access pattern and a known byte count: k_poly_f64<MODE 0> (the standalone FP64
forward NTT, `src[r * TP + tid]` u64 loads like k_dct_rows) over a buffer far larger than the Infinity
Cache, which must read exactly its input and write exactly its output.

`collect_traffic.py ctct` measures the ct x ct launch set instead (a batch of 2x2 products at P8192, every launch of
fhe_multiply summed) and writes profiles/pmc_traffic_ctct.json, tied to behz.hip + ntt_core.h by hash; tools/bench_ops.py
prints it beside its multiply line.

Writes profiles/pmc_traffic.json (tracked; read by bench.py) with the kernel names it saw and the hash of the
kernel sources it measured: bench.py prints `traffic: null` when the running sources differ."""
import collections
import csv
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = 256
CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--blocks", str(BLOCKS), "--cpu-blocks", "0", "--no-verify"]
CAL_CTS = 8192        # 8192 cts x 192 KiB = 1.5 GiB in, 1.5 GiB out
CAL = [sys.executable, "-c",
       "import sys; sys.path.insert(0, %r); import torch, fhip_amd as fhe; ctx = fhe.SEALContext.preset('P4096'); ev = fhe.Evaluator(ctx); "
       "a = ctx.random_ct(%d, seed=1); o = torch.empty_like(a); [ev.ntt_forward(a, out=o) for _ in range(3)]; torch.cuda.synchronize()" % (ROOT, CAL_CTS)]


def kernel_source_hash():
    """identifies the sources a PMC record belongs to (git is not available on the GPU box): the translation unit of the
    two measured kernels (dct_fused.hip and every header it includes) plus fhe_hip.hip, which holds their launch plan
    and wave size -- a change to another kernel file (behz.hip, dct_u64.hip) does not invalidate the record"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "csrc")
    files = [os.path.join(d, n) for n in ("dct_fused.hip", "fhe_hip.hip", "fp64_core.h", "internal.h", "modarith.h", "ntt_core.h", "host_math.h")]
    files.append(os.path.join(ROOT, "include", "fhe_hip.h"))
    for p in files:
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def run_pass(counters, tag, cmd, match):
    out = os.path.join(ROOT, "gpurun_out", "traffic_" + tag)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--pmc", *counters, "--"] + cmd,
                   check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    names = {}
    with open(os.path.join(out, "p_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            for m in match:
                if m in k:
                    acc[m][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    names[m] = k
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, names


# ---- ct x ct (csrc/behz.hip): every launch of a batch of 2x2 products at P8192 ------------------------------------------
MUL_BATCH, MUL_REPS = 256, 3
MUL = [sys.executable, "-c",
       "import sys; sys.path.insert(0, %r); import torch, fhip_amd as fhe; ctx = fhe.SEALContext.preset('P8192'); ev = fhe.Evaluator(ctx); "
       "a, b = ctx.random_ct(%d, seed=1), ctx.random_ct(%d, seed=2); [ev.multiply(a, b) for _ in range(%d)]; torch.cuda.synchronize()"
       % (ROOT, MUL_BATCH, MUL_BATCH, MUL_REPS)]
MUL_KERNELS = ["k_behz_prepare_pm", "k_behz_to_bsk", "k_ntt_fwd", "k_behz_tensor_intt", "k_behz_floor_back"]


def ctct_source_hash():
    """the sources the ct x ct record belongs to: behz.hip and the headers its kernels are built from, plus fhe_hip.hip (the
    transform kernels and their launch plan)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "csrc")
    for n in ("behz.hip", "fhe_hip.hip", "internal.h", "modarith.h", "ntt_core.h", "host_math.h"):
        h.update(n.encode())
        h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def run_pass_sum(counters, tag, cmd, match):
    """like run_pass, but SUMS the counter over every dispatch of a kernel family (a product is several launches)"""
    out = os.path.join(ROOT, "gpurun_out", "traffic_" + tag)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--pmc", *counters, "--"] + cmd,
                   check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tot, calls = collections.defaultdict(float), collections.defaultdict(int)
    with open(os.path.join(out, "p_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            for m in match:
                if m in row["Kernel_Name"]:
                    tot[m] += float(row["Counter_Value"])
                    calls[m] += 1
                    break
    return tot, calls


def ctct(f_read, f_write):
    k, n = 4, 8192
    alg = (2 + 2 + 3) * k * n * 8                                  # read two ct(2), write one ct(3): 1,835,008 B
    rd, calls = run_pass_sum(["FETCH_SIZE"], "mul_fetch", MUL, MUL_KERNELS)
    wr, _ = run_pass_sum(["WRITE_SIZE"], "mul_write", MUL, MUL_KERNELS)
    products = MUL_BATCH * MUL_REPS
    per_kernel, total = {}, 0.0
    for m in MUL_KERNELS:
        if not calls.get(m):
            continue
        r, w = rd[m] * 1024 * f_read / products, wr[m] * 1024 * f_write / products
        per_kernel[m] = {"read_bytes_per_product": r, "write_bytes_per_product": w, "launches_per_product_batch": calls[m] / MUL_REPS}
        total += r + w
    res = {"hbm_bytes_per_product": total, "algorithmic_bytes_per_product": alg, "ratio_to_algorithmic": total / alg,
           "residue_polynomials_moved_per_product": total / (n * 8), "per_kernel": per_kernel, "products_per_dispatch": MUL_BATCH,
           "kernel_source_hash": ctct_source_hash(), "read_factor": f_read, "write_factor": f_write,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), summed over every launch of %d x fhe_multiply of %d 2x2 "
                     "products at P8192 (n = 8192, four 54/55-bit moduli), per product; factors as calibrated for pmc_traffic.json; "
                     "tools/collect_traffic.py ctct" % (MUL_REPS, MUL_BATCH)}
    for d in ("profiles", "gpurun_out"):
        os.makedirs(os.path.join(ROOT, d), exist_ok=True)
        with open(os.path.join(ROOT, d, "pmc_traffic_ctct.json"), "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res))


# ---- the circuits (bench_circuits.py): every launch of the library in one process, per output pixel / per run -------------
CIRCUITS = {
    "resize": ["resize", "--max-pixels", "1024"],
    "resize_shared": ["resize", "--shared", "--max-pixels", "1024"],
    "decode": ["decode"],
    # the relinearised mode (include/fhe_circuits.h fhe_circuits_create_relin), decomposition bit count 30
    "resize_relin30": ["resize", "--relin", "30", "--max-pixels", "1024"],
    "resize_shared_relin30": ["resize", "--shared", "--relin", "30", "--max-pixels", "1024"],
    "decode_relin30": ["decode", "--relin", "30"],
}
NOT_THE_CIRCUIT = ("k_fill_random", "k_digest", "at::native", "rocclr", "k_make_shoup")      # inputs, digests, torch's own kernels, table set-up


def all_source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def run_pass_all(counter, tag, cmd):
    out = os.path.join(ROOT, "gpurun_out", "traffic_" + tag)
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--pmc", counter, "--"] + cmd,
                       check=True, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    tot = collections.defaultdict(float)
    with open(os.path.join(out, "p_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if any(x in k for x in NOT_THE_CIRCUIT):
                continue
            tot[k.split("(")[0].split("<")[0].split("::")[-1].replace("void ", "")] += float(row["Counter_Value"])
    return tot, line


def circuits(f_read, f_write):
    res = {"kernel_source_hash": all_source_hash(), "read_factor": f_read, "write_factor": f_write,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench_circuits.py at P8192, every launch of the library "
                     "summed (inputs, digests and torch's own kernels left out), divided by the job executions of the process and the units of a job; "
                     "factors as calibrated for pmc_traffic.json; tools/collect_traffic.py circuits"}
    for name, argv in CIRCUITS.items():
        cmd = [sys.executable, os.path.join(ROOT, "bench_circuits.py")] + argv
        rd, line = run_pass_all("FETCH_SIZE", "circ_%s_fetch" % name, cmd)
        wr, _ = run_pass_all("WRITE_SIZE", "circ_%s_write" % name, cmd)
        jobs = line["job_executions"]
        units = line["units_per_job"] if not name.startswith("decode") else 1           # per output pixel; per run
        per_kernel = {k: {"read_bytes_per_unit": rd[k] * 1024 * f_read / jobs / units, "write_bytes_per_unit": wr.get(k, 0.0) * 1024 * f_write / jobs / units}
                      for k in sorted(rd)}
        total = sum(v["read_bytes_per_unit"] + v["write_bytes_per_unit"] for v in per_kernel.values())
        alg = line["roofline"]["algorithmic_bytes_per_launch"] / units
        res[name] = {"unit": "output pixel" if not name.startswith("decode") else "run", "hbm_bytes_per_unit": total, "algorithmic_bytes_per_unit": alg,
                     "ratio_to_algorithmic": total / alg, "units_measured": line["units_per_job"], "job_executions": jobs, "per_kernel": per_kernel,
                     "command": "bench_circuits.py " + " ".join(argv)}
    for d in ("profiles", "gpurun_out"):
        os.makedirs(os.path.join(ROOT, d), exist_ok=True)
        with open(os.path.join(ROOT, d, "pmc_traffic_circuits.json"), "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {x: v[x] for x in v if x != "per_kernel"}) for k, v in res.items()}))


# ---- the servers' own encryptions (csrc/encrypt.hip): one batch of 8192 at P8192, fused (default) and as five launches ----------------
ENC_BATCH, ENC_REPS = 8192, 3        # 8192 x 512 KiB = 4 GiB of ciphertexts per batch: far beyond the Infinity Cache
ENC = [sys.executable, "-c",
       "import sys; sys.path.insert(0, %r); import numpy as np, torch, fhip_amd as fhe; ctx = fhe.SEALContext.preset('P8192'); "
       "der = fhe.DeviceEncryptor(ctx, fhe.KeyGenerator(ctx, seed=1).public_key()); v = np.linspace(0, 1, %d); "
       "[der.encrypt_values(v) for _ in range(%d)]; torch.cuda.synchronize()" % (ROOT, ENC_BATCH, ENC_REPS)]
ENC_KERNELS = ["k_enc_fused", "k_enc_sample_u", "k_enc_pk_mul", "k_enc_finish", "k_ntt_fwd", "k_ntt_inv", "k_frac_encode"]


def encrypt(f_read, f_write):
    k, n = 4, 8192
    out_bytes = 2 * k * n * 8                                      # the ciphertext: 524,288 B (+ 64 KiB of plaintext coefficients read)
    res = {"output_bytes_per_ciphertext": out_bytes, "ciphertexts_per_dispatch": ENC_BATCH, "read_factor": f_read, "write_factor": f_write,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), summed over every launch of %d x DeviceEncryptor.encrypt_values(%d values) "
                     "at P8192, per ciphertext; factors as calibrated for pmc_traffic.json; tools/collect_traffic.py encrypt" % (ENC_REPS, ENC_BATCH)}
    for tag, env in (("fused", {}), ("five_launches", {"FHE_ENC_UNFUSED": "1"})):
        os.environ.update(env)
        rd, calls = run_pass_sum(["FETCH_SIZE"], "enc_fetch_" + tag, ENC, ENC_KERNELS)
        wr, _ = run_pass_sum(["WRITE_SIZE"], "enc_write_" + tag, ENC, ENC_KERNELS)
        for name in env:
            os.environ.pop(name)
        cts = ENC_BATCH * ENC_REPS
        per, total = {}, 0.0
        for m in ENC_KERNELS:
            if not calls.get(m):
                continue
            r, w = rd[m] * 1024 * f_read / cts, wr[m] * 1024 * f_write / cts
            per[m] = {"read_bytes_per_ciphertext": r, "write_bytes_per_ciphertext": w, "launches_per_batch": calls[m] / ENC_REPS}
            total += r + w
        res[tag] = {"hbm_bytes_per_ciphertext": total, "ratio_to_output_bytes": total / out_bytes, "per_kernel": per}
    for d in ("profiles", "gpurun_out"):
        os.makedirs(os.path.join(ROOT, d), exist_ok=True)
        with open(os.path.join(ROOT, d, "pmc_traffic_encrypt.json"), "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res))


def main():
    cal_bytes = CAL_CTS * 2 * 3 * 4096 * 8
    crd, cnames = run_pass(["FETCH_SIZE"], "cal_fetch", CAL, ["k_poly_f64"])
    cwr, _ = run_pass(["WRITE_SIZE"], "cal_write", CAL, ["k_poly_f64"])
    f_read = cal_bytes / (crd["k_poly_f64"]["FETCH_SIZE"] * 1024)
    f_write = cal_bytes / (cwr["k_poly_f64"]["WRITE_SIZE"] * 1024)
    if "ctct" in sys.argv[1:]:
        return ctct(f_read, f_write)
    if "circuits" in sys.argv[1:]:
        return circuits(f_read, f_write)
    if "encrypt" in sys.argv[1:]:
        return encrypt(f_read, f_write)
    rd, names = run_pass(["FETCH_SIZE"], "fetch", CMD, ["k_dct_rows", "k_dct_cols"])
    wr, _ = run_pass(["WRITE_SIZE"], "write", CMD, ["k_dct_rows", "k_dct_cols"])
    per_kernel, total = {}, 0.0
    for k in ("k_dct_rows", "k_dct_cols"):
        fetch = rd[k]["FETCH_SIZE"] * 1024 * f_read / BLOCKS
        write = wr[k]["WRITE_SIZE"] * 1024 * f_write / BLOCKS
        per_kernel[k] = {"read_bytes_per_block": fetch, "write_bytes_per_block": write, "kernel_name": names[k],
                         "FETCH_SIZE_KiB_per_dispatch": rd[k]["FETCH_SIZE"], "WRITE_SIZE_KiB_per_dispatch": wr[k]["WRITE_SIZE"]}
        total += fetch + write
    res = {"hbm_bytes_per_block": total, "algorithmic_bytes_per_block": 25165824, "ratio_to_algorithmic": total / 25165824,
           "per_kernel": per_kernel, "blocks_per_dispatch": BLOCKS, "kernel_source_hash": kernel_source_hash(),
           "calibration": {"kernel": cnames.get("k_poly_f64"), "known_bytes_each_way": cal_bytes,
                           "FETCH_SIZE_KiB": crd["k_poly_f64"]["FETCH_SIZE"], "WRITE_SIZE_KiB": cwr["k_poly_f64"]["WRITE_SIZE"],
                           "read_factor": f_read, "write_factor": f_write},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --blocks %d --steps 2`, per block; "
                     "KiB -> bytes factors calibrated on the FP64 forward-NTT kernel over 1.5 GiB (same 8 B/lane access pattern, "
                     "known byte count); tools/collect_traffic.py" % BLOCKS}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w") as f:     # profiles/ is not merged back by gpurun
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
