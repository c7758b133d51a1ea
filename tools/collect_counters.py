#!/usr/bin/env python3
"""SQ / TCC performance counters of the library's kernels with rocprofv3 --pmc (run on the GPU box).

The u64 transforms, the BEHZ kernels and the fused DCT pair are bound by VALU issue, not by HBM, so their HBM fraction
says little about how close they are to THEIR ceiling.  This script collects, per kernel (mean per dispatch):

  pass A  SQ_WAVES  SQ_WAVE_CYCLES  SQ_BUSY_CYCLES  SQ_INSTS_VALU  SQ_ACTIVE_INST_VALU  SQ_INSTS_LDS  SQ_ACTIVE_INST_LDS  SQ_WAIT_ANY
  pass B  SQ_WAIT_INST_ANY  SQ_WAIT_INST_LDS  SQ_ACTIVE_INST_ANY  SQ_INSTS_SALU  SQ_INSTS_VMEM_RD  SQ_INSTS_VMEM_WR  SQ_LDS_BANK_CONFLICT  SQ_INSTS_SMEM
  pass C  GRBM_GUI_ACTIVE  GRBM_COUNT          (effective clock = GRBM_GUI_ACTIVE / kernel time)
  pass D  FETCH_SIZE        pass E  WRITE_SIZE  (separate passes, MI355X_MICROARCH.md: 3 + 2 TCC slots)

(counters this rocprofv3 does not list are dropped from a pass), for three workloads: the headline bench (k_dct_rows /
k_dct_cols), tools/bench_ops.py at P8192 (k_ntt_fwd_pm / k_ntt_inv_pm / k_mulplain_pm / k_behz_*) and the configs[2] resize
circuit.  Output: gpurun_out/<tag>/counters.json + counters.txt; every record carries the hash of ALL kernel sources
(csrc/*.hip, *.h), so a file under profiles/ can be matched to the code it describes.

Derived figures (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave, guide "s_memtime tick vs SQ PMC units"):
  valu_insts_per_wave   = SQ_INSTS_VALU / SQ_WAVES
  valu_active_frac      = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES     fraction of a wave's residency spent issuing VALU
  wait_frac             = SQ_WAIT_ANY / SQ_WAVE_CYCLES             parked at s_waitcnt / barrier
  issue_stall_frac      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        wants to issue, pipe busy (other waves of the SIMD)
  effective_clock_ghz   = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (pass C)
  simd_valu_busy        = SQ_ACTIVE_INST_VALU * 4 / (duration_cycles * 1024 SIMDs)   share of all SIMD cycles of the chip in which
                          a VALU instruction held the issue port while the kernel ran (duration from the kernel trace of the same
                          pass, cycles at the effective clock of pass C); simd_lds_busy likewise
  mean_resident_waves_per_simd = SQ_WAVE_CYCLES * 4 / (duration_cycles * 1024)        achieved occupancy
"""
import collections
import csv
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = sys.executable
XCDS, SIMDS = 8, 1024            # MI355X: 8 XCDs, 256 CUs x 4 SIMDs
WORKLOADS = {
    "bench": ([PY, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--blocks", "256", "--cpu-blocks", "0", "--no-verify"],
              ["k_dct_rows", "k_dct_cols"]),
    "ops8192": ([PY, os.path.join(ROOT, "tools", "bench_ops.py"), "P8192", "1024"],
                ["k_ntt_fwd", "k_ntt_inv", "k_mulplain", "k_behz_tensor_intt", "k_behz_floor_back", "k_behz_to_bsk", "k_eltwise", "k_dyadic"]),
    "resize": ([PY, os.path.join(ROOT, "bench_circuits.py"), "resize", "--max-pixels", "512"],
               ["k_"]),
    "decode": ([PY, os.path.join(ROOT, "bench_circuits.py"), "decode"],
               ["k_"]),
    # every kernel of the circuits' launch sequences (bench_circuits.py carries the launch-time-weighted issue fraction of these)
    "resize_shared": ([PY, os.path.join(ROOT, "bench_circuits.py"), "resize", "--shared", "--max-pixels", "1024"], ["k_"]),
    "decode_relin30": ([PY, os.path.join(ROOT, "bench_circuits.py"), "decode", "--relin", "30"], ["k_"]),
    "resize_relin30": ([PY, os.path.join(ROOT, "bench_circuits.py"), "resize", "--relin", "30", "--max-pixels", "512"], ["k_"]),
    "resize_relin30_cubic": ([PY, os.path.join(ROOT, "bench_circuits.py"), "resize", "--relin", "30", "--relin-placement", "cubic", "--max-pixels", "512"], ["k_"]),
    "resize_relin60_sample": ([PY, os.path.join(ROOT, "bench_circuits.py"), "resize", "--relin", "60", "--relin-placement", "sample", "--max-pixels", "512"], ["k_"]),
    "encrypt": ([PY, os.path.join(ROOT, "tools", "bench_encrypt.py"), "P8192", "512"], ["k_enc_fused", "k_frac_encode", "k_dec_"]),
    "seal23": ([PY, os.path.join(ROOT, "bench.py"), "--preset", "SEAL23_4096", "--steps", "2", "--warmup", "1", "--blocks", "256", "--cpu-blocks", "0", "--no-verify"],
               ["k_dct_rows_u64", "k_dct_cols_u64"]),
}
PASSES = {
    "A": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY"],
    "B": ["SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SMEM"],
    "C": ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
    "D": ["FETCH_SIZE"],
    "E": ["WRITE_SIZE"],
}


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def available():
    try:
        out = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=300, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp")).stdout
    except Exception:
        return None
    names = set()
    for tok in out.replace(",", " ").replace(":", " ").split():
        if tok.isupper() or "_" in tok:
            names.add(tok.strip())
    return names


def short(kernel, matches):
    best = None
    for m in matches:
        if m in kernel and (best is None or len(m) > len(best)):
            best = m
    if best is None:
        return None
    i = kernel.index(best)
    j = kernel.find("(", i)
    return kernel[i:j if j > 0 else i + 60].strip()


def run_pass(outdir, counters, cmd, matches):
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "p", "--pmc", *counters, "--"] + cmd,
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        return None, r.stderr[-500:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(outdir, "p_counter_collection.csv")
    if not os.path.exists(path):
        return None, "no counter file"
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"], matches)
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = collections.defaultdict(list)
    tpath = os.path.join(outdir, "p_kernel_trace.csv")
    if os.path.exists(tpath):
        with open(tpath) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"], matches)
                if k:
                    dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    res = {}
    for k, cs in acc.items():
        res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        res[k]["dispatches"] = max(len(v) for v in cs.values())
        if dur.get(k):
            res[k]["duration_us_in_this_pass"] = sum(dur[k]) / len(dur[k]) / 1e3
    return res, None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "counters"
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else list(WORKLOADS)
    outroot = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(outroot, exist_ok=True)
    avail = available()
    result = {"kernel_source_hash": source_hash(), "tool": "tools/collect_counters.py", "workloads": {}, "dropped_counters": []}
    for w in which:
        cmd, matches = WORKLOADS[w]
        merged = collections.defaultdict(dict)
        for pname, counters in PASSES.items():
            use = [c for c in counters if avail is None or c in avail]
            for c in counters:
                if c not in use and c not in result["dropped_counters"]:
                    result["dropped_counters"].append(c)
            if not use:
                continue
            res, err = run_pass(os.path.join(outroot, "%s_%s" % (w, pname)), use, cmd, matches)
            if res is None:
                result.setdefault("errors", []).append({"workload": w, "pass": pname, "error": err})
                continue
            for k, vals in res.items():
                for c, v in vals.items():
                    merged[k][c if c not in ("dispatches", "duration_us_in_this_pass") else "%s_pass%s" % (c, pname)] = v
        for k, m in merged.items():
            wc, waves = m.get("SQ_WAVE_CYCLES"), m.get("SQ_WAVES")
            d = {}
            if waves:
                d["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0) / waves
                d["lds_insts_per_wave"] = m.get("SQ_INSTS_LDS", 0) / waves
                d["wave_quadcycles_per_wave"] = (wc or 0) / waves
            if wc:
                for name, c in (("valu_active_frac", "SQ_ACTIVE_INST_VALU"), ("lds_active_frac", "SQ_ACTIVE_INST_LDS"), ("wait_frac", "SQ_WAIT_ANY"),
                                ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("any_active_frac", "SQ_ACTIVE_INST_ANY")):
                    if c in m:
                        d[name] = m[c] / wc
            dur_c, gui = m.get("duration_us_in_this_pass_passC"), m.get("GRBM_GUI_ACTIVE")
            if dur_c and gui:
                # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (a 1.5 ms kernel reads 23.7 M "cycles": 8 x 1.87 GHz x 1.59 ms)
                d["effective_clock_ghz"] = gui / XCDS / dur_c / 1e3
                dur_a = m.get("duration_us_in_this_pass_passA")
                if dur_a and "SQ_ACTIVE_INST_VALU" in m:
                    cycles = dur_a * 1e3 * d["effective_clock_ghz"]
                    # SQ_ACTIVE_INST_VALU: quad-cycles summed over every SIMD of the chip -> share of all SIMD cycles of the
                    # kernel's run in which a VALU instruction occupied the SIMD's issue port
                    d["simd_valu_busy"] = m["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * SIMDS)
                    if "SQ_ACTIVE_INST_LDS" in m:
                        d["simd_lds_busy"] = m["SQ_ACTIVE_INST_LDS"] * 4 / (cycles * SIMDS)
                    if "SQ_WAVE_CYCLES" in m:
                        d["mean_resident_waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4 / (cycles * SIMDS)
            if "FETCH_SIZE" in m:
                d["fetch_bytes_x2_guide_correction"] = m["FETCH_SIZE"] * 1024 * 2
            if "WRITE_SIZE" in m:
                d["write_bytes"] = m["WRITE_SIZE"] * 1024
            m["derived"] = d
        result["workloads"][w] = {"command": " ".join(os.path.relpath(c, ROOT) if os.path.isabs(c) and c.startswith(ROOT) else c for c in cmd), "kernels": merged}
    with open(os.path.join(outroot, "counters.json"), "w") as f:
        json.dump(result, f, indent=1)
    with open(os.path.join(outroot, "counters.txt"), "w") as f:
        f.write("kernel_source_hash %s\n" % result["kernel_source_hash"])
        for w, rec in result["workloads"].items():
            f.write("\n== %s: %s\n" % (w, rec["command"]))
            for k, m in rec["kernels"].items():
                f.write("%s\n" % k)
                for c in sorted(x for x in m if x != "derived"):
                    f.write("    %-36s %18.1f\n" % (c, m[c]))
                for c, v in m["derived"].items():
                    f.write("    -> %-33s %18.4f\n" % (c, v))
    print(open(os.path.join(outroot, "counters.txt")).read())


if __name__ == "__main__":
    main()
