#!/usr/bin/env python3
"""bench.py -- headline benchmark: encrypted 8x8 blocks/sec for homomorphic DCT+quant.

Workload (BASELINE.json configs[1]): 1024 ciphertext blocks ("256x256 gray") per GPU,
poly_modulus_degree 4096, 3 coefficient moduli {0xffffee001, 0xffffc4001, 0x1ffffe0001}, t = 2^14;
one step = encrypted_dct (homo/fhe_image.h:196-288) + quantize_fhe (:294-305) over every block,
inputs already resident in HBM (synthetic random-residue ciphertexts).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B]        (N > 1: starts N ranks itself, see ensure_world)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Multi-GPU: blocks are sharded, one process per GPU, no data-path
collective (weak scaling: 1024 blocks per GPU); RCCL is used for the barrier, the max-over-ranks
time and an all-reduce of the output digests.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver; before torch / HIP load

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_BLOCK = 128 * 2 * 3 * 4096 * 8   # read 64 ct + write 64 ct, ct = 2*3*4096*8 B  
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
NOMINAL_CLOCK_GHZ = 2.4                    # MI355X_MICROARCH.md: peak engine clock; under these kernels the chip runs at 1.75-1.9 GHz (package power limit)


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


def all_kernel_source_hash():
    """the hash tools/isa_counts.py and tools/collect_counters.py tie their records to: every file of csrc/"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def _tracked_counter_pairs():
    """(tag, isa counts, counters) of every profiles/<tag>_isa_counts.json + <tag>_counters.json pair that describes the kernel
    sources that are running (hash over csrc/), newest tag first"""
    import glob
    tags = sorted({os.path.basename(p).split("_isa_counts.json")[0] for p in glob.glob(os.path.join(ROOT, "profiles", "*_isa_counts.json"))}, reverse=True)
    want = all_kernel_source_hash()
    for tag in tags:
        cpath = os.path.join(ROOT, "profiles", tag + "_counters.json")
        if not os.path.exists(cpath):
            continue
        isa, cnt = json.load(open(os.path.join(ROOT, "profiles", tag + "_isa_counts.json"))), json.load(open(cpath))
        if isa.get("kernel_source_hash") == want and cnt.get("kernel_source_hash") == want:
            yield tag, isa, cnt


def issue_roofline_workload(workload):
    """The issue-side roofline of a whole launch SEQUENCE (bench_circuits.py: configs[2] / configs[3] are a few hundred launches of
    a dozen kernels): for every kernel of the counters workload `workload` (tools/collect_counters.py) the issue floor of one
    dispatch -- waves x dynamic VALU instructions per wave x issue cycles per instruction of ITS static mix / (1024 SIMDs x clock) --
    and its measured duration, both times the number of dispatches; frac = sum of floors / sum of durations (a launch-time-weighted
    mean of the kernels' issue fractions).  `frac` uses the clock the counters saw while the kernel ran (GRBM_GUI_ACTIVE / time:
    1.75-1.9 GHz at the package power limit), `frac_at_nominal_clock` the 2.4 GHz the guide gives as the chip's peak -- the same
    instruction stream against the clock the silicon does not sustain under this load.  None when no tracked pair matches."""
    import re
    want = all_kernel_source_hash()
    for tag, isa, cnt in _tracked_counter_pairs():
        rec = cnt["workloads"].get(workload)
        if not rec:
            continue
        norm = {re.sub(r"\s+", "", k): v for k, v in isa["kernels"].items()}
        floor = floor_nom = meas = 0.0
        per = []
        for k, m in rec["kernels"].items():
            st, d = norm.get(re.sub(r"\s+", "", k)), m.get("derived", {})
            waves, insts, dur, clk = m.get("SQ_WAVES"), m.get("SQ_INSTS_VALU"), m.get("duration_us_in_this_pass_passA"), d.get("effective_clock_ghz")
            disp = m.get("dispatches_passA")
            if not (st and st.get("valu") and waves and insts and dur and clk and disp):
                continue
            cyc = insts / waves * st["issue_cycles_per_wave"] / st["valu"] * waves / 1024.0          # SIMD-cycles of issue per dispatch / 1024 SIMDs
            f, fn = cyc / (clk * 1e3), cyc / (NOMINAL_CLOCK_GHZ * 1e3)
            floor, floor_nom, meas = floor + f * disp, floor_nom + fn * disp, meas + dur * disp
            per.append({"kernel": k, "dispatches": int(disp), "us_per_dispatch": dur, "issue_frac": f / dur, "issue_frac_at_nominal_clock": fn / dur,
                        "effective_clock_ghz": clk, "share_of_counted_time": dur * disp})
        if not meas:
            continue
        for r in per:
            r["share_of_counted_time"] /= meas
        per.sort(key=lambda r: -r["share_of_counted_time"])
        return {"bound": "valu-issue", "frac": floor / meas, "frac_at_nominal_clock": floor_nom / meas, "nominal_clock_ghz": NOMINAL_CLOCK_GHZ,
                "counted_kernel_time_us": meas, "kernels": per[:8],
                "source": "profiles/%s_isa_counts.json + profiles/%s_counters.json, workload '%s': %s (kernel_source_hash %s)" % (tag, tag, workload, rec.get("command"), want)}
    return {"bound": "valu-issue", "frac": None,
            "source": "no profiles/*_isa_counts.json + *_counters.json pair with a '%s' workload matches the running kernel sources (%s); re-run tools/isa_counts.py and "
                      "tools/collect_counters.py" % (workload, want)}


def issue_roofline(kernel_names, blocks, dev_ms_per_step):
    """Issue-side view of the fused kernels from TRACKED files only (the way tools/issue_roofline.py does it): the static
    instruction mix of each kernel (profiles/<tag>_isa_counts.json: cycles per VALU instruction of ITS mix) times the VALU
    instructions a wave really executes and the waves per block (profiles/<tag>_counters.json: SQ_INSTS_VALU, SQ_WAVES, and the
    effective clock GRBM_GUI_ACTIVE / time) = the time one block needs if every SIMD of the chip issued back to back and nothing
    ever waited; frac = that floor / the time measured now.  Quoted only when both files describe the sources that are running."""
    import glob
    import re
    tags = sorted({os.path.basename(p).split("_isa_counts.json")[0] for p in glob.glob(os.path.join(ROOT, "profiles", "*_isa_counts.json"))}, reverse=True)
    want = all_kernel_source_hash()
    for tag in tags:
        cpath = os.path.join(ROOT, "profiles", tag + "_counters.json")
        if not os.path.exists(cpath):
            continue
        isa, cnt = json.load(open(os.path.join(ROOT, "profiles", tag + "_isa_counts.json"))), json.load(open(cpath))
        if isa.get("kernel_source_hash") != want or cnt.get("kernel_source_hash") != want:
            continue
        rec = cnt["workloads"].get("bench", {})
        launches = int(re.search(r"--blocks (\d+)", rec.get("command", "--blocks 256")).group(1))
        floor_us, floor_nom_us, per_kernel = 0.0, 0.0, {}
        for name in kernel_names:
            m = next((v for k, v in rec.get("kernels", {}).items() if k.startswith(name + "<") or k == name), None)
            st = next((v for k, v in isa["kernels"].items() if re.sub(r"\s+", "", k).startswith(name + "<")), None)
            if not (m and st and st.get("valu")):
                return None
            clk = m["derived"]["effective_clock_ghz"]
            cyc = st["issue_cycles_per_wave"] / st["valu"]
            us = m["SQ_WAVES"] / launches * m["derived"]["valu_insts_per_wave"] * cyc / (1024 * clk * 1e3)
            per_kernel[name] = {"valu_insts_per_wave": m["derived"]["valu_insts_per_wave"], "issue_cycles_per_valu_inst": cyc, "effective_clock_ghz": clk,
                                "simd_valu_busy": m["derived"].get("simd_valu_busy"), "issue_floor_us_per_block": us,
                                "issue_floor_us_per_block_at_nominal_clock": us * clk / NOMINAL_CLOCK_GHZ}
            floor_us += us
            floor_nom_us += us * clk / NOMINAL_CLOCK_GHZ
        measured = dev_ms_per_step * 1e3 / blocks
        # frac: against the clock the counters saw under this pair (1.8-1.9 GHz: the package power limit); frac_at_nominal_clock: the same
        # instruction stream against the 2.4 GHz the guide gives as the chip's peak clock
        return {"bound": "valu-issue", "issue_floor_us_per_block": floor_us, "measured_us_per_block": measured,
                "frac": floor_us / measured, "frac_at_nominal_clock": floor_nom_us / measured, "nominal_clock_ghz": NOMINAL_CLOCK_GHZ,
                "issue_floor_us_per_block_at_nominal_clock": floor_nom_us, "kernels": per_kernel,
                "source": "profiles/%s_isa_counts.json + profiles/%s_counters.json (kernel_source_hash %s)" % (tag, tag, want)}
    return {"bound": "valu-issue", "frac": None,
            "source": "no profiles/*_isa_counts.json + *_counters.json pair matches the running kernel sources (%s); re-run tools/isa_counts.py and tools/collect_counters.py" % want}


def ensure_world(gpus, script, argv):
    """`python bench.py --gpus N` is the WHOLE multi-GPU launch: without WORLD_SIZE in the environment and N > 1 the process
    replaces itself by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port
    <free port> <script> <same arguments>` (one rank per GPU).  Under a launcher (WORLD_SIZE set) the two numbers must agree:
    a line measured on another number of ranks than --gpus says is refused, not printed.  More ranks than HIP devices is an
    error as well (RCCL refuses two ranks on one device) unless FHE_BENCH_BACKEND=gloo (tests: ranks share the devices)."""
    env_world = os.environ.get("WORLD_SIZE")
    if gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if env_world is not None:
        if int(env_world) != gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE is %s: the launcher and the argument disagree (launch with --nproc-per-node %d, or run "
                             "`python %s --gpus %d` without a launcher)" % (gpus, env_world, gpus, os.path.basename(script), gpus))
        return
    if gpus == 1:
        return
    if os.environ.get("FHE_BENCH_BACKEND", "nccl") != "gloo":
        import torch
        have = torch.cuda.device_count()
        if gpus > have:
            raise SystemExit("--gpus %d but this node exposes %d HIP device(s): one rank per GPU (FHE_BENCH_BACKEND=gloo lets test ranks share devices)" % (gpus, have))
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


_RCCL_PROBE_CHILD = r"""
import datetime, os, sys
if os.environ.get("FHE_BENCH_PROBE_FAIL") == "1":        # test hook: the fall-back path of open_process_group
    sys.exit(3)
import torch, torch.distributed as dist
r, w, lr = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", rank=r, world_size=w, timeout=datetime.timedelta(seconds=float(sys.argv[1])), device_id=torch.device("cuda", lr))
ok = 1
t = torch.ones(1, dtype=torch.int64, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
ok &= int(int(t.item()) == w)
if w > 1:                                                # the point-to-point pattern of parallel.WaveGather, once around the ring
    a = torch.full((1 << 20,), r, dtype=torch.int64, device="cuda")
    b = torch.empty_like(a)
    for x in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, (r + 1) % w), dist.P2POp(dist.irecv, b, (r - 1) % w)]):
        x.wait()
    torch.cuda.synchronize()
    ok &= int(int(b[0].item()) == (r - 1) % w and int(b[-1].item()) == (r - 1) % w)
f = torch.tensor([ok], dtype=torch.int64, device="cuda")
dist.all_reduce(f, op=dist.ReduceOp.MIN)                 # every rank learns whether EVERY rank passed: the parents decide alike
torch.cuda.synchronize()
print("RCCL PROBE OK" if int(f.item()) == 1 else "RCCL PROBE MISMATCH", flush=True)
os._exit(0)
"""


def rccl_probe(limit):
    """No box this repository ever ran on had two devices: the first RCCL rendezvous between two GPUs happens on the driver's node.  So that a
    fabric / IPC problem there cannot hang the measurement, every rank first runs RCCL's initialisation, an all-reduce, one send/recv around the
    ring and a closing all-reduce of the verdicts in a CHILD process (its own store: MASTER_PORT + 29, rank 0's child hosts it), bounded by
    `limit` seconds.  All children pass or none does (the closing all-reduce), so every rank takes the same branch in open_process_group."""
    import subprocess
    env = dict(os.environ)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)            # the children rendezvous among themselves, not on the launcher's store
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    base = int(env.get("MASTER_PORT", "29500"))
    env["MASTER_PORT"] = str(base + 29 if base + 29 < 65536 else base - 29)
    t0 = time.perf_counter()
    child = subprocess.Popen([sys.executable, "-c", _RCCL_PROBE_CHILD, str(limit)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    detail = None
    try:
        out, err = child.communicate(timeout=limit + 30)
    except subprocess.TimeoutExpired:
        child.kill()                                         # the process this function started, by its own handle
        out, err = child.communicate()
        detail = "no completion within %.0f s" % (limit + 30)
    ok = "RCCL PROBE OK" in (out or "")
    if not ok and detail is None:
        lines = [ln for ln in ((out or "") + "\n" + (err or "")).splitlines() if ln.strip()]
        detail = ("exit code %s: %s" % (child.returncode, lines[-1] if lines else "no output"))[:300]
    return {"ok": ok, "seconds": time.perf_counter() - t0, "detail": detail}


def open_process_group(world, local_rank, requested):
    """-> (torch.distributed, backend in use, probe record or None).  requested == "nccl" with more than one rank: RCCL after rccl_probe passed
    on every rank; otherwise the SAME job continues with gloo for its control plane (barrier, max over ranks, digests, per-rank times on host
    tensors; parallel.WaveGather stages through pinned host memory) -- every rank still computes on its own GPU, the compute-only `value` has no
    data-path collective either way, and the line says which backend ran.  FHE_BENCH_RCCL_PROBE=0 skips the probe."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend, probe = requested, None
    probing = world > 1 or os.environ.get("FHE_BENCH_PROBE_FORCE") == "1"          # the second: tests on a one-GPU box
    if requested == "nccl" and probing and os.environ.get("FHE_BENCH_RCCL_PROBE", "1") != "0":
        probe = rccl_probe(float(os.environ.get("FHE_BENCH_RCCL_PROBE_TIMEOUT", "120")))
        if not probe["ok"]:
            backend = "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist, backend, probe


def backend_label(backend, requested, shared_devices):
    if backend == "nccl":
        return "rccl"
    if requested == "nccl":
        return "gloo (FALL-BACK: the RCCL probe between this job's devices failed; control plane on host tensors, every rank computes on its own GPU)"
    return backend + (" (test mode: ranks share devices)" if shared_devices else " (requested with FHE_BENCH_BACKEND)")


def cpu_baseline(n_blocks_sample):
    """The CPU oracle (op-at-a-time port of the SEAL path) timed on this host, 1 thread."""
    from oracle import oracle as om
    om.build()
    orc = om.Oracle.preset("P4096")
    blocks = orc.random_ct(n_blocks_sample * 64, seed=om.SEED).reshape(n_blocks_sample, 64, 2, orc.k, orc.n)
    t0 = time.perf_counter()
    digs = []
    for b in range(n_blocks_sample):
        digs.append(om.digest(orc.dct_quant(blocks[b], om.YQT)))
    dt = time.perf_counter() - t0
    cpu = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": n_blocks_sample / dt, "unit": "blocks/s", "cores": 1, "kind": "port",
        # context only: the reference's own published timings (benchmark/results.txt:41, n=4096, unknown CPU, one
        # thread): 199 ms per encrypted_dct call + 64 multiply_plain for quantize_fhe at ~0.75 ms => ~4.0 blocks/s
        "reference_published_blocks_per_s": 4.0,
        "sample": "%d blocks of the same workload (n=4096,k=3), oracle/libfhe_oracle.so op-at-a-time, 1 thread of %d on %s"
                  % (n_blocks_sample, os.cpu_count() or 0, cpu),
    }


def cpu_baseline_all_cores(blocks_per_thread=1):
    """Same oracle, OpenMP over independent blocks (extra field, not the `cpu_baseline` object)."""
    from oracle import oracle as om
    orc = om.Oracle.preset("P4096", omp=True)          # the optional OpenMP build (oracle/libfhe_oracle_omp.so)
    nb = max(1, (os.cpu_count() or 1) * blocks_per_thread)
    nb = min(nb, 256)                                              # 12 MiB per block
    blocks = orc.random_ct(nb * 64, seed=om.SEED).reshape(nb, 64, 2, orc.k, orc.n)
    t0 = time.perf_counter()
    _, threads = orc.dct_quant_blocks(blocks, om.YQT)
    dt = time.perf_counter() - t0
    return {"value": nb / dt, "unit": "blocks/s", "cores": threads, "kind": "port",
            "sample": "%d blocks, OpenMP over blocks, %d threads, same oracle" % (nb, threads)}


XGMI_LINK_GBS = 153.0                      # MI355X_MICROARCH.md: 7 xGMI links per GPU, ~153 GB/s each, point to point
PCIE_GBS = 64.0                            # PCIe Gen5 x16 per direction (nominal); 55-60 GB/s is what pinned copies sustain


class Watchdog:
    """Bounds a leg that contains point-to-point transfers no box has run yet (the RCCL wave gather between two devices): when the
    deadline passes, `on_timeout()` runs (rank 0: print the line measured so far, with the leg marked as timed out) and the process
    ends with os._exit(0) -- every rank arms the same deadline after the same barrier, so the whole job ends instead of one rank
    waiting for another inside a collective.  `value` (the compute-only leg) is measured BEFORE any such leg."""

    def __init__(self, seconds, on_timeout):
        import threading
        self.cancelled = threading.Event()
        self.t = threading.Thread(target=self._run, args=(seconds, on_timeout), daemon=True)
        self.t.start()

    def _run(self, seconds, on_timeout):
        if not self.cancelled.wait(seconds):
            try:
                on_timeout()
            finally:
                sys.stdout.flush()
                os._exit(0)

    def cancel(self):
        self.cancelled.set()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=1024, help="blocks per GPU per step")
    ap.add_argument("--cpu-blocks", type=int, default=16, help="CPU baseline sample size (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--preset", default="P4096", help="parameter set (default: the BASELINE.json configuration)")
    ap.add_argument("--gather", choices=["none", "wave", "local"], default="none",
                    help="what the TIMED region (`value`) does with the outputs.  wave: every wave of output ciphertexts is sent to rank 0 "
                         "(RCCL send/recv over xGMI, overlapped with the next wave's compute) and drained there by digest; local: every "
                         "rank drains ITS OWN waves to pinned host memory over its own PCIe link (parallel.LocalDrain; the consumer that "
                         "scales with the GPU count); none (default): outputs stay sharded in HBM (SURVEY.md 8e).  With N > 1 and the "
                         "default, both gather variants are ALSO measured after the timed region and reported in `gather` (--gather-legs)")
    ap.add_argument("--gather-wave-blocks", type=int, default=64)
    ap.add_argument("--gather-legs", choices=["auto", "on", "off"], default="auto",
                    help="the `gather` object of the line (with-gather figures next to the compute-only `value`): auto = when N > 1")
    ap.add_argument("--gather-steps", type=int, default=3, help="timed steps of each gather leg (one untimed step before them)")
    ap.add_argument("--n1-leg", choices=["auto", "on", "off"], default="auto",
                    help="rank 0 alone times the same step before the all-rank region (in-run N = 1 figure -> weak_scaling.efficiency): auto = when N > 1")
    args = ap.parse_args()
    ensure_world(args.gpus, os.path.abspath(__file__), sys.argv[1:])

    import numpy as np
    import torch
    import fhip_amd as fhe

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists)")
    # FHE_BENCH_BACKEND=gloo (tests only): the world > 1 code path of this file on a box with fewer devices than ranks -- RCCL refuses
    # two ranks on one device, gloo does not care; ranks then share devices round-robin and the collectives run on host tensors
    backend = os.environ.get("FHE_BENCH_BACKEND", "nccl")
    shared_devices = False
    if backend == "gloo":
        shared_devices = world > torch.cuda.device_count()
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    requested_backend, rccl_probe_record = backend, None
    dist = None
    # FHE_BENCH_FORCE_DIST=1: a process group even for ONE rank -- the RCCL initialisation, barrier, all-reduce and all-gather of the
    # multi-GPU path execute (trivially) on a one-GPU box; tests/test_gpu_multi.py uses it so that the collective code has run on RCCL
    if world > 1 or os.environ.get("FHE_BENCH_FORCE_DIST") == "1":
        dist, backend, rccl_probe_record = open_process_group(world, local_rank, backend)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    ctx = fhe.SEALContext.preset(args.preset, device=local_rank)
    ev = fhe.Evaluator(ctx)
    plan = fhe.DctPlan(ctx, fhe.YQT)
    B = args.blocks
    words_per_block = 64 * 2 * ctx.k * ctx.n
    out_bytes_per_block = words_per_block * 8                # 64 ct(2) written per block: 12 MiB at n=4096, k=3
    # global block index g = rank * B + b: any GPU count generates the same bytes for block g
    first_index = rank * B * words_per_block
    blocks = ctx.random_ct(B, 64, seed=fhe.SEED, first_index=first_index)
    out = torch.empty_like(blocks)
    torch.cuda.synchronize()

    legs = args.gather_legs == "on" or (args.gather_legs == "auto" and world > 1 and args.gather == "none")
    need_waves = args.gather != "none" or legs
    wave = n_waves = None
    if need_waves:
        if B % args.gather_wave_blocks:
            raise SystemExit("--blocks must be a multiple of --gather-wave-blocks")
        wave, n_waves = args.gather_wave_blocks, B // args.gather_wave_blocks
    wave_shape = ((wave,) + tuple(blocks.shape[1:])) if need_waves else None

    # ---- the three step functions: outputs stay in HBM / waves to rank 0 over RCCL / waves to pinned host memory --------------------
    def step_none():
        ev.dct8x8_quant(plan, blocks, out=out)

    gather = None
    wave_digests = None
    if args.gather == "wave" or legs:
        # rank 0 drains every wave (its own and the peers') by digest: one u64 per (source rank, wave)
        wave_digests = torch.zeros(world * n_waves, dtype=torch.int64, device=blocks.device)

        def consume(src, w, t):
            ctx.digest_into(t.view(-1), wave_digests[src * n_waves + w:src * n_waves + w + 1],
                            index0=(src * B + w * wave) * words_per_block)
        if world > 1:
            gather = fhe.parallel.WaveGather(wave_shape, blocks.dtype, blocks.device, n_waves, consume=consume)

    def step_wave():
        for w in range(n_waves):
            src = blocks[w * wave:(w + 1) * wave]
            if gather is None:                                   # one GPU: nothing to move, drain in place
                ev.dct8x8_quant(plan, src, out=out[w * wave:(w + 1) * wave])
                consume(0, w, out[w * wave:(w + 1) * wave])
            else:
                buf = gather.acquire()
                ev.dct8x8_quant(plan, src, out=buf)
                gather.commit(w)
        if gather is not None:
            gather.finish()
            gather.reset()

    local = None
    drained = [0]
    if args.gather == "local" or legs:
        def on_host(w, host_tensor):                            # where a per-GPU stream writer would take over
            drained[0] += host_tensor.numel() * 8
        local = fhe.parallel.LocalDrain(wave_shape, blocks.dtype, blocks.device, consume=on_host)

    def step_local():
        for w in range(n_waves):
            buf = local.acquire()
            ev.dct8x8_quant(plan, blocks[w * wave:(w + 1) * wave], out=buf)
            local.commit(w)
        local.finish()
        local.reset()

    step = {"none": step_none, "wave": step_wave, "local": step_local}[args.gather]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        """exactly `steps` calls of fn bracketed by barrier + synchronize on both sides; (max-over-ranks wall s, this rank's wall s,
        this rank's device ms per step from HIP events on the launch stream)"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()                      # same stream the C ABI launches on (torch current stream)
        for _ in range(steps):
            fn()
        e1.record()
        barrier()
        mine = time.perf_counter() - t0
        worst = mine
        if dist is not None:
            tt = torch.tensor([mine], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            worst = float(tt.item())
        return worst, mine, e0.elapsed_time(e1) / steps

    for _ in range(args.warmup):
        step()

    # ---- in-run N = 1 leg: rank 0 alone, the other ranks idle at the barrier (weak scaling: the same B blocks per GPU) ----------------
    n1 = None
    if args.n1_leg == "on" or (args.n1_leg == "auto" and world > 1):
        barrier()
        if rank == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(args.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            n1_wall = time.perf_counter() - t0
            n1 = {"blocks_per_s": B * args.steps / n1_wall, "ms_per_step": n1_wall / args.steps * 1e3, "device_ms_per_step": e0.elapsed_time(e1) / args.steps}
        barrier()

    # ---- timed region: exactly K steps ------------------------------------------------------------
    wall, my_wall, dev_ms_per_step = timed(step, args.steps)
    rank_ms, rank_dev_ms, rccl_ranks = [my_wall / args.steps * 1e3], [dev_ms_per_step], 1
    if dist is not None:
        # self-check of the multi-GPU run: every rank contributes 1 to a SUM all-reduce (= ranks RCCL really connected)
        # and its own per-step times to an all-gather, so the one JSON line shows the whole job
        ones = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(ones.item())
        mine = torch.tensor([my_wall / args.steps * 1e3, dev_ms_per_step], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t[0].item()) for t in every]
        rank_dev_ms = [float(t[1].item()) for t in every]

    # ---- verification: sampled blocks against the CPU oracle, digest over everything -----------------
    def verify_local_drain():
        """the drained path ends in page-locked HOST buffers: one more pass whose consumer brings every drained wave back and digests
        it with its global index -- the bytes that really left over PCIe -- next to the digest of an in-HBM evaluation"""
        back = torch.empty(wave_shape, dtype=blocks.dtype, device=blocks.device)
        host_digests = torch.zeros(n_waves, dtype=torch.int64, device=blocks.device)

        def verify_on_host(w, host_tensor):
            back.copy_(host_tensor)
            ctx.digest_into(back.view(-1), host_digests[w:w + 1], index0=first_index + w * wave * words_per_block)
        keep, local.consume = local.consume, verify_on_host
        step_local()
        torch.cuda.synchronize()
        local.consume = keep
        drained_digest = int(host_digests.cpu().numpy().view(np.uint64).sum(dtype=np.uint64))
        ev.dct8x8_quant(plan, blocks, out=out)
        in_hbm = ctx.digest(out.view(-1), index0=first_index)
        return drained_digest, in_hbm

    if args.gather == "local":
        drained_digest, in_hbm = verify_local_drain()
        if drained_digest != in_hbm:
            raise SystemExit("--gather local: the waves drained to host memory differ from the in-HBM result (%016x != %016x)" % (drained_digest, in_hbm))
        digest_all = fhe.parallel.combine_digests(in_hbm)
    elif args.gather == "wave":     # the root holds the digest of every rank's last step; the ciphertexts were not kept
        digest_all = int(wave_digests.cpu().numpy().view(np.uint64).sum(dtype=np.uint64)) if rank == 0 else 0
        if gather is not None:
            out = None
        if dist is not None:
            dist.barrier()
    else:
        digest_all = fhe.parallel.combine_digests(ctx.digest(out.view(-1), index0=first_index))
    verified = None
    if rank == 0 and not args.no_verify:
        from oracle import oracle as om
        om.build()
        orc = om.Oracle.preset(args.preset)
        sample = [0, B - 1] if B > 1 else [0]
        ok = True
        if out is None:                                      # gathered run: recompute the sampled blocks locally
            out = ev.dct8x8_quant(plan, blocks)
        for b in sample:
            ref = orc.dct_quant(fhe.to_host(blocks[b]), om.YQT)
            ok &= bool(np.array_equal(fhe.to_host(out[b]), ref))
        verified = ok

    res = None
    if rank == 0:
        total_blocks = B * world * args.steps
        value = total_blocks / wall
        bytes_per_block = 128 * 2 * ctx.k * ctx.n * 8          # = BYTES_PER_BLOCK for the default preset
        achieved = B * bytes_per_block / (dev_ms_per_step * 1e-3) / 1e9
        # HBM-side traffic: profiles/pmc_traffic.json, written by tools/collect_traffic.py from separate rocprofv3
        # --pmc passes.  It is only quoted when it was measured on the kernel sources that are running now.
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and args.preset == "P4096":
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("kernel_source_hash") == kernel_source_hash():
                traffic = tj.get("hbm_bytes_per_block", 0) * B or None
                traffic_src = tj.get("source")
            else:
                traffic_src = "profiles/pmc_traffic.json was measured on other kernel sources (%s); re-run tools/collect_traffic.py" % tj.get("kernel_source_hash")
        path = fhe._lib.load().fhe_dct_path(ctx.h)             # 1 fused FP64 pair, 2 fused u64 pair, 0 general three-launch path
        pm = fhe._lib.load().fhe_arith_path(ctx.h) & 3           # pseudo-Mersenne butterflies on the q-base (csrc/ntt_core.h)
        kernels = {1: "k_dct_rows + k_dct_cols", 2: "k_dct_rows_u64 + k_dct_cols_u64",
                   0: "k_ntt_fwd_pm + k_dct_lines_pm x 2 + k_ntt_inv_pm" if pm == 1 else "k_ntt_fwd + k_dct_slots + k_ntt_inv"}[path]
        res = {
            "metric": "encrypted 8x8 blocks/sec (homomorphic DCT+quant)",
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if path == 1 else "u64", "data": "synthetic",
            "rccl_ranks": rccl_ranks, "collective_backend": backend_label(backend, requested_backend, shared_devices) if dist is not None else None, "rccl_probe": rccl_probe_record,
            "ms_per_step_per_rank": rank_ms,
            "config": {"workload": "homomorphic 8x8 DCT+quant, %d ciphertext blocks per GPU, n=%d, %d coeff moduli, t=2^14" % (B, ctx.n, ctx.k),
                       "blocks_per_gpu": B, "poly_modulus_degree": ctx.n, "coeff_moduli": [hex(x) for x in ctx.q],
                       "sharding": ("blocks x%d, no data-path collective" % world) if args.gather == "none" else
                                   ("blocks x%d, every rank drains its own %d-block waves to pinned host memory over its own PCIe link" % (world, args.gather_wave_blocks))
                                   if args.gather == "local" else
                                   ("blocks x%d, every %d-block wave of outputs sent to rank 0 (RCCL send/recv) and drained by digest" % (world, args.gather_wave_blocks)),
                       "gather": args.gather,
                       "arithmetic": {1: "exact integer residues carried by FP64 FMA (primes < 2^47), u64 ciphertexts in and out",
                                      2: "u64 pseudo-Mersenne butterflies (primes 2^b - delta, <= 55 bits) + Shoup line products, fused row / column kernels" if pm == 1
                                         else "u64 Shoup modular arithmetic, fused row / column kernels",
                                      0: "u64 Shoup modular arithmetic, general three-launch path"}[path]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernels + " (the launches of fhe_dct8x8_quant, all waves of one step; HIP events on the launch stream)",
                         "algorithmic_bytes_per_launch": B * bytes_per_block, "algorithmic_bytes_per_block": bytes_per_block,
                         "ms_per_launch": dev_ms_per_step},
            "verified_bit_exact_vs_oracle": verified, "output_digest": "%016x" % digest_all,
        }
        if world > 1:
            # per GPU above (rank 0's own events); the whole job beside it: every rank's algorithmic bytes over ITS device time, summed,
            # against N x the per-GPU peak
            per_rank = [B * bytes_per_block / (ms * 1e-3) / 1e9 for ms in rank_dev_ms]
            res["roofline"]["per_gpu"] = "rank 0's launches; `aggregate` sums every rank"
            res["roofline"]["aggregate"] = {"achieved": sum(per_rank), "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": sum(per_rank) / (HBM_PEAK_GBS * world),
                                            "achieved_per_rank": per_rank, "device_ms_per_step_per_rank": rank_dev_ms}
        if n1 is not None:
            res["weak_scaling"] = {
                "n1_blocks_per_s": n1["blocks_per_s"], "n1_ms_per_step": n1["ms_per_step"], "n1_device_ms_per_step": n1["device_ms_per_step"],
                "per_gpu_blocks_per_s": value / world, "efficiency": value / (world * n1["blocks_per_s"]),
                "how": "rank 0 alone ran the same %d steps of %d blocks right before the all-rank region while the other ranks waited at a barrier; "
                       "efficiency = value / (n_gpus x n1_blocks_per_s)" % (args.steps, B) +
                       ("; TEST MODE: the ranks share %d device(s), so the efficiency is ~1/ranks-per-device by construction" % torch.cuda.device_count() if shared_devices else "")}
        # Which resource the pair really runs against: the counters (profiles/*_counters_summary.txt) show VALU issue, not HBM --
        # `roofline` above stays the HBM figure BASELINE.json asks for, this object is the issue-side one, from tracked files
        names = {1: ["k_dct_rows", "k_dct_cols"], 2: ["k_dct_rows_u64", "k_dct_cols_u64"]}.get(path)
        if names and ctx.n == 4096 and args.preset in ("P4096", "SEAL23_4096"):
            res["issue_roofline"] = issue_roofline(names, B, dev_ms_per_step)
            ir = res["issue_roofline"]
            if ir and ir.get("frac") and ir["frac"] > res["roofline"]["frac"]:
                # what the counters say binds the pair; achieved / peak / frac stay the HBM figures BASELINE.json's metric asks for
                res["roofline"]["bound"] = "valu-issue"
                res["roofline"]["frac_of"] = "hbm peak (the metric's roofline; the binding resource is VALU issue: issue_roofline.frac %.2f, %.2f at the nominal clock)" % (
                    ir["frac"], ir["frac_at_nominal_clock"])
            res["roofline"]["limiter"] = ("valu-issue at the package power limit (see issue_roofline; 1.37 kW and sclk 2.13-2.20 GHz measured under this pair, "
                                          "profiles/r04_power_clocks_headline.txt, again in round 6: profiles/r06_power_clocks_headline.txt); the HBM fraction is reported because BASELINE.json's metric asks for it")
        # SEAL's own CPU path "in the same run" (north_star): rank 0's host, every N; the other ranks wait at the barrier below
        if args.cpu_blocks > 0 and args.preset == "P4096":
            res["cpu_baseline"] = cpu_baseline(args.cpu_blocks)
            try:
                res["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
            except Exception as exc:      # the extra field must never break the bench line
                res["cpu_baseline_all_cores"] = {"error": str(exc)}

    # ---- the gather legs: the same step WITH the final ciphertext gather, both ways, after `value` was measured -----------------------
    broken = False
    if legs:
        gsteps = max(1, args.gather_steps)
        limit = float(os.environ.get("FHE_BENCH_GATHER_TIMEOUT", "150"))
        link_blocks = XGMI_LINK_GBS * 1e9 / out_bytes_per_block
        gobj = {"steps": gsteps, "wave_blocks": wave, "output_bytes_per_block": out_bytes_per_block,
                "note": "`value` is the compute-only figure (outputs stay sharded in HBM); these are the same step with the final ciphertext gather "
                        "(SURVEY.md 8e: with and without the gather), measured after it in the same process"}
        if res is not None:
            res["gather"] = gobj

        def run_leg(name, fn, finish, check=None):
            """one untimed + gsteps timed steps (+ the leg's verification) under a watchdog; a leg that fails or hangs is reported as such
            and ends the job cleanly -- every collective of the leg sits inside the watchdog's window"""
            nonlocal broken
            if broken:
                gobj[name] = {"error": "skipped: an earlier leg failed"}
                return
            torch.cuda.synchronize()

            def give_up():
                if res is not None:
                    gobj[name] = {"error": "no completion within %.0f s (watchdog): the leg was abandoned, the figures above it stand" % limit}
                    print(json.dumps(res), flush=True)
            dog = Watchdog(limit, give_up)
            try:
                fn()
                w_all, _, _ = timed(fn, gsteps)
                gobj[name] = finish(w_all / gsteps)
                if check is not None:
                    gobj[name].update(check())
            except Exception as exc:                            # the other ranks may be inside a transfer: no further collectives
                gobj[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                broken = True
            finally:
                dog.cancel()

        def wave_done(sec):
            dg = int(wave_digests.cpu().numpy().view(np.uint64).sum(dtype=np.uint64)) if rank == 0 else 0
            into_root = (world - 1) * B * out_bytes_per_block / sec / 1e9
            return {"blocks_per_s": B * world / sec, "ms_per_step": sec * 1e3, "fraction_of_compute_only": (B * world / sec) / (res["value"] if res else 1.0),
                    "xgmi_GB_per_s_into_root": into_root, "peers": world - 1,
                    "link_ceiling_blocks_per_s_per_peer": link_blocks,
                    "root_ceiling_blocks_per_s": (world - 1) * link_blocks + (res["value"] / world if res else 0.0),
                    "digest_matches_compute_only": (dg == digest_all) if rank == 0 else None,
                    "how": "every %d-block wave of every peer goes to rank 0 by RCCL send/recv (one xGMI link per peer, ~%.0f GB/s each: a peer can ship "
                           "~%.1f k blocks/s of %.0f MiB outputs while it computes ~%.0f k), overlapped with the next wave's compute; rank 0 drains every wave by digest.  "
                           "The ceiling of this consumer is the links into ONE root, not a defect of the overlap" % (
                               wave, XGMI_LINK_GBS, link_blocks / 1e3, out_bytes_per_block / 2 ** 20, (res["value"] / world / 1e3) if res else 0.0) +
                           ("; NOT RCCL (%s): the transfer is staged through pinned host memory" % backend_label(backend, requested_backend, shared_devices) if backend != "nccl" else "")}

        def local_done(sec):
            per_gpu = B * out_bytes_per_block / sec / 1e9
            return {"blocks_per_s": B * world / sec, "ms_per_step": sec * 1e3, "fraction_of_compute_only": (B * world / sec) / (res["value"] if res else 1.0),
                    "pcie_GB_per_s_per_gpu": per_gpu, "pcie_GB_per_s_aggregate": per_gpu * world,
                    "link_ceiling_blocks_per_s_per_gpu": PCIE_GBS * 1e9 / out_bytes_per_block,
                    "how": "every rank copies ITS OWN %d-block waves to page-locked host memory on a side stream (its own PCIe Gen5 x16 link, ~%.0f GB/s nominal), "
                           "overlapped with the next wave's compute; no inter-GPU traffic, so this consumer scales with the GPU count" % (wave, PCIE_GBS)}

        def local_check():
            dd, ih = verify_local_drain()
            ok_all = torch.tensor([1 if dd == ih else 0], dtype=torch.int64, device=coll_dev)
            if dist is not None:
                dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
            return {"drained_bytes_equal_in_hbm_result_on_every_rank": bool(ok_all.item())}

        barrier()                       # rank 0 comes from its CPU baseline: every rank arms a leg's watchdog at the same moment
        run_leg("local", step_local, local_done, local_check)
        run_leg("wave", step_wave, wave_done)

    if rank == 0:
        print(json.dumps(res), flush=True)
    if broken:                          # a failed leg may have left a peer inside a transfer: leave without another collective
        sys.stdout.flush()
        os._exit(0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
