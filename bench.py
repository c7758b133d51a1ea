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
                    help="wave: every wave of output ciphertexts is sent to rank 0 (RCCL send/recv over xGMI, overlapped with the "
                         "next wave's compute) and drained there by digest; local: every rank drains ITS OWN waves to pinned host "
                         "memory over its own PCIe link (parallel.LocalDrain; the consumer that scales with the GPU count); "
                         "none (default): outputs stay sharded in HBM (SURVEY.md 8e)")
    ap.add_argument("--gather-wave-blocks", type=int, default=64)
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
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    dist = None
    # FHE_BENCH_FORCE_DIST=1: a process group even for ONE rank -- the RCCL initialisation, barrier, all-reduce and all-gather of the
    # multi-GPU path execute (trivially) on a one-GPU box; tests/test_gpu_multi.py uses it so that the collective code has run on RCCL
    if world > 1 or os.environ.get("FHE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    ctx = fhe.SEALContext.preset(args.preset, device=local_rank)
    ev = fhe.Evaluator(ctx)
    plan = fhe.DctPlan(ctx, fhe.YQT)
    B = args.blocks
    words_per_block = 64 * 2 * ctx.k * ctx.n
    # global block index g = rank * B + b: any GPU count generates the same bytes for block g
    first_index = rank * B * words_per_block
    blocks = ctx.random_ct(B, 64, seed=fhe.SEED, first_index=first_index)
    out = torch.empty_like(blocks)
    torch.cuda.synchronize()

    gather = None
    if args.gather == "wave":
        if B % args.gather_wave_blocks:
            raise SystemExit("--blocks must be a multiple of --gather-wave-blocks")
        wave, n_waves = args.gather_wave_blocks, B // args.gather_wave_blocks
        # rank 0 drains every wave (its own and the peers') by digest: one u64 per (source rank, wave)
        wave_digests = torch.zeros(world * n_waves, dtype=torch.int64, device=blocks.device)

        def consume(src, w, t):
            ctx.digest_into(t.view(-1), wave_digests[src * n_waves + w:src * n_waves + w + 1],
                            index0=(src * B + w * wave) * words_per_block)
        if world > 1:
            gather = fhe.parallel.WaveGather((wave,) + tuple(blocks.shape[1:]), blocks.dtype, blocks.device, n_waves, consume=consume)
    local = None
    if args.gather == "local":
        if B % args.gather_wave_blocks:
            raise SystemExit("--blocks must be a multiple of --gather-wave-blocks")
        wave, n_waves = args.gather_wave_blocks, B // args.gather_wave_blocks
        drained = [0]

        def on_host(w, host_tensor):                            # where a per-GPU stream writer would take over
            drained[0] += host_tensor.numel() * 8
        local = fhe.parallel.LocalDrain((wave,) + tuple(blocks.shape[1:]), blocks.dtype, blocks.device, consume=on_host)

    def step():
        if args.gather == "local":
            for w in range(n_waves):
                buf = local.acquire()
                ev.dct8x8_quant(plan, blocks[w * wave:(w + 1) * wave], out=buf)
                local.commit(w)
            local.finish()
            local.reset()
            return
        if args.gather != "wave":
            ev.dct8x8_quant(plan, blocks, out=out)
            return
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

    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps ------------------------------------------------------------
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()                      # same stream the C ABI launches on (torch current stream)
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms_per_step = ev0.elapsed_time(ev1) / args.steps
    rank_ms, rccl_ranks = [wall / args.steps * 1e3], 1
    if dist is not None:
        tt = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # self-check of the multi-GPU run: every rank contributes 1 to a SUM all-reduce (= ranks RCCL really connected)
        # and its own per-step time to an all-gather, so the one JSON line shows the whole job
        ones = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(ones.item())
        mine = torch.tensor([wall / args.steps * 1e3], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) for t in every]
        wall = float(tt.item())

    # ---- verification: sampled blocks against the CPU oracle, digest over everything -----------------
    if args.gather == "local":
        # the timed path ends in page-locked HOST buffers: one more pass whose consumer brings every drained wave back and digests
        # it with its global index -- the bytes that really left over PCIe -- next to the digest of an in-HBM evaluation
        back = torch.empty((wave,) + tuple(blocks.shape[1:]), dtype=blocks.dtype, device=blocks.device)
        host_digests = torch.zeros(n_waves, dtype=torch.int64, device=blocks.device)

        def verify_on_host(w, host_tensor):
            back.copy_(host_tensor)
            ctx.digest_into(back.view(-1), host_digests[w:w + 1], index0=first_index + w * wave * words_per_block)
        local.consume = verify_on_host
        step()
        torch.cuda.synchronize()
        drained_digest = int(host_digests.cpu().numpy().view(np.uint64).sum(dtype=np.uint64))
        ev.dct8x8_quant(plan, blocks, out=out)
        in_hbm = ctx.digest(out.view(-1), index0=first_index)
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
            "rccl_ranks": rccl_ranks, "collective_backend": ("rccl" if backend == "nccl" else backend + " (test mode: ranks share devices)") if dist is not None else None,
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
                                          "profiles/r04_power_clocks_headline.txt); the HBM fraction is reported because BASELINE.json's metric asks for it")
        if world == 1 and args.cpu_blocks > 0 and args.preset == "P4096":
            res["cpu_baseline"] = cpu_baseline(args.cpu_blocks)
            try:
                res["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
            except Exception as exc:      # the extra field must never break the bench line
                res["cpu_baseline_all_cores"] = {"error": str(exc)}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
