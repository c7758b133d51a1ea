#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): BASELINE.json configs[2] and configs[3], in the BENCH schema.

  python bench_circuits.py resize [--shared] [--gpus N]   bicubic 128x128 -> 64x64 via the Cubic circuit of
                                                         homo/fhe_resize.h:143-392, n=8192 (P8192), one channel
  python bench_circuits.py decode [--gpus N]             approximated_step (homo/fhe_decode.h:202-242), W*H=16,
                                                         degree 12, n=8192: one run
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench_circuits.py <workload> --gpus N ...
  (`python bench_circuits.py <workload> --gpus N` without a launcher starts the N ranks itself: bench.ensure_world)

Inputs are synthetic random-residue ciphertexts; the server-side encryptions of the reference's circuits (fractional
offsets, Enc(0)) are inputs.  Multi-GPU (one process per GPU, RCCL for the barrier, the
max-over-ranks time and the digest all-reduce only): resize shards the destination ROWS (parallel.row_range; every rank
generates / loads its source rows +- the sampler's halo), decode shards the output POSITIONS of the run; both are fixed-size
jobs, so the scaling is "strong".  Rank 0 prints ONE JSON line with `roofline` (HBM on algorithmic bytes: inputs read once,
outputs written once) and, when asked for (--cpu-pixels / --cpu-terms), `cpu_baseline` (the oracle on one host core, on a
bounded sample).  `output_digest` is the position-dependent digest of everything produced: equal for every GPU count.

  --relin DBC   the RELINEARISED mode (include/fhe_circuits.h fhe_circuits_create_relin): the same
                Evaluator call sequences with evaluator.relinearize after every multiply / square, decomposition bit count DBC
                (the reference's unused DBC is 30, homo/fhe_image.h:28).  NOT the reference's ciphertext bits: outputs have 2
                polynomials instead of 6 / 22; `config.mode` says so and the digests differ from the default lines' by design.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver; before torch / HIP load

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0
M64 = (1 << 64) - 1


def _dist_setup(args):
    import torch
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench_circuits.py needs a HIP device (no CPU path exists)")
    if world != args.gpus:            # bench.ensure_world has started the ranks or refused already; this guards direct calls of resize() / decode()
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # FHE_BENCH_BACKEND=gloo (tests only, as in bench.py): ranks share the devices there are, collectives on host tensors
    backend = os.environ.get("FHE_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        # RCCL after a bounded probe in a child process, else the same job on gloo (bench.open_process_group); the line's `collective_backend` says which
        from bench import backend_label, open_process_group
        dist, used, probe = open_process_group(world, local, backend)
        _DIST_INFO.update(collective_backend=backend_label(used, backend, backend == "gloo" and world > torch.cuda.device_count()), rccl_probe=probe)
    return rank, world, local, dist


_DIST_INFO = {}                         # what _dist_setup found out; _strong_scaling adds it to the N > 1 lines


def _timed(fn, dist, reps=1):
    """barrier + synchronize on both sides, HIP events on the launch stream, max over ranks (seconds, device ms per rep)"""
    import torch

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        res = fn()
    e1.record()
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    return res, wall / reps, e0.elapsed_time(e1) / reps


def _server_side_encryptions(fhe, ctx, count, values, wall, units, what):
    """The encryptions the reference's function makes for this workload (inside SampleBicubic: homo/fhe_resize.h:262,266; inside
    homomorphic_sin / cos: homo/fhe_decode.h:54,134), which the timed region above takes as resident inputs -- timed
    here as the device batch a server runs (keys.DeviceEncryptor: fhe_frac_encode_batch + fhe_encrypt_batch), so the line also says
    what the job costs with them.  values=None: encryptions of encode(0)."""
    import torch
    der = fhe.DeviceEncryptor(ctx, fhe.KeyGenerator(ctx).public_key())
    run = (lambda: der.encrypt_values(values)) if values is not None else (lambda: der.encrypt_zeros(count))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"count": count, "what": what, "ms": dt * 1e3, "share_of_step": dt / wall, "value_including_them": units / (wall + dt),
            "note": "not in `value` (inputs resident, SURVEY 8d); one device batch, wall time incl. the host call; 1.2 ms EACH with the host sampler of rounds 2-4"}


def _cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _all_source_hash():
    """every file of csrc/ (the hash tools/collect_traffic.py circuits ties its record to)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def _traffic(workload, units):
    """HBM bytes of one job from profiles/pmc_traffic_circuits.json (FETCH_SIZE + WRITE_SIZE of every launch of the circuit, per
    output pixel / per run; tools/collect_traffic.py circuits), quoted only for the sources it was measured on"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_circuits.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_traffic_circuits.json is missing; run tools/collect_traffic.py circuits"
    with open(path) as f:
        tj = json.load(f)
    if tj.get("kernel_source_hash") != _all_source_hash():
        return None, "profiles/pmc_traffic_circuits.json was measured on other kernel sources (%s); re-run tools/collect_traffic.py circuits" % tj.get("kernel_source_hash")
    rec = tj.get(workload)
    if not rec:
        return None, "no record for %s" % workload
    return rec["hbm_bytes_per_unit"] * units, tj.get("source")


def _roofline(alg_bytes, dev_ms, kernels, workload=None, units=0, issue=None):
    """HBM roofline on algorithmic bytes (the figure BASELINE.json's metric asks for) -- `bound` says what the counters say binds
    the launch sequence (the issue-side object `issue`, bench.issue_roofline_workload) when they say something else"""
    achieved = alg_bytes / (dev_ms * 1e-3) / 1e9
    traffic, src = _traffic(workload, units) if workload else (None, None)
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
           "algorithmic_bytes_per_launch": alg_bytes, "ms_per_launch": dev_ms, "kernel": kernels}
    if issue and issue.get("frac") and issue["frac"] > out["frac"]:
        out["bound"] = "valu-issue"
        out["frac_of"] = ("hbm peak (the metric's roofline); the launch sequence is bound by VALU issue: launch-time-weighted issue fraction %.2f at the clock the "
                          "counters saw, %.2f at the nominal %.1f GHz (issue_roofline)" % (issue["frac"], issue["frac_at_nominal_clock"], issue["nominal_clock_ghz"]))
    return out


def _relin(args, fhe, ctx):
    """(evk, dbc[, "cubic"]) for --relin: keys of a seeded key pair (the inputs are random residues, so any well-formed keys do)"""
    if not args.relin:
        return None
    kg = fhe.KeyGenerator(ctx, seed=1)
    if args.relin_placement == "cubic":        # keys for s^2 and s^3: one evaluator.relinearize takes a Cubic's size-4 result to 2
        return (kg.generate_evaluation_keys(args.relin, 2).contiguous(), args.relin, "cubic")
    if args.relin_placement == "sample":       # keys for s^2 .. s^5: one evaluator.relinearize takes an output pixel's size-6 ciphertext to 2
        return (kg.generate_evaluation_keys(args.relin, 4).contiguous(), args.relin, "sample")
    return (kg.generate_evaluation_keys(args.relin).contiguous(), args.relin)


def _mode(args):
    if args.relin and args.relin_placement == "sample":
        return ("the reference's SampleBicubic sequence unchanged (sizes 2 -> 4 -> 6) + ONE evaluator.relinearize of every output pixel (keys for s^2 .. s^5), "
                "dbc = %d (NOT the reference's bits: the reference never relinearises)" % args.relin)
    if args.relin and args.relin_placement == "cubic":
        return ("the reference's Cubic sequence unchanged + ONE evaluator.relinearize of its size-4 result (keys for s^2, s^3: two key switches per Cubic), "
                "dbc = %d (NOT the reference's bits: the reference never relinearises)" % args.relin)
    if args.relin:
        return "relinearised after every multiply / square, dbc = %d (NOT the reference's bits: the reference never relinearises)" % args.relin
    return "reference (no relinearisation: homo/fhe_resize.h:174-179, homo/fhe_decode.h:67-98,235,239)"


def _wl_suffix(args):
    return ("_relin%d%s" % (args.relin, {"cubic": "_cubic", "sample": "_sample"}.get(args.relin_placement, ""))) if args.relin else ""


def _oracle_for_mode(args, om, orc):
    """the checker object whose op-by-op composition defines the mode that is being timed (cpu_baseline of the --relin lines)"""
    if not args.relin:
        return orc
    sk, _ = orc.keygen(1)
    if args.relin_placement == "cubic":
        return om.TailRelinOracle(orc, orc.evk_gen_powers(sk, dbc=args.relin, count=2), args.relin)
    if args.relin_placement == "sample":
        return om.SampleRelinOracle(orc, orc.evk_gen_powers(sk, dbc=args.relin, count=4), args.relin)
    return om.RelinOracle(orc, orc.evk_gen(sk, dbc=args.relin), args.relin)


def _strong_scaling(n1_seconds, wall, world, what):
    return {"n1_seconds": n1_seconds, "speedup": n1_seconds / wall, "efficiency": n1_seconds / wall / world,
            "how": "rank 0 alone ran the WHOLE job (%s) once, after a warm-up pass, while the other ranks waited at a barrier; speedup = that time / the "
                   "all-rank time of the same job" % what}


def resize(args):
    import numpy as np
    import torch
    import fhip_amd as fhe
    rank, world, local, dist = _dist_setup(args)
    ctx = fhe.SEALContext.preset(args.preset, device=local)
    ev = fhe.Evaluator(ctx)
    pc = fhe.circuits.PlainCache(ctx)
    relin = _relin(args, fhe, ctx)
    so = fhe.circuits.circuits_of(pc, relin).out_size(fhe.circuits.SAMPLE_BICUBIC)
    W = H = args.src
    w = h = args.dst
    ctw = 2 * ctx.k * ctx.n                                     # words of a ct(2)
    y0, y1 = fhe.parallel.row_range(rank, world, h)
    if y1 == y0:
        raise SystemExit("more GPUs than destination rows")
    if args.max_pixels:                                          # profiling runs: only the first rows of the (single) shard
        if world > 1:
            raise SystemExit("--max-pixels is a single-GPU profiling switch")
        y1 = min(y1, y0 + (args.max_pixels + w - 1) // w)
    taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
    words = so * ctx.k * ctx.n

    def build(y0, y1):
        """the job of destination rows [y0, y1): its resident inputs (rows +- halo, offsets) and the callable"""
        first, count = fhe.parallel.source_rows(H, h, y0, y1, True)
        # this rank's rows +- halo of ONE colour channel, resident in HBM; the global pixel index seeds the bytes, so any GPU count sees the same image
        pixels = ctx.random_ct(count * W, size=2, seed=fhe.SEED, first_index=first * W * ctw)
        my_taps = (taps[y0 * w:y1 * w].astype(np.int64) - first * W).astype(np.uint32)
        n_mine = (y1 - y0) * w
        n_out = n_mine if args.max_pixels else w * h
        P = min(args.pixels, n_mine)
        if args.shared:
            # SURVEY.md 8(d) configs[2] input convention: one offset ciphertext per distinct fractional value, i.e. per output column / row
            xc = ctx.random_ct(w, size=2, seed=11)
            yc = ctx.random_ct(y1 - y0, size=2, seed=12, first_index=y0 * ctw)
            acc = torch.zeros(1, dtype=torch.int64, device=ctx.device)

            def consume(first_px, t):
                part = torch.zeros(1, dtype=torch.int64, device=ctx.device)
                ctx.digest_into(t, part, index0=first_px * words)
                acc.add_(part)

            def job(digest=False):          # the timed passes hand the bands to a no-op consumer; one untimed pass digests them
                acc.zero_()
                fhe.circuits.resize_bicubic_shared(ev, pc, pixels, W, H, w, h, xc, yc, batch=P, consume=consume if digest else (lambda first_px, t: None),
                                                   rows=(y0, y1), src_rows=(first, count), relin=relin)
                return None
            alg = (W * (H if not args.max_pixels else count) + w + h) * ctw * 8 + n_out * words * 8
            form = "one offset ciphertext per output column / row; repeated row Cubics, squares and prepared operands formed once"
        else:
            # one offset ciphertext pair per output pixel, as the reference's server draws them (homo/fhe_resize.h:262,266)
            xf = ctx.random_ct(n_mine, size=2, seed=11, first_index=y0 * w * ctw)
            yf = ctx.random_ct(n_mine, size=2, seed=12, first_index=y0 * w * ctw)
            acc = torch.zeros(1, dtype=torch.int64, device=ctx.device)

            def job(digest=False):
                acc.zero_()
                for s in range(0, n_mine, P):
                    e = min(s + P, n_mine)
                    out = fhe.circuits.sample_bicubic(ev, pc, pixels, my_taps[s:e], xf[s:e].contiguous(), yf[s:e].contiguous(), relin=relin)
                    if digest:
                        part = torch.zeros(1, dtype=torch.int64, device=ctx.device)
                        ctx.digest_into(out, part, index0=(y0 * w + s) * words)
                        acc.add_(part)
                return None
            alg = (W * (H if not args.max_pixels else count) + 2 * n_out) * ctw * 8 + n_out * words * 8
            form = "one offset ciphertext pair per output pixel (five Cubic evaluations per pixel)"
        return {"job": job, "acc": acc, "n_mine": n_mine, "n_out": n_out, "P": P, "alg": alg, "form": form, "pixels": pixels}

    J = build(y0, y1)
    job, acc, n_mine, n_out, P, alg, form, pixels = (J[k] for k in ("job", "acc", "n_mine", "n_out", "P", "alg", "form", "pixels"))
    job()                                                       # warm-up at full size: ct x ct tables, cached plaintexts, the allocator's pools
    _, first_pass, _ = _timed(job, dist)
    _, wall, dev_ms = _timed(job, dist)
    job(digest=True)                                            # untimed: the position-dependent digest of everything produced
    torch.cuda.synchronize()
    digest = fhe.parallel.combine_digests(int(acc.cpu().numpy().view(np.uint64)[0]))
    n1_seconds = None
    if world > 1:                                               # in-run N = 1 leg of the fixed-size job: rank 0 alone over ALL destination rows
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            del J, job, pixels
            whole = build(0, h)
            whole["job"]()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            whole["job"]()
            torch.cuda.synchronize()
            n1_seconds = time.perf_counter() - t0
            del whole
        dist.barrier()
    if rank == 0:
        from bench import issue_roofline_workload
        wl = ("resize_shared" if args.shared else "resize") + _wl_suffix(args)
        issue = issue_roofline_workload(wl) if args.preset == "P8192" else None
        res = {"metric": "bicubic-resized output pixels/sec (one colour channel)", "value": n_out / wall, "unit": "pixels/s", "n_gpus": world,
               "steps": 1, "warmup": 2, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u64", "data": "synthetic",
               "config": {"workload": "bicubic resize %dx%d -> %dx%d via the Cubic circuit, one channel, %s (n=%d, %d coeff moduli)" % (W, H, w, h, args.preset, ctx.n, ctx.k),
                          "mode": _mode(args), "offsets": form, "batch_pixels": P, "sharding": "destination rows x%d, source rows +- halo per rank, no data-path collective" % world},
               "seconds": wall, "first_pass_seconds": first_pass, "cubic_calls_per_s": 5 * n_out / wall, "out_size": so,
               "issue_roofline": issue,
               "roofline": _roofline(alg // world, dev_ms, "k_cubic_coeffs_g, k_behz_*_pm, k_ntt_fwd_pm" + (", k_relin_*_pm" if relin else ""),
                                     wl, n_mine if args.preset == "P8192" else 0, issue),
               "job_executions": 4, "units_per_job": n_mine,
               "output_digest": "%016x" % digest}
        if n1_seconds is not None:
            res["strong_scaling"] = _strong_scaling(n1_seconds, wall, world, "all %d destination rows" % h)
        res.update(_DIST_INFO)
        if world == 1 and not args.max_pixels:
            _, fxs, fys = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
            vals = ([fxs[x] for x in range(w)] + [fys[y * w] for y in range(h)]) if args.shared else [v for pair in zip(fxs, fys) for v in pair]
            res["server_side_encryptions"] = _server_side_encryptions(fhe, ctx, len(vals), vals, wall, n_out,
                                                                      "Enc(encode(frac)) per output column and row" if args.shared else
                                                                      "Enc(encode(frac(x))), Enc(encode(frac(y))) per output pixel (shared by the three channels; all charged to this one)")
        if args.cpu_pixels:
            from oracle import oracle as om
            orc = om.Oracle.preset(args.preset)
            hp = fhe.to_host(ctx.random_ct(16, size=2, seed=fhe.SEED))
            t = fhe.to_host(ctx.random_ct(1, size=2, seed=11))[0]
            mode_orc = _oracle_for_mode(args, om, orc)           # --relin: the op-by-op composition that defines the mode (RelinOracle / TailRelinOracle)
            t0 = time.perf_counter()
            for _ in range(args.cpu_pixels):
                if relin:
                    om.oracle_sample_bicubic_calls(mode_orc, list(hp), t, t)
                else:
                    cols = [orc.cubic(hp[4 * r], hp[4 * r + 1], hp[4 * r + 2], hp[4 * r + 3], t) for r in range(4)]
                    orc.cubic(cols[0], cols[1], cols[2], cols[3], t)
            cdt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": args.cpu_pixels / cdt, "unit": "pixels/s", "cores": 1, "kind": "port",
                                   "sample": "%d output pixels (5 Cubic each%s) of the same workload, oracle/libfhe_oracle.so op at a time, 1 thread of %d on %s"
                                             % (args.cpu_pixels, ", every Evaluator call of the relinearised mode's definition" if relin else "", os.cpu_count() or 0, _cpu_name())}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def decode(args):
    import numpy as np
    import torch
    import fhip_amd as fhe
    rank, world, local, dist = _dist_setup(args)
    ctx = fhe.SEALContext.preset(args.preset, device=local)
    ev = fhe.Evaluator(ctx)
    pc = fhe.circuits.PlainCache(ctx)
    relin = _relin(args, fhe, ctx)
    npos, degree = args.positions, args.degree
    p0, p1 = fhe.parallel.block_range(rank, world, npos)
    if p1 == p0:
        raise SystemExit("more GPUs than output positions")
    amp, idx, cnt = (ctx.random_ct(1, size=2, seed=900 + i) for i in range(3))
    ctw = 2 * ctx.k * ctx.n
    so = fhe.circuits.circuits_of(pc, relin).out_size(fhe.circuits.STEP, degree)

    def build(p0, p1):
        # the Enc(0) accumulators are inputs: [position][harmonic][sin, cos], seeded by the global position
        zeros = ctx.random_ct((p1 - p0) * degree * 2, size=2, seed=1000, first_index=p0 * degree * 2 * ctw).reshape(p1 - p0, degree, 2, 2, ctx.k, ctx.n) if degree else None

        def job():
            return fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, order=64, degree=degree, delta=0.5, width=npos, height=1, zeros=zeros, positions=(p0, p1),
                                                  relin=relin)
        return job, zeros
    job, zeros = build(p0, p1)
    job()                                                       # warm-up at full size: scratch buffers and the allocator cache reach their steady state
    run, wall, dev_ms = _timed(job, dist)
    out = torch.cat(run)
    digest = fhe.parallel.combine_digests(ctx.digest(out, index0=p0 * so * ctx.k * ctx.n))
    n1_seconds = None
    if world > 1:                                               # in-run N = 1 leg of the fixed-size job: rank 0 alone over ALL positions
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            whole, _z = build(0, npos)
            whole()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            whole()
            torch.cuda.synchronize()
            n1_seconds = time.perf_counter() - t0
            del whole, _z
        dist.barrier()
    if rank == 0:
        from bench import issue_roofline_workload
        wl = "decode" + _wl_suffix(args)
        issue = issue_roofline_workload(wl) if (args.preset == "P8192" and degree == 12) else None
        alg = (3 + npos * degree * 2) * ctw * 8 + npos * so * ctx.k * ctx.n * 8
        res = {"metric": "approximated_step runs/sec (all W*H output positions of one run)", "value": 1 / wall, "unit": "runs/s", "n_gpus": world,
               "steps": 1, "warmup": 1, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u64", "data": "synthetic",
               "config": {"workload": "approximated_step W*H=%d degree=%d, %s (n=%d, %d coeff moduli)" % (npos, degree, args.preset, ctx.n, ctx.k),
                          "mode": _mode(args), "sharding": "output positions x%d; run operands, offset chain and sine polynomials replicated; no data-path collective" % world},
               "seconds": wall, "steps_per_s": 1 / wall, "out_size": so, "outputs": npos,
               "issue_roofline": issue,
               "roofline": _roofline(alg // world, dev_ms, "k_behz_*_pm, k_ntt_fwd_pm, k_mulplain*, k_sum_inv_pm" + (", k_relin_*_pm" if relin else ""), wl,
                                     (p1 - p0) / npos if (args.preset == "P8192" and degree == 12) else 0, issue),
               "job_executions": 2, "units_per_job": p1 - p0,
               "output_digest": "%016x" % digest}
        if n1_seconds is not None:
            res["strong_scaling"] = _strong_scaling(n1_seconds, wall, world, "all %d output positions" % npos)
        res.update(_DIST_INFO)
        if world == 1 and degree:
            res["server_side_encryptions"] = _server_side_encryptions(fhe, ctx, npos * degree * 2, None, wall, 1, "Enc(encode(0)) per (position, harmonic) for homomorphic_sin and homomorphic_cos")
        if args.cpu_terms:
            from oracle import oracle as om
            orc = om.Oracle.preset(args.preset)
            h_amp, h_idx, h_cnt = (fhe.to_host(t)[0] for t in (amp, idx, cnt))
            hz = fhe.to_host(zeros[:1, :args.cpu_terms].reshape(-1, 2, ctx.k, ctx.n))
            t0 = time.perf_counter()
            om.oracle_approximated_step(_oracle_for_mode(args, om, orc), h_amp, h_idx, h_cnt, 64, args.cpu_terms, 0.5, 1, 1, lambda i, j, wh: hz[2 * (j - 1) + (wh == "cos")])
            cdt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": 1 / (cdt * npos * max(degree, 1) / args.cpu_terms), "unit": "runs/s", "cores": 1, "kind": "port",
                                   "sample": "one output position with %d of the %d harmonics (sin + cos Taylor polynomials and their 11 x 11 product each), scaled by positions x harmonics; "
                                             "oracle/libfhe_oracle.so op at a time, 1 thread of %d on %s" % (args.cpu_terms, degree, os.cpu_count() or 0, _cpu_name())}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["resize", "decode"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--preset", default="P8192")
    ap.add_argument("--src", type=int, default=128)
    ap.add_argument("--dst", type=int, default=64)
    ap.add_argument("--pixels", type=int, default=256, help="output pixels per launch sequence")
    ap.add_argument("--max-pixels", type=int, default=0, help="resize: stop after the rows holding this many output pixels (profiling runs)")
    ap.add_argument("--cpu-pixels", type=int, default=0, help="resize: CPU baseline sample (output pixels through the oracle; ~0.6 s each at n = 8192)")
    ap.add_argument("--cpu-terms", type=int, default=0, help="decode: CPU baseline sample (harmonics of one position through the oracle; ~5 s each at n = 8192)")
    ap.add_argument("--shared", action="store_true", help="resize: one offset ciphertext per output column / row (SURVEY.md 8d) instead of one pair per pixel")
    ap.add_argument("--degree", type=int, default=12)
    ap.add_argument("--positions", type=int, default=16)
    ap.add_argument("--relin", type=int, default=0, metavar="DBC", help="relinearised mode with this decomposition bit count (0 = the reference's mode)")
    ap.add_argument("--relin-placement", choices=["product", "cubic", "sample"], default="product",
                    help="product: evaluator.relinearize after every multiply / square (five key switches per Cubic); cubic (resize only): the reference's "
                         "Cubic unchanged and ONE relinearize of its size-4 result (two key switches, keys for s^2 and s^3) -- include/fhe_circuits.h FHE_RELIN_PER_CUBIC; "
                         "sample (resize only): SampleBicubic unchanged and ONE relinearize of every output pixel, 6 -> 2 (FHE_RELIN_PER_SAMPLE)")
    a = ap.parse_args()
    from bench import ensure_world
    ensure_world(a.gpus, os.path.abspath(__file__), sys.argv[1:])      # `python bench_circuits.py <workload> --gpus N` starts its N ranks itself
    (resize if a.workload == "resize" else decode)(a)
