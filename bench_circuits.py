#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): BASELINE.json configs[2] and configs[3].

  python bench_circuits.py resize [--pixels P]   bicubic 128x128 -> 64x64 via the Cubic circuit of
                                                 homo/fhe_resize.h:143-305, n=8192 (P8192), one channel
                                                 batch of P output pixels per launch sequence
  python bench_circuits.py decode                approximated_step (homo/fhe_decode.h:202-242), W*H=16,
                                                 degree 12, n=8192: one run

Inputs are synthetic random-residue ciphertexts; the server-side encryptions of the reference's
circuits (fractional offsets, Enc(0)) are inputs.  Prints one JSON line per workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def resize(args):
    import torch
    import fhip_amd as fhe
    ctx = fhe.SEALContext.preset(args.preset)
    ev = fhe.Evaluator(ctx)
    pc = fhe.circuits.PlainCache(ctx)
    W = H = args.src
    w = h = args.dst
    taps, fx, fy = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
    # one colour channel of the source image, resident in HBM
    pixels = ctx.random_ct(W * H, size=2, seed=fhe.SEED)
    n_out = w * h
    P = min(args.pixels, n_out)
    xf = ctx.random_ct(P, size=2, seed=11)     # Enc(frac(u)), Enc(frac(v)) are inputs
    yf = ctx.random_ct(P, size=2, seed=12)
    torch.cuda.synchronize()
    # warm-up (builds BEHZ tables, caches plaintexts)
    fhe.circuits.sample_bicubic(ev, pc, pixels, taps[:min(P, 8)], xf[:min(P, 8)].contiguous(), yf[:min(P, 8)].contiguous())
    torch.cuda.synchronize()
    passes = []
    for _ in range(1 if args.max_pixels else 2):      # the first full-size pass sizes the allocator's pools (scratch of several GiB); the second is the steady state
        t0 = time.perf_counter()
        done = 0
        for s in range(0, n_out, P):
            e = min(s + P, n_out)
            out = fhe.circuits.sample_bicubic(ev, pc, pixels, taps[s:e], xf[: e - s].contiguous(), yf[: e - s].contiguous())
            done += e - s
            if args.max_pixels and done >= args.max_pixels:
                break
        torch.cuda.synchronize()
        passes.append(time.perf_counter() - t0)
    dt = passes[-1]
    res = {"workload": "bicubic resize %dx%d -> %dx%d, one channel, %s (n=%d, k=%d)" % (W, H, w, h, args.preset, ctx.n, ctx.k),
           "output_pixels": done, "seconds": dt, "pixels_per_s": done / dt, "cubic_calls_per_s": 5 * done / dt,
           "out_size": int(out.shape[-3]), "batch_pixels": P, "first_pass_seconds": passes[0]}
    if args.shared:
        # SURVEY.md 8(d) config 3 input convention: one offset ciphertext per distinct fractional value, i.e. per output
        # column / row; every repeated ring element (row Cubics of overlapping windows, squares, prepared operands) is
        # then formed once -- bit-identical to the per-pixel evaluation with those ciphertexts (tests/test_gpu_configs.py)
        xc, yc = ctx.random_ct(w, size=2, seed=11), ctx.random_ct(h, size=2, seed=12)
        fhe.circuits.resize_bicubic_shared(ev, pc, pixels[: 16 * W], W, 16, w, 4, xc, yc[:4].contiguous(), batch=P)       # warm-up
        torch.cuda.synchronize()
        for _ in range(2):            # the first full-size pass sizes the allocator's pools; the second is the steady state
            sink = []
            t0 = time.perf_counter()
            fhe.circuits.resize_bicubic_shared(ev, pc, pixels, W, H, w, h, xc, yc, batch=P, consume=lambda first, t: sink.append(int(t.shape[0])))
            torch.cuda.synchronize()
            sdt = time.perf_counter() - t0
            assert sum(sink) == n_out
        rows_touched = len({taps[y * w][4 * j] // W for y in range(h) for j in range(4)})
        res["shared_offsets"] = {"seconds": sdt, "pixels_per_s": n_out / sdt, "row_cubics": rows_touched * w, "column_cubics": n_out,
                                 "note": "one offset ciphertext per output column / row; repeated row Cubics, squares and prepared operands formed once"}
    if args.cpu_pixels:
        from oracle import oracle as om
        orc = om.Oracle.preset(args.preset)
        hp = fhe.to_host(pixels[: 16])
        t = fhe.to_host(xf[0])
        t0 = time.perf_counter()
        for _ in range(args.cpu_pixels):
            cols = [orc.cubic(hp[4 * r], hp[4 * r + 1], hp[4 * r + 2], hp[4 * r + 3], t) for r in range(4)]
            orc.cubic(cols[0], cols[1], cols[2], cols[3], t)
        cdt = time.perf_counter() - t0
        res["cpu_oracle_pixels_per_s"] = args.cpu_pixels / cdt
    print(json.dumps(res), flush=True)


def decode(args):
    import torch
    import fhip_amd as fhe
    ctx = fhe.SEALContext.preset(args.preset)
    ev = fhe.Evaluator(ctx)
    pc = fhe.circuits.PlainCache(ctx)
    amp, idx, cnt = (ctx.random_ct(1, size=2, seed=900 + i) for i in range(3))

    # the Enc(0) accumulators are inputs: generated before the timed region
    bank = {(i, j, w): ctx.random_ct(1, size=2, seed=1000 + 100 * i + 10 * j + (w == "cos"))
            for i in range(max(args.positions, 1)) for j in range(1, max(args.degree, 1) + 1) for w in ("sin", "cos")}

    def zeros(i, j, which):
        return bank[(i, j, which)]

    # warm-up at full size: scratch buffers and the allocator cache reach their steady state
    fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, order=64, degree=args.degree, delta=0.5, width=args.positions, height=1, zeros=zeros)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run = fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, order=64, degree=args.degree, delta=0.5,
                                         width=args.positions, height=1, zeros=zeros)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "approximated_step W*H=%d degree=%d, %s (n=%d, k=%d)" % (args.positions, args.degree, args.preset, ctx.n, ctx.k),
                      "seconds": dt, "steps_per_s": 1 / dt, "out_size": int(run[0].shape[-3]), "outputs": len(run)}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["resize", "decode"])
    ap.add_argument("--preset", default="P8192")
    ap.add_argument("--src", type=int, default=128)
    ap.add_argument("--dst", type=int, default=64)
    ap.add_argument("--pixels", type=int, default=256)
    ap.add_argument("--max-pixels", type=int, default=0)
    ap.add_argument("--cpu-pixels", type=int, default=0)
    ap.add_argument("--shared", action="store_true", help="resize: also time the shared-offset form (one offset ciphertext per output column / row)")
    ap.add_argument("--degree", type=int, default=12)
    ap.add_argument("--positions", type=int, default=16)
    a = ap.parse_args()
    (resize if a.workload == "resize" else decode)(a)
