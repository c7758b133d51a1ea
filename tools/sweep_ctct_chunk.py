#!/usr/bin/env python3
"""ct x ct in chunks: does a launch set small enough for its intermediates to stay in L2 / MALL (14.6 MB of
traffic per 2x2 product at n = 8192) run faster per product than one launch set over the whole batch?
Prints products/s per chunk size for a fixed total (experiment; profiles/EXPERIMENTS.md section 2)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

preset = sys.argv[1] if len(sys.argv) > 1 else "P8192"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = fhe.SEALContext.preset(preset)
ev = fhe.Evaluator(ctx)
a, b = ctx.random_ct(total, seed=1), ctx.random_ct(total, seed=2)
ref = ev.multiply(a, b)
for chunk in (4, 8, 16, 32, 64, 128, 256, 512, 1024):
    if chunk > total:
        break
    outs = None

    def run():
        global outs
        outs = [ev.multiply(a[s:s + chunk], b[s:s + chunk]) for s in range(0, total, chunk)]
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    same = bool(torch.equal(torch.cat(outs), ref))
    print(json.dumps({"preset": preset, "total": total, "chunk": chunk, "ms": round(ms, 3), "products_per_s": round(total / ms * 1e3), "same_bits": same}), flush=True)
