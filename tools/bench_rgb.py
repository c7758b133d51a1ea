#!/usr/bin/env python3
"""Throughput of the fused rgb_to_ycc_fhe launch (homo/fhe_image.h:310-325) on resident ciphertexts:
pixels/s and algorithmic GB/s (3 ct in + 3 ct out per pixel).  Secondary measurement, not bench.py's."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe
px = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
preset = sys.argv[2] if len(sys.argv) > 2 else "P4096"
ctx = fhe.SEALContext.preset(preset)
ev = fhe.Evaluator(ctx)
r, g, b = (ctx.random_ct(px, 1, seed=fhe.SEED + i).reshape(px, 2, ctx.k, ctx.n) for i in range(3))
for _ in range(2):
    ev.rgb_to_ycc(r, g, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
steps = 5
e0.record()
for _ in range(steps):
    ev.rgb_to_ycc(r, g, b)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
by = 6 * 2 * ctx.k * ctx.n * 8
print(json.dumps({"workload": "rgb_to_ycc_fhe, %s (n=%d k=%d)" % (preset, ctx.n, ctx.k), "pixels": px, "ms": ms, "pixels_per_s": px / ms * 1e3,
                  "algorithmic_GB_per_s": px * by / ms / 1e6, "hbm_frac": px * by / ms / 1e6 / 8000}))
