import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fhip_amd as fhe
ctx = fhe.SEALContext.preset("P8192"); ev = fhe.Evaluator(ctx); pc = fhe.circuits.PlainCache(ctx)
W = H = 64; w = h = 32
pix = ctx.random_ct(W * H, size=2, seed=1)
taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
xf, yf = ctx.random_ct(256, size=2, seed=2), ctx.random_ct(256, size=2, seed=3)
t0 = time.time(); n = 0
while time.time() - t0 < 25:
    for s in range(0, w * h, 256):
        fhe.circuits.sample_bicubic(ev, pc, pix, taps[s:s + 256], xf, yf)
    torch.cuda.synchronize(); n += w * h
print("resize loop: %.0f output pixels/s over %.1f s" % (n / (time.time() - t0), time.time() - t0))
