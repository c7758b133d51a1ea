#!/usr/bin/env python3
"""End-to-end rate of the streaming server_resize loop (homo/fhe_resize.h:308-392 over a ciphertext stream):
file -> pinned host -> HBM (sliding row window) -> batched SampleBicubic / SampleLinear -> pinned host -> file.
Files live in /dev/shm.  The circuit's server-side encryptions are pre-made ciphertexts (inputs).
Prints one JSON line (NOT the bench.py metric: I/O-inclusive)."""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="P8192")
ap.add_argument("--src", type=int, default=48)
ap.add_argument("--dst", type=int, default=24)
ap.add_argument("--bilinear", action="store_true")
ap.add_argument("--rows", type=int, default=4)
ap.add_argument("--dir", default="/dev/shm")
a = ap.parse_args()
ctx = fhe.SEALContext.preset(a.preset)
fin, fout = os.path.join(a.dir, "fhe_rs_in.ct"), os.path.join(a.dir, "fhe_rs_out.ct")
rng = np.random.default_rng(1)
one = np.stack([rng.integers(0, q, size=(2, ctx.n), dtype=np.uint64) for q in ctx.q], axis=1)   # [2, k, n]
with open(fin, "wb") as f:
    for _ in range(a.src * a.src * 3):
        fhe.server.write_ciphertext(f, one)
bank = ctx.random_ct(a.rows * a.dst * 2, size=2, seed=5)


def fractions(values):
    return bank[:len(values)]


try:
    t0 = time.time()
    done = fhe.server.server_resize(ctx, fin, fout, a.src, a.src, a.dst, a.dst, not a.bilinear, fractions, rows_per_step=a.rows)
    torch.cuda.synchronize()
    dt = time.time() - t0
    in_bytes, out_bytes = os.path.getsize(fin), os.path.getsize(fout)
finally:
    for p in (fin, fout):
        if os.path.exists(p):
            os.remove(p)
print(json.dumps({"workload": "server_resize stream %dx%d -> %dx%d %s, three channels, %s" % (a.src, a.src, a.dst, a.dst, "bilinear" if a.bilinear else "bicubic", a.preset),
                  "output_pixels": done, "rows_per_step": a.rows, "seconds": dt, "pixels_per_s": done / dt,
                  "stream_GB_in": in_bytes / 1e9, "stream_GB_out": out_bytes / 1e9, "files": a.dir}))
