#!/usr/bin/env python3
"""End-to-end rate of the streaming server_jpeg loop: ciphertext stream
file -> pinned host -> HBM -> rgb_to_ycc + DCT (fused kernels) -> pinned host -> file, colour blocks
(3 channels x 64 ciphertexts) at n = 4096, k = 3.  Files live in /dev/shm so that the number shows the
host-side loop + PCIe, not a disk.  Prints one JSON line (NOT the bench.py metric: I/O-inclusive)."""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=96)
ap.add_argument("--wave", type=int, default=16)
ap.add_argument("--dir", default="/dev/shm")
a = ap.parse_args()
ctx = fhe.SEALContext.preset("P4096") if hasattr(fhe.SEALContext, "preset") else fhe.SEALContext(4096, [0xffffee001, 0xffffc4001, 0x1ffffe0001], 1 << 14)
fin, fout = os.path.join(a.dir, "fhe_in.ct"), os.path.join(a.dir, "fhe_out.ct")
rng = np.random.default_rng(1)
one = np.stack([rng.integers(0, q, size=(2, ctx.n), dtype=np.uint64) for q in ctx.q], axis=1)   # [2, k, n]
with open(fin, "wb") as f:
    for _ in range(a.blocks * 3 * 64):
        fhe.server.write_ciphertext(f, one)
in_bytes = os.path.getsize(fin)
try:
    fhe.server.server_jpeg(ctx, fin, fout, min(a.blocks, a.wave), wave_blocks=a.wave)           # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    done = fhe.server.server_jpeg(ctx, fin, fout, a.blocks, wave_blocks=a.wave)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out_bytes = os.path.getsize(fout)
finally:
    for p in (fin, fout):
        if os.path.exists(p):
            os.remove(p)
print(json.dumps({"workload": "server_jpeg stream, colour blocks, n=4096 k=3", "blocks": done, "wave_blocks": a.wave,
                  "seconds": dt, "colour_blocks_per_s": done / dt, "block_channels_per_s": 3 * done / dt,
                  "stream_GB_per_s_in_plus_out": (in_bytes + out_bytes) / dt / 1e9, "files": a.dir}))
