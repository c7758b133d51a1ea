#!/usr/bin/env python3
"""End-to-end time of the streaming server_decode (homo/server_decode.cpp:113-148) on a 4x4 image (16 positions = BASELINE.json
configs[3]'s run length), order 64, degree 12, n = 8192, with the server's own encryptions made for real: per run 16 * 12 * 2
fresh encode(0) for homomorphic_sin / cos (homo/fhe_decode.h:54,134) + the index and accumulators (homo/server_decode.cpp:121,126).
--encrypt host: keys.Encryptor one at a time (rounds 2-4); device: keys.DeviceEncryptor batches (fhe_encrypt_batch).
Prints one JSON line (I/O-inclusive; not the bench.py metric)."""
import argparse, json, os, sys, tempfile, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="P8192")
ap.add_argument("--encrypt", choices=["host", "device"], default="device")
ap.add_argument("--degree", type=int, default=12)
a = ap.parse_args()
ctx = fhe.SEALContext.preset(a.preset)
kg = fhe.KeyGenerator(ctx)
enc = fhe.FractionalEncoder(ctx)
rgb = np.zeros((4, 4, 3), dtype=np.uint8)
rgb[:, :, 0] = 200
rgb[2:, :, 0] = 30                       # two runs
rgb[:, :, 1] = 90                        # one run
rgb[:1, :, 2], rgb[1:3, :, 2], rgb[3:, :, 2] = 10, 120, 250      # three runs
d = tempfile.mkdtemp(dir="/dev/shm")
fin, fout = os.path.join(d, "runs.ct"), os.path.join(d, "out.ct")
try:
    w, h, pairs = fhe.client.send_decode(ctx, fhe.DeviceEncryptor(ctx, kg.public_key()), enc, rgb, fin)
    zeros = fhe.server.make_zero_encryptor(ctx, kg.public_key(), device=a.encrypt == "device")
    fhe.server.server_decode(ctx, fin, fout, w, h, pairs, zeros, order=64, degree=a.degree)        # first pass: tables, page-locking
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fhe.server.server_decode(ctx, fin, fout, w, h, pairs, zeros, order=64, degree=a.degree)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_enc = sum(1 + w * h + p * w * h * a.degree * 2 for p in pairs)
    # the same job from the C++ host (seal/server_decode_hip.cpp); a single process: includes its context, tables and first-touch allocations
    cpp = None
    exe = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "seal", "server_decode_hip")
    if os.path.exists(exe) and a.encrypt == "device":
        import subprocess
        fpk = os.path.join(d, "pk.txt")
        with open(fpk, "wb") as f:
            fhe.server.write_ciphertext(f, fhe.to_host(kg.public_key()))
        env = dict(os.environ, FHE_SEAL23_MODULI="1") if a.preset != "P4096" else dict(os.environ)
        t1 = time.perf_counter()
        r = subprocess.run([exe, fin, fout, fpk, str(w), str(h)] + [str(p) for p in pairs] + ["64", str(a.degree), "0.5", str(ctx.n), str(ctx.t)], capture_output=True, text=True, timeout=900, env=env)
        cpp = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]) if r.returncode == 0 else {"error": r.stderr[-400:]}
        cpp["process_wall_seconds"] = time.perf_counter() - t1
        os.remove(fpk)
    print(json.dumps({"workload": "server_decode stream, 4x4 image, runs per channel %s, order 64, degree %d, %s" % (pairs, a.degree, a.preset),
                      "server_side_encryptions": a.encrypt, "encryptions": n_enc, "runs": sum(pairs), "seconds": dt, "ms_per_run": dt * 1e3 / sum(pairs), "cpp_host": cpp}))
finally:
    for p in (fin, fout):
        if os.path.exists(p):
            os.remove(p)
    os.rmdir(d)
