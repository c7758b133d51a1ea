#!/usr/bin/env python3
"""End-to-end rate of the streaming server_resize loop (homo/fhe_resize.h:308-392 over a ciphertext stream) at the
size of BASELINE.json configs[2]: 128x128 -> 64x64, three channels, n = 8192 (24 GiB in, 18 GiB out for bicubic):
file -> page-locked host -> HBM (ring of source rows) -> fhe_sample_bicubic / fhe_sample_linear -> page-locked host -> file.
Files live in a tmpfs and stay mapped (a long-lived server's spool files); the circuit's server-side encryptions are
pre-made ciphertexts (inputs).  Prints one JSON line (NOT the bench.py metric: I/O-inclusive)."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="P8192")
ap.add_argument("--src", type=int, default=128)
ap.add_argument("--dst", type=int, default=64)
ap.add_argument("--bilinear", action="store_true")
ap.add_argument("--rows", type=int, default=4)
ap.add_argument("--io-threads", type=int, default=16)
ap.add_argument("--shared", action="store_true", help="one offset ciphertext per output column / row (server_resize(shared_offsets=True)) instead of two per output pixel")
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--encrypt", choices=["bank", "host", "device"], default="bank",
                help="the circuit's server-side encryptions (two per output pixel): bank = pre-made ciphertexts (the circuit alone, rounds 2-4), "
                     "host = keys.Encryptor one at a time (numpy sampler), device = keys.DeviceEncryptor batches (fhe_encrypt_batch)")
ap.add_argument("--relin", type=int, default=0, metavar="DBC", help="the relinearised mode of the circuits (records of 2 polynomials instead of 6 / 4): decomposition bit count")
ap.add_argument("--relin-placement", choices=["product", "cubic", "sample"], default="product", help="after every product, or once per Cubic / Linear (include/fhe_circuits.h)")
a = ap.parse_args()
ctx = fhe.SEALContext.preset(a.preset)
relin = None
if a.relin:
    kg = fhe.KeyGenerator(ctx, seed=1)
    relin = ((kg.generate_evaluation_keys(a.relin, 2).contiguous(), a.relin, "cubic") if a.relin_placement == "cubic" else
             (kg.generate_evaluation_keys(a.relin, 4).contiguous(), a.relin, "sample") if a.relin_placement == "sample" else (kg.generate_evaluation_keys(a.relin).contiguous(), a.relin))
fin, fout = os.path.join(a.dir, "fhe_rs_in.ct"), os.path.join(a.dir, "fhe_rs_out.ct")
out_size = 2 if relin else (4 if a.bilinear else 6)
rec_in = fhe.server.RECORD_HEADER + 2 * ctx.k * ctx.n * 8
rec_out = fhe.server.RECORD_HEADER + out_size * ctx.k * ctx.n * 8
n_in, n_out = a.src * a.src * 3, a.dst * a.dst * 3
# input stream generated on the device, one source row at a time
row = torch.empty((a.src, 3, 2, ctx.k, ctx.n), dtype=torch.int64).pin_memory()
sin = fhe.server.StreamFile(fin, write=True, size=n_in * rec_in)
for r in range(a.src):
    row.copy_(ctx.random_ct(a.src, 3, size=2, seed=fhe.SEED, first_index=r * a.src * 3 * 2 * ctx.k * ctx.n))
    sin.transfer(r * a.src * 3, a.src * 3, 2, ctx, row, 8)
sin.close()
bank = ctx.random_ct(max(a.rows * a.dst * 2, a.dst), size=2, seed=5)


def fractions(values):
    return bank[:len(values)]


if a.encrypt != "bank":
    fractions = fhe.server.make_fraction_encryptor(ctx, fhe.KeyGenerator(ctx).public_key(), device=a.encrypt == "device")


try:
    sin = fhe.server.StreamFile(fin)
    sout = fhe.server.StreamFile(fout, write=True, size=n_out * rec_out)
    fresh = {}
    fhe.server.server_resize(ctx, sin, sout, a.src, a.src, a.dst, a.dst, not a.bilinear, fractions, rows_per_step=a.rows, io_threads=a.io_threads, stats=fresh, shared_offsets=a.shared, relin=relin)   # first pass: page-locking, page allocation
    torch.cuda.synchronize()
    stats = {}
    t0 = time.time()
    done = fhe.server.server_resize(ctx, sin, sout, a.src, a.src, a.dst, a.dst, not a.bilinear, fractions, rows_per_step=a.rows, io_threads=a.io_threads, stats=stats, shared_offsets=a.shared, relin=relin)
    torch.cuda.synchronize()
    dt = time.time() - t0
    sin.close()
    sout.close()
    # the same job from the C++ host (seal/server_resize_hip.cpp): three passes over its own mappings, the last one reported
    cpp = None
    exe = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "seal", "server_resize_hip")
    if os.path.exists(exe) and a.preset in ("P8192", "P4096", "SEAL23_4096") and not relin:      # the C++ host streams the reference's mode
        import subprocess
        fpk = os.path.join(a.dir, "fhe_rs_pk.txt")
        with open(fpk, "wb") as f:
            fhe.server.write_ciphertext(f, fhe.to_host(fhe.KeyGenerator(ctx).public_key()))
        env = dict(os.environ, FHE_SEAL23_MODULI="1") if a.preset != "P4096" else dict(os.environ)
        try:
            r = subprocess.run([exe, fin, fout, fpk, str(a.src), str(a.src), str(a.dst), str(a.dst), "0" if a.bilinear else "1", str(a.rows), str(a.io_threads), str(ctx.n), str(ctx.t), "-", "3", "1" if a.shared else "0"],
                               capture_output=True, text=True, timeout=900, env=env)
            cpp = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]) if r.returncode == 0 else {"error": r.stderr[-400:]}
        finally:
            os.remove(fpk)
finally:
    for p in (fin, fout):
        if os.path.exists(p):
            os.remove(p)
print(json.dumps({"workload": "server_resize stream %dx%d -> %dx%d %s, three channels, %s, files in %s" % (a.src, a.src, a.dst, a.dst, "bilinear" if a.bilinear else "bicubic", a.preset, a.dir),
                  "mode": ("relinearised %s, dbc %d: records of 2 polynomials" % ({"cubic": "once per Cubic / Linear", "sample": "once per output pixel"}.get(a.relin_placement, "after every product"), a.relin)) if relin else "reference (no relinearisation)",
                  "output_pixels": done, "rows_per_step": a.rows, "server_side_encryptions": a.encrypt, "offsets": "shared (one per output column / row)" if a.shared else "per output pixel (the reference's)", "seconds": dt, "pixels_per_s": done / dt,
                  "stream_GB_in": stats["bytes_in"] / 1e9, "stream_GB_out": stats["bytes_out"] / 1e9, "stream_GB_per_s_in_plus_out": (stats["bytes_in"] + stats["bytes_out"]) / dt / 1e9,
                  "device_compute_seconds": stats["device_compute_seconds"], "device_compute_share": stats["device_compute_seconds"] / dt,
                  "file_read_seconds": stats["file_read_seconds"], "file_write_seconds": stats["file_write_seconds"],
                  "first_pass_seconds_incl_page_locking_and_page_allocation": fresh.get("seconds"), "cpp_host": cpp}))
