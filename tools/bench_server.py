#!/usr/bin/env python3
"""End-to-end rate of the streaming server_jpeg loop at real size: ciphertext stream
file -> page-locked host -> HBM -> rgb_to_ycc + DCT (fused kernels) -> page-locked host -> file, colour blocks
(3 channels x 64 ciphertexts = 36 MiB in, 36 MiB out) at n = 4096, k = 3.  Files live in a tmpfs so that the number shows
the host-side loop + PCIe, not a disk; the input file is generated on the device and written with the library's own
record writer.  The block count adapts to the space the tmpfs has (--blocks is the wish, default 1024 = 36 GiB + 36 GiB).
Steady state: one warm-up call on a few waves first (page-locks the staging buffers, builds the constants), then the
timed call.  Prints one JSON line (NOT the bench.py metric: I/O-inclusive)."""
import argparse, ctypes as C, json, os, shutil, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fhip_amd as fhe

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=1024)
ap.add_argument("--wave", type=int, default=32)
ap.add_argument("--io-threads", type=int, default=16)
ap.add_argument("--slots", type=int, default=3)
ap.add_argument("--dir", default="/dev/shm")
a = ap.parse_args()
ctx = fhe.SEALContext.preset("P4096")
rec = fhe.server.RECORD_HEADER + 2 * ctx.k * ctx.n * 8
per_block = 192 * rec
free = shutil.disk_usage(a.dir).free
staging = 2 * a.slots * a.wave * per_block                              # page-locked buffers also come out of host memory
blocks = max(a.wave, min(a.blocks, int((free * 0.85 - staging) // (2 * per_block)) // a.wave * a.wave))
fin, fout = os.path.join(a.dir, "fhe_in.ct"), os.path.join(a.dir, "fhe_out.ct")
# input stream: random-residue ciphertexts generated on the device, wave by wave, written with fhe_io_write_records
host = torch.empty((a.wave, 3, 64, 2, ctx.k, ctx.n), dtype=torch.int64).pin_memory()
fd = os.open(fin, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
os.ftruncate(fd, blocks * per_block)
t0 = time.time()
for s in range(0, blocks, a.wave):
    host.copy_(ctx.random_ct(a.wave, 3, 64, seed=fhe.SEED, first_index=s * 192 * 2 * ctx.k * ctx.n))
    fhe._lib.call("fhe_io_write_records", fd, s * 192, a.wave * 192, 2, ctx.k, ctx.n, C.c_void_p(host.data_ptr()), 8)
os.close(fd)
gen_s = time.time() - t0
del host
in_bytes = os.path.getsize(fin)
try:
    warm = os.path.join(a.dir, "fhe_warm.ct")
    fhe.server.server_jpeg(ctx, fin, warm, min(blocks, 3 * a.wave), wave_blocks=a.wave, io_threads=a.io_threads, slots=a.slots)     # warm-up: page-locking, constants
    os.remove(warm)
    torch.cuda.synchronize()
    fresh = {}
    t0 = time.time()
    fhe.server.server_jpeg(ctx, fin, fout, blocks, wave_blocks=a.wave, io_threads=a.io_threads, slots=a.slots, stats=fresh)          # output file does not exist yet
    torch.cuda.synchronize()
    fresh_dt = time.time() - t0
    # steady state of a long-lived server: its spool files stay mapped from call to call (page-table entries in place)
    sin = fhe.server.StreamFile(fin)
    sout = fhe.server.StreamFile(fout, write=True, size=blocks * per_block)
    fhe.server.server_jpeg(ctx, sin, sout, blocks, wave_blocks=a.wave, io_threads=a.io_threads, slots=a.slots)
    torch.cuda.synchronize()
    stats = {}
    t0 = time.time()
    done = fhe.server.server_jpeg(ctx, sin, sout, blocks, wave_blocks=a.wave, io_threads=a.io_threads, slots=a.slots, stats=stats)
    torch.cuda.synchronize()
    dt = time.time() - t0
    sin.close()
    sout.close()
    if os.environ.get("FHE_SERVER_TRACE"):
        for row in sorted(stats["trace"], key=lambda r: r[2]):
            print("%-6s wave %3d  %8.1f -> %8.1f ms" % (row[0], row[1], row[2] * 1e3, row[3] * 1e3), file=sys.stderr)
    out_bytes = os.path.getsize(fout)
    # the same job from the C++ host (seal/server_jpeg_hip.cpp): three passes over its own mappings, the last one reported
    cpp = None
    exe = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "seal", "server_jpeg_hip")
    if os.path.exists(exe):
        import subprocess
        want_bytes = open(fout, "rb").read(4 * per_block)
        r = subprocess.run([exe, fin, fout, str(blocks), str(a.wave), str(a.io_threads), "3"], capture_output=True, text=True)
        if r.returncode == 0:
            cpp = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            cpp["first_blocks_equal_python_server"] = open(fout, "rb").read(4 * per_block) == want_bytes
        else:
            cpp = {"error": r.stderr[-500:]}
    # spot check: the first block of the output stream against the kernels run directly on the first block of the input
    ev = fhe.Evaluator(ctx)
    first = ctx.random_ct(1, 3, 64, seed=fhe.SEED, first_index=0)
    ev.rgb_to_ycc_blocks(first)
    want = fhe.to_host(ev.dct8x8_quant(fhe.DctPlan(ctx, None), first.view(3, 64, 2, ctx.k, ctx.n)))
    got = np.zeros((192, 2, ctx.k, ctx.n), dtype=np.uint64)
    fd = os.open(fout, os.O_RDONLY)
    fhe._lib.call("fhe_io_read_records", fd, 0, 192, 2, ctx.k, ctx.n, got.ctypes.data_as(C.c_void_p), 4)
    os.close(fd)
    ok = bool(np.array_equal(got.reshape(want.shape), want))
finally:
    for p in (fin, fout):
        if os.path.exists(p):
            os.remove(p)
print(json.dumps({"workload": "server_jpeg stream (rgb_to_ycc + encrypted_dct per colour block), n=4096 k=3, files in " + a.dir,
                  "blocks": done, "wave_blocks": a.wave, "staging_slots": a.slots, "io_threads": a.io_threads,
                  "seconds": dt, "colour_blocks_per_s": done / dt, "block_channels_per_s": 3 * done / dt,
                  "stream_GB_per_s_in_plus_out": (in_bytes + out_bytes) / dt / 1e9, "stream_GiB_in": in_bytes / 2**30,
                  "device_compute_seconds": stats.get("device_compute_seconds"), "device_compute_share": stats.get("device_compute_seconds", 0) / dt,
                  "file_read_GB_per_s_while_reading": in_bytes / max(stats.get("file_read_seconds", 0), 1e-9) / 1e9,
                  "file_write_GB_per_s_while_writing": out_bytes / max(stats.get("file_write_seconds", 0), 1e-9) / 1e9,
                  "fresh_output_file": {"seconds": fresh_dt, "colour_blocks_per_s": blocks / fresh_dt, "file_write_GB_per_s_while_writing": out_bytes / max(fresh.get("file_write_seconds", 0), 1e-9) / 1e9,
                                        "note": "first pass into a file that does not exist yet: every output page is allocated by the kernel on first touch (tmpfs), which is serialised inside the kernel; the headline figures are a long-lived server's steady state: spool files that exist and stay mapped from call to call"},
                  "cpp_host": cpp, "input_generation_seconds": gen_s, "first_block_equals_direct_kernels": ok, "host_cpus": os.cpu_count()}))
