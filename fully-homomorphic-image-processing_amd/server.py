"""server_jpeg over a ciphertext stream: the driver loop of homo/server_jpeg.cpp:109-153 on the GPU.

The reference reads one 8x8 block of R, G and B ciphertexts at a time from ./image/nothingpersonnel.txt
(3 x 64 Ciphertext::load calls, :115-124), runs rgb_to_ycc_fhe on the 64 pixels and encrypted_dct on
the three channels (:127-135) and appends the block's 64 Y, then 64 Cb, then 64 Cr ciphertexts to
./image/zoop.txt (save_three_blocks_interleaved_ycc, :146-153: "interleaved by block", channel-major
inside a block -- the order homo/client_jpeg.cpp:266-271 reads back: for k<3, for j<64).  Here the same stream is processed in waves of many blocks: a reader thread fills pinned
host buffers while the GPU works on the previous wave (file -> pinned -> HBM overlap), the colour
conversion and the block transform are the fused kernels, and the writer drains the previous wave.

Stream format = the facade's Ciphertext::save records (seal/seal.h: "FHEHIP1" magic, u32 polys,
u32 k, u32 n, u32 reserved, raw little-endian u64).  SEAL 2.3's own wire format is not pinned by
anything in the reference (no sample files).
"""
import os
import struct

import numpy as np
import torch

from .evaluator import DctPlan, Evaluator

MAGIC = b"FHEHIP1\x00"
HEADER = struct.Struct("<8sIIII")


def write_ciphertext(f, ct):
    """ct: numpy uint64 [size, k, n] -> one stream record."""
    size, k, n = ct.shape
    f.write(HEADER.pack(MAGIC, size, k, n, 0))
    f.write(np.ascontiguousarray(ct, dtype="<u8").tobytes())


def read_ciphertext_into(f, out):
    """Read one record into out (numpy uint64 [size, k, n], e.g. a view of a pinned buffer)."""
    hdr = f.read(HEADER.size)
    if len(hdr) != HEADER.size:
        raise EOFError("ciphertext stream ended")
    magic, size, k, n, _ = HEADER.unpack(hdr)
    if magic != MAGIC:
        raise ValueError("not a ciphertext record")
    if (size, k, n) != out.shape:
        raise ValueError("ciphertext shape %r does not match the context %r" % ((size, k, n), out.shape))
    got = f.readinto(memoryview(out).cast("B"))
    if got != out.nbytes:
        raise EOFError("truncated ciphertext record")


RECORD_HEADER = HEADER.size
_IO_THREADS = 8
_pinned_cache = {}


def _pinned(tag, shape):
    """pinned staging buffers are expensive to create (page locking): keep them between calls"""
    key = (tag, shape)
    if key not in _pinned_cache:
        _pinned_cache[key] = torch.empty(shape, dtype=torch.int64).pin_memory()
    return _pinned_cache[key]


def _pread_records(fd, views, first_record, rec_bytes, expect):
    """positional scatter reads (header -> scratch, payload -> straight into pinned memory); the
    records are of fixed size, so any number of threads can read disjoint ranges concurrently"""
    hdr = bytearray(RECORD_HEADER)
    for j, v in enumerate(views):
        mv = memoryview(v).cast("B")
        got = os.preadv(fd, [hdr, mv], (first_record + j) * rec_bytes)
        if got != rec_bytes:
            raise EOFError("ciphertext stream ended")
        magic, size, k, n, _ = HEADER.unpack(hdr)
        if magic != MAGIC:
            raise ValueError("not a ciphertext record")
        if (size, k, n) != expect:
            raise ValueError("ciphertext shape %r does not match the context %r" % ((size, k, n), expect))


def _pwrite_records(fd, views, first_record, rec_bytes, hdr):
    for j, v in enumerate(views):
        mv = memoryview(v).cast("B")
        if os.pwritev(fd, [hdr, mv], (first_record + j) * rec_bytes) != rec_bytes:
            raise IOError("short write on the ciphertext stream")


def server_jpeg(ctx, in_path, out_path, n_blocks, wave_blocks=8, quant=None, do_dct=True):
    """Process `n_blocks` colour blocks.  Input order per block: 64 R, 64 G, 64 B ciphertexts
    (homo/server_jpeg.cpp:115-124).  Output order per block: 64 Y, 64 Cb, 64 Cr
    (homo/server_jpeg.cpp:146-153; read back channel-major by homo/client_jpeg.cpp:266-271).  quant=None reproduces the reference server (no quantize_fhe call);
    a 64-entry table applies quantize_fhe to every channel as well.  Returns blocks processed.

    Pipeline per wave of blocks: a pool of I/O threads reads the next wave's fixed-size records with
    positional reads straight into a pinned buffer and writes the previous wave's results, while
    the GPU converts and transforms the current one (file -> pinned -> HBM -> pinned -> file)."""
    from concurrent.futures import ThreadPoolExecutor
    ev = Evaluator(ctx)
    plan = DctPlan(ctx, quant) if do_dct else None
    shape = (3, 64, 2, ctx.k, ctx.n)                       # one block: channel, pixel, poly, prime, coeff
    wave_blocks = max(1, min(wave_blocks, n_blocks))
    host = [_pinned(("in", i), (wave_blocks,) + shape) for i in range(2)]
    host_out = [_pinned(("out", i), (wave_blocks,) + shape) for i in range(2)]
    copy_stream = torch.cuda.Stream()
    rec_bytes = RECORD_HEADER + 2 * ctx.k * ctx.n * 8
    expect = (2, ctx.k, ctx.n)
    out_hdr = HEADER.pack(MAGIC, 2, ctx.k, ctx.n, 0)
    if os.path.getsize(in_path) < n_blocks * 192 * rec_bytes:
        raise EOFError("ciphertext stream ended")

    def read_wave(pool, fd, buf, first_block, nb):
        arr = buf.numpy().view(np.uint64)
        # one task per (block, channel): 64 consecutive records
        return [pool.submit(_pread_records, fd, [arr[b, ch, i] for i in range(64)],
                            ((first_block + b) * 3 + ch) * 64, rec_bytes, expect)
                for b in range(nb) for ch in range(3)]

    def write_wave(pool, fd, buf, first_block, nb):
        arr = buf.numpy().view(np.uint64)
        # record index (first_block + b) * 192 + ch * 64 + i: one task per (block, channel)
        return [pool.submit(_pwrite_records, fd, [arr[b, ch, i] for i in range(64)],
                            ((first_block + b) * 3 + ch) * 64, rec_bytes, out_hdr)
                for b in range(nb) for ch in range(3)]

    def wait(futs):
        for f in futs:
            f.result()

    waves = [(s, min(s + wave_blocks, n_blocks)) for s in range(0, n_blocks, wave_blocks)]
    fin = os.open(in_path, os.O_RDONLY)
    fout = os.open(out_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.ftruncate(fout, n_blocks * 192 * rec_bytes)
        # separate pools: reads of the next wave must not queue behind the writes of the previous one
        with ThreadPoolExecutor(_IO_THREADS) as pool, ThreadPoolExecutor(_IO_THREADS) as wpool:
            wait(read_wave(pool, fin, host[0], 0, waves[0][1] - waves[0][0]))
            pending = [[], []]
            for wi, (s, e) in enumerate(waves):
                nb = e - s
                cur = host[wi & 1]
                reading = []
                if wi + 1 < len(waves):                     # prefetch the next wave from the file
                    ns, ne = waves[wi + 1]
                    reading = read_wave(pool, fin, host[(wi + 1) & 1], ns, ne - ns)
                with torch.cuda.stream(copy_stream):
                    dev = cur[:nb].to(ctx.device, non_blocking=True)
                torch.cuda.current_stream().wait_stream(copy_stream)
                r, g, b = (dev[:, ch].reshape(nb * 64, 2, ctx.k, ctx.n).contiguous() for ch in range(3))
                ev.rgb_to_ycc(r, g, b)                      # in place: r,g,b now hold Y, Cb, Cr
                chans = []
                for t in (r, g, b):
                    t = t.reshape(nb, 64, 2, ctx.k, ctx.n)
                    chans.append(ev.dct8x8_quant(plan, t) if do_dct else t)
                res = torch.stack(chans, dim=1)             # [nb, 3, 64, 2, k, n]
                wait(pending[wi & 1])                       # host_out[wi & 1] was queued two waves ago
                host_out[wi & 1][:nb].copy_(res, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                pending[wi & 1] = write_wave(wpool, fout, host_out[wi & 1], s, nb)
                wait(reading)
            wait(pending[0])
            wait(pending[1])
    finally:
        os.close(fin)
        os.close(fout)
    return n_blocks
