"""server_jpeg over a ciphertext stream: the driver loop of homo/server_jpeg.cpp:109-153 on the GPU.

The reference reads one 8x8 block of R, G and B ciphertexts at a time from ./image/nothingpersonnel.txt
(3 x 64 Ciphertext::load calls, :115-124), runs rgb_to_ycc_fhe on the 64 pixels and encrypted_dct on
the three channels (:127-135) and appends Y, Cb, Cr interleaved per coefficient to ./image/zoop.txt
(:146-153).  Here the same stream is processed in waves of many blocks: a reader thread fills pinned
host buffers while the GPU works on the previous wave (file -> pinned -> HBM overlap), the colour
conversion and the block transform are the fused kernels, and the writer drains the previous wave.

Stream format = the facade's Ciphertext::save records (seal/seal.h: "FHEHIP1" magic, u32 polys,
u32 k, u32 n, u32 reserved, raw little-endian u64).  SEAL 2.3's own wire format is not pinned by
anything in the reference (no sample files).
"""
import struct
import threading

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


def server_jpeg(ctx, in_path, out_path, n_blocks, wave_blocks=16, quant=None, do_dct=True):
    """Process `n_blocks` colour blocks.  Input order per block: 64 R, 64 G, 64 B ciphertexts
    (homo/server_jpeg.cpp:115-124).  Output order per block: for i in 0..63: Y[i], Cb[i], Cr[i]
    (homo/server_jpeg.cpp:150-152).  quant=None reproduces the reference server (no quantize_fhe call);
    a 64-entry table applies quantize_fhe to every channel as well.  Returns blocks processed."""
    ev = Evaluator(ctx)
    plan = DctPlan(ctx, quant) if do_dct else None
    shape = (3, 64, 2, ctx.k, ctx.n)                       # one block: channel, pixel, poly, prime, coeff
    wave_blocks = max(1, min(wave_blocks, n_blocks))
    host = [torch.empty((wave_blocks,) + shape, dtype=torch.int64).pin_memory() for _ in range(2)]
    host_out = [torch.empty((wave_blocks,) + shape, dtype=torch.int64).pin_memory() for _ in range(2)]
    copy_stream = torch.cuda.Stream()

    def read_wave(f, buf, nb):
        arr = buf.numpy().view(np.uint64)
        for b in range(nb):
            for ch in range(3):
                for i in range(64):
                    read_ciphertext_into(f, arr[b, ch, i])

    def write_wave(f, buf, nb):
        arr = buf.numpy().view(np.uint64)
        for b in range(nb):
            for i in range(64):
                for ch in range(3):
                    write_ciphertext(f, arr[b, ch, i])

    waves = [(s, min(s + wave_blocks, n_blocks)) for s in range(0, n_blocks, wave_blocks)]
    with open(in_path, "rb") as fin, open(out_path, "wb") as fout:
        read_wave(fin, host[0], waves[0][1] - waves[0][0])
        writer = None
        for wi, (s, e) in enumerate(waves):
            nb = e - s
            cur = host[wi & 1]
            reader = None
            if wi + 1 < len(waves):                         # prefetch the next wave from the file
                ns, ne = waves[wi + 1]
                reader = threading.Thread(target=read_wave, args=(fin, host[(wi + 1) & 1], ne - ns))
                reader.start()
            with torch.cuda.stream(copy_stream):
                dev = cur[:nb].to(ctx.device, non_blocking=True)
            torch.cuda.current_stream().wait_stream(copy_stream)
            r, g, b = (dev[:, ch].reshape(nb * 64, 2, ctx.k, ctx.n).contiguous() for ch in range(3))
            ev.rgb_to_ycc(r, g, b)                          # in place: r,g,b now hold Y, Cb, Cr
            chans = []
            for t in (r, g, b):
                t = t.reshape(nb, 64, 2, ctx.k, ctx.n)
                chans.append(ev.dct8x8_quant(plan, t) if do_dct else t)
            res = torch.stack(chans, dim=1)                 # [nb, 3, 64, 2, k, n]
            if writer is not None:
                writer.join()                               # host_out[wi & 1] was written two waves ago
            host_out[wi & 1][:nb].copy_(res, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            writer = threading.Thread(target=write_wave, args=(fout, host_out[wi & 1], nb))
            writer.start()
            if reader is not None:
                reader.join()
        if writer is not None:
            writer.join()
    return n_blocks
