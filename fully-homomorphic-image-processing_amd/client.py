"""Client side of the JPEG pipeline: the two halves of homo/client_jpeg.cpp over this package's
KeyGenerator / Encryptor / Decryptor.  Plaintext, CPU-side work.

  send_jpeg     -- homo/client_jpeg.cpp:73-166: every pixel encoded (FractionalEncoder) and encrypted,
                   written as 8x8 blocks: 64 R, 64 G, 64 B ciphertexts per block, blocks row-major
                   (split_image_eight_block, homo/fhe_image.h:108-124)
  receive_jpeg  -- homo/client_jpeg.cpp:196-296: per block and channel decrypt + decode 64 coefficients,
                   round half away from zero (:277), place in zig-zag order, entropy-code the block
                   (DC difference + run-length AC symbols, the baseline Huffman tables of ITU-T T.81
                   Annex K) into a JFIF file with all-ones quantisation tables (QUALITY = 0,
                   homo/fhe_image.h:32) and 1x1 sampling for Y, Cb, Cr
  rms_error     -- the quantity of compare_jpeg_jojpeg (homo/fhe_image.h:508-521) between two decoded images

The Huffman code tables are derived here from the (counts per length, symbol values) specification of
Annex K (tables K.3-K.6) by the canonical-code construction of T.81 Annex C, not transcribed."""
import math
import os

import numpy as np

from . import server

# zig-zag position of natural-order coefficient i (T.81 Figure 5)
ZIGZAG = np.zeros(64, dtype=np.int64)
_order, _i, _j = [], 0, 0
for _s in range(15):
    _rng = range(max(0, _s - 7), min(_s, 7) + 1)
    for _r in (_rng if _s % 2 else reversed(_rng)):
        _order.append(_r * 8 + (_s - _r))
for _pos, _nat in enumerate(_order):
    ZIGZAG[_nat] = _pos

# T.81 Annex K.3: number of codes of each length 1..16 and the symbol values, in order
DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHROMA = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D], [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xA1,
    0x08, 0x23, 0x42, 0xB1, 0xC1, 0x15, 0x52, 0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0A, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA,
    0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6,
    0xD7, 0xD8, 0xD9, 0xDA, 0xE1, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF1, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9,
    0xFA])
AC_CHROMA = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77], [
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xA1, 0xB1, 0xC1, 0x09, 0x23, 0x33, 0x52, 0xF0, 0x15, 0x62, 0x72, 0xD1, 0x0A, 0x16, 0x24, 0x34, 0xE1, 0x25, 0xF1, 0x17, 0x18, 0x19,
    0x1A, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8,
    0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4,
    0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9,
    0xFA])


def _canonical_codes(spec):
    """T.81 Annex C: symbol -> (code, length) from (counts per length, values)."""
    counts, values = spec
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[values[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


class _BitWriter:
    def __init__(self, f):
        self.f, self.acc, self.cnt = f, 0, 0

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.cnt += length
        while self.cnt >= 8:
            byte = (self.acc >> (self.cnt - 8)) & 0xFF
            self.f.write(bytes([byte, 0]) if byte == 0xFF else bytes([byte]))      # byte stuffing after 0xFF
            self.cnt -= 8
        self.acc &= (1 << self.cnt) - 1


def _magnitude(v):
    """(additional bits, category): category = bit length of |v|, negative values as v - 1 (T.81 F.1.2.1)"""
    size = int(abs(v)).bit_length()
    return (v if v >= 0 else v - 1) & ((1 << size) - 1), size


def _encode_block(bw, du, dc_pred, htdc, htac):
    """one 8x8 block of zig-zag-ordered integer coefficients; returns the new DC predictor"""
    diff = du[0] - dc_pred
    if diff == 0:
        bw.put(*htdc[0])
    else:
        bits, size = _magnitude(diff)
        bw.put(*htdc[size])
        bw.put(bits, size)
    last = 63
    while last > 0 and du[last] == 0:
        last -= 1
    if last == 0:
        bw.put(*htac[0x00])                                  # end of block
        return du[0]
    i = 1
    while i <= last:
        run = 0
        while du[i] == 0 and i <= last:
            run += 1
            i += 1
        while run >= 16:
            bw.put(*htac[0xF0])                              # sixteen zeros
            run -= 16
        bits, size = _magnitude(du[i])
        bw.put(*htac[(run << 4) + size])
        bw.put(bits, size)
        i += 1
    if last != 63:
        bw.put(*htac[0x00])
    return du[0]


def _dht_segment(tc_th, spec):
    return bytes([tc_th]) + bytes(spec[0]) + bytes(spec[1])


def write_jpeg_from_coefficients(path, coeff_blocks, width, height):
    """coeff_blocks: iterable of [3][64] integer arrays (Y, Cb, Cr per 8x8 block, NATURAL order, already rounded),
    blocks row-major.  Writes a baseline JFIF file the way homo/client_jpeg.cpp:218-292 does: quantisation tables
    of ones, 1x1 sampling, the Annex K Huffman tables, one interleaved scan."""
    tables = [_canonical_codes(s) for s in (DC_LUMA, AC_LUMA, DC_CHROMA, AC_CHROMA)]
    with open(path, "wb") as f:
        f.write(bytes([0xFF, 0xD8, 0xFF, 0xE0, 0, 0x10]) + b"JFIF\x00" + bytes([1, 1, 0, 0, 1, 0, 1, 0, 0]))
        f.write(bytes([0xFF, 0xDB, 0, 0x84, 0]) + bytes([1] * 64) + bytes([1]) + bytes([1] * 64))
        f.write(bytes([0xFF, 0xC0, 0, 0x11, 8, height >> 8, height & 0xFF, width >> 8, width & 0xFF, 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1]))
        f.write(bytes([0xFF, 0xC4, 0x01, 0xA2]) + _dht_segment(0x00, DC_LUMA) + _dht_segment(0x10, AC_LUMA)
                + _dht_segment(0x01, DC_CHROMA) + _dht_segment(0x11, AC_CHROMA))
        f.write(bytes([0xFF, 0xDA, 0, 0xC, 3, 1, 0, 2, 0x11, 3, 0x11, 0, 0x3F, 0]))
        bw = _BitWriter(f)
        pred = [0, 0, 0]
        for blk in coeff_blocks:
            for ch in range(3):
                zz = [0] * 64
                for j in range(64):
                    zz[int(ZIGZAG[j])] = int(blk[ch][j])
                ht = (tables[0], tables[1]) if ch == 0 else (tables[2], tables[3])
                pred[ch] = _encode_block(bw, zz, pred[ch], ht[0], ht[1])
        bw.put(0x7F, 7)                                       # pad the last byte with ones
        f.write(bytes([0xFF, 0xD9]))


def round_half_away(v):
    """homo/client_jpeg.cpp:277: v < 0 ? ceilf(v - 0.5f) : floorf(v + 0.5f), in float32 like the reference"""
    f = np.float32(v)
    return int(np.ceil(f - np.float32(0.5))) if v < 0 else int(np.floor(f + np.float32(0.5)))


def blocks_of(channel, width, height):
    """split_image_eight_block (homo/fhe_image.h:108-124): 8x8 blocks row-major, 64 values each"""
    a = np.asarray(channel).reshape(height, width)
    return [a[j:j + 8, i:i + 8].reshape(64) for j in range(0, height - height % 8, 8) for i in range(0, width - width % 8, 8)]


def send_jpeg(ctx, encryptor, encoder, rgb, out_path):
    """rgb: uint8 [H, W, 3].  Writes the ciphertext stream server.server_jpeg reads; returns the block count."""
    import torch
    h, w, _ = rgb.shape
    chans = [blocks_of(rgb[:, :, c].astype(np.float64), w, h) for c in range(3)]
    with open(out_path, "wb") as f:
        _encrypt_values(ctx, encryptor, encoder, [v for b in range(len(chans[0])) for c in range(3) for v in chans[c][b]], f)
    return len(chans[0])


def receive_jpeg(ctx, decryptor, encoder, in_path, width, height, out_path):
    """Decrypt the server's stream (per block: 64 Y, 64 Cb, 64 Cr) and write the JPEG.  Returns the rounded
    coefficient blocks [n_blocks][3][64]."""
    n_blocks = (width // 8) * (height // 8)
    blocks = []
    with open(in_path, "rb") as f:
        for _ in range(n_blocks):
            recs = []
            for _ in range(192):
                ct = np.zeros((2, ctx.k, ctx.n), dtype=np.uint64)
                server.read_ciphertext_into(f, ct)
                recs.append(ct)
            plains = _decrypt_records(ctx, decryptor, recs)                 # one block's 64 Y, 64 Cb, 64 Cr in one fhe_decrypt_batch
            blk = np.zeros((3, 64), dtype=np.int64)
            for ch in range(3):
                for j in range(64):
                    blk[ch, j] = round_half_away(encoder.decode(plains[ch * 64 + j]))
            blocks.append(blk)
    write_jpeg_from_coefficients(out_path, blocks, width, height)
    return blocks


def rms_error(a, b):
    """compare_jpeg_jojpeg's number (homo/fhe_image.h:513-520) for two decoded uint8 images of equal shape"""
    d = np.asarray(a, dtype=np.int64) - np.asarray(b, dtype=np.int64)
    return math.sqrt(float((d * d).sum()) / d.size)


# ------------------------------------------------------------------------------------------------
# client halves of the resize and decode pipelines (homo/client_resize.cpp, homo/client_decode.cpp)
# ------------------------------------------------------------------------------------------------
def _encrypt_values(ctx, encryptor, encoder, values, f, chunk=1024):
    """encode + encrypt + write, in stream order.  A keys.DeviceEncryptor (anything with `encrypt_values`) works in device batches
    (fhe_frac_encode_batch + fhe_encrypt_batch, one download per chunk); a keys.Encryptor one ciphertext at a time."""
    import torch
    if hasattr(encryptor, "encrypt_values"):
        if (encryptor.int_coeffs, encryptor.frac_coeffs) != (encoder.int_coeffs, encoder.frac_coeffs):
            raise ValueError("the encryptor's encoder parameters differ from the encoder's")
        values = [float(v) for v in values]
        for s in range(0, len(values), chunk):
            cts = encryptor.encrypt_values(values[s:s + chunk])
            torch.cuda.synchronize()
            for ct in cts.cpu().numpy().view(np.uint64):
                server.write_ciphertext(f, ct)
        return
    for v in values:
        ct = encryptor.encrypt(encoder.encode(float(v)))
        torch.cuda.synchronize()
        server.write_ciphertext(f, ct.cpu().numpy().view(np.uint64))


def send_resize(ctx, encryptor, encoder, rgb, out_path):
    """homo/client_resize.cpp:141-158: every sample of the image encoded and encrypted, "RGBRGB... row by row" (:141),
    the stream server.server_resize reads.  rgb: uint8 [H, W, 3].  Returns (width, height)."""
    h, w, _ = rgb.shape
    with open(out_path, "wb") as f:
        _encrypt_values(ctx, encryptor, encoder, np.asarray(rgb, dtype=np.uint8).reshape(-1), f)
    return w, h


def to_pixel(value, clamp=True):
    """`int pixel = encoder.decode(p); CLAMP(pixel, 0, 255); (uint8_t) pixel` (homo/client_resize.cpp:206-209,
    homo/client_decode.cpp:205-207): truncation toward zero, saturated to int as x86 cvttsd2si does for values
    outside the int range (INT_MIN), then the clamp.  clamp=False is the same conversion without the CLAMP line --
    the client that produced the bicubic t = 11 entries of benchmark/results.txt."""
    v = float(value)
    pixel = int(v) if math.isfinite(v) and abs(v) < 2147483648.0 else -2147483648
    if clamp:
        pixel = 0 if pixel < 0 else 255 if pixel > 255 else pixel
    return pixel & 0xFF


def _decrypt_records(ctx, decryptor, recs):
    """recs: host arrays [size, k, n] (sizes may differ: a decode stream interleaves 22- and 2-polynomial records) -> plaintexts in the
    same order; records of one size go through keys.Decryptor.decrypt_batch (fhe_decrypt_batch) together"""
    import torch
    plains = [None] * len(recs)
    if not hasattr(decryptor, "decrypt_batch"):
        return [decryptor.decrypt(torch.from_numpy(r.view(np.int64).copy()).to(ctx.device)) for r in recs]
    for size in sorted({r.shape[0] for r in recs}):
        idx = [i for i, r in enumerate(recs) if r.shape[0] == size]
        batch = torch.from_numpy(np.stack([recs[i] for i in idx]).view(np.int64)).to(ctx.device)
        for i, p in zip(idx, decryptor.decrypt_batch(batch)):
            plains[i] = p
    return plains


def receive_pixels(ctx, decryptor, encoder, in_path, width, height, clamp=True, decoded=None):
    """The receiving half shared by homo/client_resize.cpp:197-211 and homo/client_decode.cpp:200-209: width * height * 3
    records (ciphertexts of any size: 4 or 6 polynomials after a resize, 22 after the run-length decoder), each
    decrypted, decoded and converted by to_pixel.  Returns uint8 [height, width, 3]; `decoded` (a list) receives the
    decoded doubles in stream order."""
    out = np.zeros(width * height * 3, dtype=np.uint8)
    with open(in_path, "rb") as f:
        for first in range(0, out.size, 256):
            recs = []
            for i in range(first, min(first + 256, out.size)):
                hdr = f.read(server.RECORD_HEADER)
                if len(hdr) != server.RECORD_HEADER:
                    raise EOFError("ciphertext stream ended")
                magic, size, k, n, _ = server.HEADER.unpack(hdr)
                if magic != server.MAGIC or (k, n) != (ctx.k, ctx.n) or not 1 <= size <= 64:
                    raise ValueError("not a ciphertext record of this context")
                ct = np.frombuffer(f.read(size * k * n * 8), dtype=np.uint64)
                if ct.size != size * k * n:
                    raise EOFError("truncated ciphertext record")
                recs.append(ct.reshape(size, k, n))
            for i, plain in enumerate(_decrypt_records(ctx, decryptor, recs)):
                v = encoder.decode(plain)
                if decoded is not None:
                    decoded.append(v)
                out[first + i] = to_pixel(v, clamp)
    return out.reshape(height, width, 3)


def receive_resize(ctx, decryptor, encoder, in_path, width, height, clamp=True, decoded=None):
    """homo/client_resize.cpp:163-222 without its OpenCV comparison: the resized image, uint8 [height, width, 3]"""
    return receive_pixels(ctx, decryptor, encoder, in_path, width, height, clamp, decoded)


def run_length_pairs(channel):
    """The run-length encoder of homo/client_decode.cpp:126-148 for one colour channel (values in scan order):
    [(value, count), ...]"""
    vals = [int(v) for v in channel]
    runs, curr, count = [], vals[0], 1
    for v in vals[1:]:
        if v == curr:
            count += 1
        else:
            runs.append((curr, count))
            curr, count = v, 1
    runs.append((curr, count))
    return runs


def send_decode(ctx, encryptor, encoder, rgb, out_path):
    """homo/client_decode.cpp:122-153: per colour channel the run-length pairs of the image, each as two ciphertexts
    (value, count), channel after channel -- the stream server.server_decode reads.  Returns (width, height, pairs[3]),
    the five integers the reference writes to keys/params.txt (:95-98,149)."""
    h, w, _ = rgb.shape
    flat = np.asarray(rgb, dtype=np.uint8).reshape(-1, 3)
    pairs = []
    with open(out_path, "wb") as f:
        for ch in range(3):
            runs = run_length_pairs(flat[:, ch])
            pairs.append(len(runs))
            _encrypt_values(ctx, encryptor, encoder, [x for run in runs for x in run], f)
    return w, h, pairs


def receive_decode(ctx, decryptor, encoder, in_path, width, height, clamp=True, decoded=None):
    """homo/client_decode.cpp:157-214: the decoded image, uint8 [height, width, 3] (position-major, channels interleaved:
    the order homo/server_decode.cpp:139-143 saves)"""
    return receive_pixels(ctx, decryptor, encoder, in_path, width, height, clamp, decoded)
