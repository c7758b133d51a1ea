#!/usr/bin/env python3
"""Independent exact model of what the reference's resize pipeline computes at the PLAINTEXT level:
the ring Z_t[x]/(x^n+1) with SEAL's FractionalEncoder (base 2, 100+100 coefficients), the op sequence
of homo/fhe_resize.h (Linear :191-204, Cubic :143-189 with t3 = t*t, SampleLinear/SampleBicubic
:222-305, index arithmetic :350-388 in float32) and the client's decode/truncate/clamp
(homo/client_resize.cpp:200-203).  No ciphertexts, no RNS, no NTT: a correct BFV implementation
decrypts to exactly these polynomials whenever the noise budget is positive, including when
coefficients wrap modulo a small t.

Source pixels: tests/golden/boazbarak_stb_rgb.npy = stbi_load(image/boazbarak.jpg, 3 channels) as the
reference's client decodes it (homo/client_resize.cpp:96; stb_image differs from libjpeg in 59 samples
of this image).  Reference image for the RMS: tests/stubs/opencv2/opencv.hpp's imread + resize, passed
in as an array.

  bilinear, any t >= 11 -> 17.9597 (published)      bicubic, t >= 31 -> 19.8048 (published)
  bicubic, t = 11       -> 29.715 with the committed client's CLAMP(pixel, 0, 255) (homo/client_resize.cpp:208),
                           34.4 (published) when the decoded int is cast to uint8_t WITHOUT the clamp (modulo 256):
      at t = 11 plaintext coefficients wrap and 23 of the 867 decoded samples leave [0, 255] (-142 ... 397); at every
      t >= 31, and for bilinear at every t, all samples stay inside [17, 239], so the clamp changes nothing there.
      The table in benchmark/results.txt was therefore produced by a client without the clamp line; every
      deterministic entry of it is reproduced.

  usage: plain_ring_model.py T [bicubic] [wrap]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1024          # any n >= 512 gives the same polynomials: integer digits stay below x^40, fractional ones above x^(n-100)


def encode(v, t, n=N):
    p = np.zeros(n, dtype=np.int64)
    whole = int(v)
    frac = v - whole
    a, d = abs(whole), 0
    while a:
        if a & 1:
            p[d] = 1 if v >= 0 else t - 1
        a >>= 1
        d += 1
    for i in range(1, 101):
        frac *= 2
        b = int(frac)
        frac -= b
        if b:
            p[n - i] = (t - 1) if v >= 0 else 1
    return p % t


def mul(a, b, t):
    n = len(a)
    c = np.convolve(a, b)
    r = c[:n].copy()
    r[:n - 1] -= c[n:]
    return r % t


def decode(p, t):
    n = len(p)
    c = np.where(p >= (t + 1) // 2, p - t, p)
    val = sum(int(c[i]) * (2.0 ** i) for i in range(100) if c[i])
    val -= sum(int(c[n - i]) * (2.0 ** -i) for i in range(1, n - 100 + 1) if c[n - i])
    return val


def to_pixel(r, conversion):
    """`int pixel = encoder.decode(p); CLAMP(pixel, 0, 255) (uint8_t) pixel` (homo/client_resize.cpp:207-209):
    'clamp' is the committed code, 'wrap' the same cast without the CLAMP line."""
    pv = -2 ** 31 if abs(r) >= 2 ** 31 else int(r)       # cvttsd2si: out of range -> INT_MIN
    if conversion == "wrap":
        return pv % 256
    return max(0, min(255, pv))


def model(src_rgb, w, h, t, bicubic, conversion="clamp"):
    H, W = src_rgb.shape[:2]
    f32 = np.float32
    E = lambda v: encode(v, t)
    M = lambda a, b: mul(a, b, t)
    c3, c2, c5, c4, chalf, one = E(3), E(2), E(5), E(4), E(0.5), E(1.0)

    def cubic(A, B, C, D, tt):
        a = (M(B, c3) - A - M(C, c3) + D) % t
        b = (M(A, c2) - M(B, c5) + M(C, c4) - D) % t
        c = (C - A) % t
        t2 = M(tt, tt)
        r = (M(a, t2) + M(b, t2) + M(c, tt)) % t          # t3 = t*t (homo/fhe_resize.h:175)
        return (M(r, chalf) + B) % t

    def linear(A, B, tt):
        return (M((one - tt) % t, A) + M(B, tt)) % t

    cl = lambda v, lo, hi: max(lo, min(hi, v))
    pix = {v: E(float(v)) for v in range(256)}
    out = np.zeros((h, w, 3), dtype=np.int64)
    for y in range(h):
        v = f32(f32(f32(f32(y) / f32(h - 1)) * f32(H)) - 0.5)
        for x in range(w):
            u = f32(f32(f32(f32(x) / f32(w - 1)) * f32(W)) - 0.5)
            xi, yi = int(u), int(v)
            xf, yf = E(float(f32(u) - f32(math.floor(u)))), E(float(f32(v) - f32(math.floor(v))))
            for ch in range(3):
                P = lambda dx, dy: pix[int(src_rgb[cl(yi + dy, 0, H - 1), cl(xi + dx, 0, W - 1), ch])]
                if bicubic:
                    cols = [cubic(P(-1, j), P(0, j), P(1, j), P(2, j), xf) for j in (-1, 0, 1, 2)]
                    r = decode(cubic(cols[0], cols[1], cols[2], cols[3], yf), t)
                else:
                    r = decode(linear(linear(P(0, 0), P(1, 0), xf), linear(P(0, 1), P(1, 1), xf), yf), t)
                out[y, x, ch] = to_pixel(r, conversion)
    return out


def rms_string(out_rgb, ref_rgb):
    d = out_rgb.astype(np.int64) - ref_rgb.astype(np.int64)
    return "%.6g" % math.sqrt(float((d * d).sum()) / d.size)       # std::cout << double, homo/fhe_resize.h:65-67


def reference_image(width=17, height=17):
    """cv::imread + cv::resize(INTER_LINEAR) of the benchmark image through the validated stand-in
    (tests/stubs/opencv2/opencv.hpp), as RGB [h, w, 3]: what compare_resize_opencv (homo/fhe_resize.h:40-68)
    measures the decrypted image against."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "standin_check")
        subprocess.check_call(["g++", "-O2", "-std=c++11", "-I" + os.path.join(ROOT, "tests", "stubs"),
                               os.path.join(ROOT, "tests", "stubs", "standin_check.cpp"), "-o", exe])
        subprocess.check_call([exe, os.path.join(ROOT, "tests", "golden", "boazbarak.jpg"), os.path.join(d, "r.raw"), str(width), str(height), "1"])
        raw = open(os.path.join(d, "r.raw"), "rb").read()
    return np.frombuffer(raw[8:], dtype=np.uint8).reshape(height, width, 3)[:, :, ::-1]


if __name__ == "__main__":
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 101
    bicubic = "bicubic" in sys.argv[2:]
    conversion = "wrap" if "wrap" in sys.argv[2:] else "clamp"
    src = np.load(os.path.join(ROOT, "tests", "golden", "boazbarak_stb_rgb.npy"))
    print("t=%d %s %s: RMSError %s" % (t, "bicubic" if bicubic else "bilinear", conversion,
                                       rms_string(model(src, 17, 17, t, bicubic, conversion), reference_image())))
