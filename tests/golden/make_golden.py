#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the independent big-integer model (oracle/bigint_model.py).

The reference holds no golden ciphertexts (SEAL is an un-vendored submodule; tests/*.cpp only
print), so these vectors are produced by a model that shares no code with the C oracle or the HIP
library: composite-modulus big-integer arithmetic, Kronecker-substitution products, and the
integer-level definition of BEHZ.  Run from the repository root:

    python tests/golden/make_golden.py

Inputs are the deterministic synthetic ciphertexts of BASELINE.md section 3
(u64 = splitmix64(0x5EA12026 ^ linear_index) mod q_i), regenerated here in pure Python.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bigint_model as bm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
Q3 = [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001]      # BASELINE.json: n=4096, 3 coeff moduli
T = 1 << 14
SEED = 0x5EA12026
M64 = (1 << 64) - 1
YQT = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
       14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
       49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]
CONSTS = [0.541196100, 0.765366865, -1.847759065, 1.175875602, 0.298631336, 2.053119869, 3.072711026,
          1.501321110, -0.899976223, -2.562915447, -1.961570560, -0.390180644, 0.125, 128.0, 3.0, 0.5,
          -0.168736, 1 / 16.0, 1 / 99.0, -4.71238898038469, 0.0, 1.0, -1.0, 255.0]


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def random_ct(n, q, n_polys, seed=SEED, first=0):
    out = np.zeros((n_polys, len(q), n), dtype=np.uint64)
    idx = first
    for p in range(n_polys):
        for i, qi in enumerate(q):
            for c in range(n):
                out[p, i, c] = splitmix64(seed ^ idx) % qi
                idx += 1
    return out


def rns(model, ct):
    return np.array(model.to_rns(ct), dtype=np.uint64)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def gen_exact(n=256):
    m = bm.Model(n, Q3, T)
    raw = random_ct(n, Q3, 10).reshape(5, 2, 3, n)
    A, B = m.from_rns(raw[0]), m.from_rns(raw[1])
    big = random_ct(n, Q3, 4, first=77777)           # a size-4 ciphertext
    C4 = m.from_rns(big)
    d = dict(n=n, q=np.array(Q3, dtype=np.uint64), t=T, inputs=raw, input4=big,
             add=rns(m, m.add(A, B)), sub=rns(m, m.sub(A, B)), negate=rns(m, m.negate(A)),
             add_2_4=rns(m, m.add(A, C4)), sub_2_4=rns(m, m.sub(A, C4)), sub_4_2=rns(m, m.sub(C4, A)),
             consts=np.array(CONSTS))
    enc = np.zeros((len(CONSTS), n), dtype=np.uint64)
    mp = np.zeros((len(CONSTS), 2, 3, n), dtype=np.uint64)
    ap = np.zeros_like(mp)
    sp = np.zeros_like(mp)
    for i, v in enumerate(CONSTS):
        pl = bm.frac_encode(v, n, T)
        enc[i] = pl
        mp[i] = rns(m, m.multiply_plain(A, pl))
        ap[i] = rns(m, m.add_plain(A, pl))
        sp[i] = rns(m, m.sub_plain(A, pl))
    d.update(encoded=enc, multiply_plain=mp, add_plain=ap, sub_plain=sp)
    # dense plaintext with upper-half coefficients
    rs = np.random.RandomState(5)
    dense = rs.randint(0, T, size=n).astype(np.uint64)
    dense[0], dense[1], dense[2] = T - 1, (T + 1) // 2, (T + 1) // 2 - 1
    d.update(dense_plain=dense, multiply_plain_dense=rns(m, m.multiply_plain(C4, [int(x) for x in dense])))
    np.savez_compressed(os.path.join(OUT, "exact_ops_n256.npz"), **d)
    print("exact_ops_n256.npz")


def gen_behz(n=64):
    m = bm.Model(n, Q3, T)
    raw = random_ct(n, Q3, 8, first=4242).reshape(4, 2, 3, n)
    a, b, c = (m.from_rns(raw[i]) for i in range(3))
    ab = m.multiply(a, b)              # 2 x 2 -> 3
    abc = m.multiply(ab, c)            # 3 x 2 -> 4
    sq = m.square(a)                   # size 2
    sq3 = m.square(ab)                 # size 3 -> 5
    m43 = m.multiply(abc, ab)          # 4 x 3 -> 6
    np.savez_compressed(os.path.join(OUT, "behz_n64.npz"), n=n, q=np.array(Q3, dtype=np.uint64), t=T, inputs=raw,
                        mul_2x2=rns(m, ab), mul_3x2=rns(m, abc), square_2=rns(m, sq), square_3=rns(m, sq3),
                        mul_4x3=rns(m, m43))
    # a second parameter set: SEAL 2.3.1's n=4096 default (55/54-bit primes), exercised at n=64
    q2 = [0x7FFFFFFF380001, 0x3FFFFFFF000001]
    m2 = bm.Model(n, q2, T)
    raw2 = random_ct(n, q2, 4, first=999).reshape(2, 2, 2, n)
    x, y = m2.from_rns(raw2[0]), m2.from_rns(raw2[1])
    np.savez_compressed(os.path.join(OUT, "behz_seal23_n64.npz"), n=n, q=np.array(q2, dtype=np.uint64), t=T,
                        inputs=raw2, mul_2x2=rns(m2, m2.multiply(x, y)), square_2=rns(m2, m2.square(x)))
    print("behz_n64.npz behz_seal23_n64.npz")


def gen_dct(n, name, full):
    m = bm.Model(n, Q3, T)
    raw = random_ct(n, Q3, 128).reshape(64, 2, 3, n)
    data = [m.from_rns(raw[i]) for i in range(64)]
    dct = m.encrypted_dct(data)
    out = m.quantize(dct, YQT)
    dct_r = np.stack([rns(m, x) for x in dct])
    out_r = np.stack([rns(m, x) for x in out])
    d = dict(n=n, q=np.array(Q3, dtype=np.uint64), t=T, quant=np.array(YQT, dtype=np.float64),
             sha256_dct=sha(dct_r), sha256_dct_quant=sha(out_r),
             sample_index=np.array([0, 9, 36, 63]), dct_sample=dct_r[[0, 9, 36, 63], :, :, :16],
             dct_quant_sample=out_r[[0, 9, 36, 63], :, :, :16])
    if full:
        d.update(dct_quant_full=out_r, dct_full_ct63=dct_r[63])
    np.savez_compressed(os.path.join(OUT, name), **d)
    print(name)


def gen_encoder():
    rows = []
    for n in (256, 4096):
        for v in CONSTS + [1 / 16, 1 / 11, 37.25, -2.5, 1e-9, 12345.678, -255.75]:
            pl = bm.frac_encode(v, n, T)
            nz = [(i, c) for i, c in enumerate(pl) if c]
            rows.append((n, v, nz))
    np.savez_compressed(os.path.join(OUT, "encoder.npz"),
                        n=np.array([r[0] for r in rows]), value=np.array([r[1] for r in rows]),
                        nz_index=np.array([np.array([i for i, _ in r[2]], dtype=np.int64) for r in rows], dtype=object),
                        nz_coeff=np.array([np.array([c for _, c in r[2]], dtype=np.int64) for r in rows], dtype=object),
                        t=T, allow_pickle=True)
    print("encoder.npz")


if __name__ == "__main__":
    gen_encoder()
    gen_exact()
    gen_behz()
    gen_dct(256, "dct_quant_n256.npz", full=True)
    gen_dct(4096, "dct_quant_n4096_digest.npz", full=False)
