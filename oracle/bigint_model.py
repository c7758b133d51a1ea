"""Independent big-integer model of the SEAL-2.3 Evaluator semantics (TEST INFRASTRUCTURE).

Nothing here shares code with oracle/fhe_oracle.c or with the HIP library: polynomials are
lists of Python ints modulo the *composite* q (no RNS, no NTT), products are computed by
Kronecker substitution on Python big integers, and BEHZ is evaluated from its integer-level
definition without any auxiliary base.  It is used to (a) pin the C oracle (tests/
test_oracle_vs_model.py) and (b) generate tests/golden/*.npz (tests/golden/make_golden.py).

Citations are to /root/reference (call sites) and SURVEY.md Appendix A (SEAL semantics,
restated from the SEAL 2.3 manual / BEHZ paper; SEAL is not vendored => parity unpinned).
"""
from functools import reduce


def prod(xs):
    return reduce(lambda a, b: a * b, xs, 1)


# ---------------------------------------------------------------- CRT ------
def crt_compose(residues, q):
    """residues[i][c] -> list of ints in [0, Q)."""
    Q = prod(q)
    n = len(residues[0])
    out = [0] * n
    for i, qi in enumerate(q):
        Mi = Q // qi
        inv = pow(Mi, -1, qi)
        f = Mi * inv
        ri = residues[i]
        for c in range(n):
            out[c] += int(ri[c]) * f
    return [x % Q for x in out]


def crt_decompose(poly, q):
    return [[x % qi for x in poly] for qi in q]


# ---------------------------------------------- negacyclic integer product --
def _pack(a, bits):
    # signed Kronecker packing: sum a_i 2^(bits*i) evaluated with Python ints
    x = 0
    for c in reversed(a):
        x = (x << bits) + c
    return x


def _unpack(x, count, bits):
    out = []
    mask = (1 << bits) - 1
    half = 1 << (bits - 1)
    neg = x < 0
    if neg:
        x = -x
    for _ in range(count):
        c = x & mask
        x >>= bits
        if c >= half:  # borrow: this limb is negative
            c -= 1 << bits
            x += 1
        out.append(c)
    return [-c for c in out] if neg else out


def polymul_negacyclic(a, b):
    """Exact product of integer polynomials a, b (len n) in Z[x]/(x^n + 1)."""
    n = len(a)
    amax = max(1, max(abs(x) for x in a))
    bmax = max(1, max(abs(x) for x in b))
    bits = (amax * bmax * n).bit_length() + 2
    full = _unpack(_pack(a, bits) * _pack(b, bits), 2 * n - 1, bits)
    out = full[:n]
    for i in range(n, 2 * n - 1):
        out[i - n] -= full[i]
    return out


def polymul_negacyclic_schoolbook(a, b):
    n = len(a)
    out = [0] * n
    for i, x in enumerate(a):
        if x == 0:
            continue
        for j, y in enumerate(b):
            k = i + j
            if k < n:
                out[k] += x * y
            else:
                out[k - n] -= x * y
    return out


# -------------------------------------------------- FractionalEncoder -------
def frac_encode(v, n, t, int_coeffs=100, frac_coeffs=100):
    """SEAL 2.3 FractionalEncoder(t, poly, 100, 100, base 2).encode(double)
    (call sites homo/fhe_image.h:221-236)."""
    plain = [0] * n
    ip = int(v)  # truncation toward zero
    f = v - ip
    mag = abs(ip)
    d = 0
    while mag and d < int_coeffs:
        if mag & 1:
            plain[d] = (t - 1) if ip < 0 else 1
        mag >>= 1
        d += 1
    if f != 0.0:
        neg = v < 0
        for i in range(1, frac_coeffs + 1):
            f *= 2.0
            b = int(f)
            f -= b
            if b:
                plain[n - i] = 1 if neg else t - 1
    return plain


def centre(m, t):
    return m - t if m >= (t + 1) // 2 else m


def frac_decode(plain, t, int_coeffs=100):
    n = len(plain)
    ip = sum(centre(plain[d], t) << d for d in range(int_coeffs))
    fr = 0.0
    for idx in range(int_coeffs, n):
        fr = (fr + centre(plain[idx], t)) / 2.0
    return ip - fr


# ------------------------------------------------------- the model ---------
class Model:
    def __init__(self, n, q, t):
        self.n, self.q, self.t = n, list(q), t
        self.Q = prod(q)
        self.delta = self.Q // t

    # ciphertexts here are lists of polys; each poly a list of n ints in [0,Q)
    def from_rns(self, ct):
        return [crt_compose(p, self.q) for p in ct]

    def to_rns(self, ct):
        return [crt_decompose(p, self.q) for p in ct]

    def add(self, a, b):
        Q = self.Q
        s = max(len(a), len(b))
        out = []
        for j in range(s):
            x = a[j] if j < len(a) else [0] * self.n
            y = b[j] if j < len(b) else [0] * self.n
            out.append([(u + v) % Q for u, v in zip(x, y)])
        return out

    def sub(self, a, b):
        Q = self.Q
        s = max(len(a), len(b))
        out = []
        for j in range(s):
            x = a[j] if j < len(a) else [0] * self.n
            y = b[j] if j < len(b) else [0] * self.n
            out.append([(u - v) % Q for u, v in zip(x, y)])
        return out

    def negate(self, a):
        return [[(-u) % self.Q for u in p] for p in a]

    def _plain_centered(self, plain):
        p = [centre(int(m), self.t) for m in plain]
        return p + [0] * (self.n - len(p))

    def multiply_plain(self, a, plain):
        P = self._plain_centered(plain)
        return [[x % self.Q for x in polymul_negacyclic(p, P)] for p in a]

    def add_plain(self, a, plain, sign=1):
        P = self._plain_centered(plain)
        out = [list(p) for p in a]
        out[0] = [(u + sign * self.delta * m) % self.Q for u, m in zip(out[0], P)]
        return out

    def sub_plain(self, a, plain):
        return self.add_plain(a, plain, -1)

    # --- BEHZ, integer-level -----------------------------
    MTILDE = 1 << 32

    def _fastbconv_int(self, residues_fn):
        """sum_i |x_i (Q/q_i)^-1|_{q_i} (Q/q_i), as an integer in [0, kQ)."""
        tot = 0
        for qi in self.q:
            Mi = self.Q // qi
            tot += ((residues_fn(qi) * pow(Mi, -1, qi)) % qi) * Mi
        return tot

    def _behz_lift(self, poly):
        mt, Q = self.MTILDE, self.Q
        negqinv = (-pow(Q, -1, mt)) % mt
        out = []
        for a in poly:
            x = self._fastbconv_int(lambda qi: (a % qi) * (mt % qi))
            r = (x * negqinv) % mt
            if r >= mt // 2:
                r -= mt
            num = x + Q * r
            assert num % mt == 0
            out.append(num // mt)
        return out

    def multiply(self, a, b):
        Q, t = self.Q, self.t
        al = [self._behz_lift(p) for p in a]
        bl = [self._behz_lift(p) for p in b]
        so = len(a) + len(b) - 1
        D = [[0] * self.n for _ in range(so)]
        for i, x in enumerate(al):
            for j, y in enumerate(bl):
                pr = polymul_negacyclic(x, y)
                D[i + j] = [u + v for u, v in zip(D[i + j], pr)]
        out = []
        for d in D:
            res = []
            for v in d:
                tv = t * v
                Y = self._fastbconv_int(lambda qi: tv % qi)
                assert (tv - Y) % Q == 0
                res.append(((tv - Y) // Q) % Q)
            out.append(res)
        return out

    def square(self, a):
        return self.multiply(a, a)

    # --- decrypt (exact) --------------------------------------------------------
    def decrypt(self, sk, ct):
        """sk: centred-integer secret poly; returns (plain coeffs, noise budget bits)."""
        Q, t = self.Q, self.t
        acc = list(ct[-1])
        for j in range(len(ct) - 2, -1, -1):
            acc = [(u + v) % Q for u, v in zip(polymul_negacyclic(acc, sk), ct[j])]
        plain, worst = [], 0
        for x in acc:
            m = (t * x + Q // 2) // Q
            worst = max(worst, abs(t * x - m * Q))
            plain.append(m % t)
        budget = Q.bit_length() - worst.bit_length() - 1
        return plain, max(budget, 0)

    # --- circuits as the reference issues them -------------------------------------
    def enc(self, v):
        return frac_encode(v, self.n, self.t)

    def dct_line(self, d, scale):
        """homo/fhe_image.h:206-244 (scale False) / :246-284 (scale True)."""
        A, S, MP, E = self.add, self.sub, self.multiply_plain, self.enc
        tmp0, tmp7 = A(d[0], d[7]), S(d[0], d[7])
        tmp1, tmp6 = A(d[1], d[6]), S(d[1], d[6])
        tmp2, tmp5 = A(d[2], d[5]), S(d[2], d[5])
        tmp3, tmp4 = A(d[3], d[4]), S(d[3], d[4])
        tmp10, tmp13 = A(tmp0, tmp3), S(tmp0, tmp3)
        tmp11, tmp12 = A(tmp1, tmp2), S(tmp1, tmp2)
        out = [None] * 8
        out[0] = A(tmp10, tmp11)
        out[4] = S(tmp10, tmp11)
        z1 = MP(A(tmp12, tmp13), E(0.541196100))
        out[2] = A(z1, MP(tmp13, E(0.765366865)))
        out[6] = A(z1, MP(tmp12, E(-1.847759065)))
        z1, z2, z3, z4 = A(tmp4, tmp7), A(tmp5, tmp6), A(tmp4, tmp6), A(tmp5, tmp7)
        z5 = MP(A(z3, z4), E(1.175875602))
        tmp4 = MP(tmp4, E(0.298631336))
        tmp5 = MP(tmp5, E(2.053119869))
        tmp6 = MP(tmp6, E(3.072711026))
        tmp7 = MP(tmp7, E(1.501321110))
        z1 = MP(z1, E(-0.899976223))
        z2 = MP(z2, E(-2.562915447))
        z3 = MP(z3, E(-1.961570560))
        z4 = MP(z4, E(-0.390180644))
        z3, z4 = A(z3, z5), A(z4, z5)
        out[7] = A(A(tmp4, z1), z3)
        out[5] = A(A(tmp5, z2), z4)
        out[3] = A(A(tmp6, z2), z3)
        out[1] = A(A(tmp7, z1), z4)
        if scale:
            out = [MP(o, E(0.125)) for o in out]
        return out

    def encrypted_dct(self, data):
        data = list(data)
        for r in range(8):
            idx = [8 * r + i for i in range(8)]
            for i, o in zip(idx, self.dct_line([data[i] for i in idx], False)):
                data[i] = o
        for c in range(8):
            idx = [c + 8 * i for i in range(8)]
            for i, o in zip(idx, self.dct_line([data[i] for i in idx], True)):
                data[i] = o
        return data

    def quantize(self, data, quant):
        return [self.multiply_plain(d, self.enc(1 / q)) for d, q in zip(data, quant)]


# ---------------------------------------- plaintext DCT model (known answer) --
def plain_dct(block):
    """The floating-point twin of encrypted_dct: homo/fhe_image.h:400-484 (also tests/dct.cpp:127-213).
    block: 64 doubles row-major; returns 64 doubles."""
    d = list(block)

    def line(v, scale):
        tmp0, tmp7 = v[0] + v[7], v[0] - v[7]
        tmp1, tmp6 = v[1] + v[6], v[1] - v[6]
        tmp2, tmp5 = v[2] + v[5], v[2] - v[5]
        tmp3, tmp4 = v[3] + v[4], v[3] - v[4]
        tmp10, tmp13 = tmp0 + tmp3, tmp0 - tmp3
        tmp11, tmp12 = tmp1 + tmp2, tmp1 - tmp2
        o = [0.0] * 8
        o[0] = tmp10 + tmp11
        o[4] = tmp10 - tmp11
        z1 = (tmp12 + tmp13) * 0.541196100
        o[2] = z1 + tmp13 * 0.765366865
        o[6] = z1 + tmp12 * -1.847759065
        z1, z2, z3, z4 = tmp4 + tmp7, tmp5 + tmp6, tmp4 + tmp6, tmp5 + tmp7
        z5 = (z3 + z4) * 1.175875602
        tmp4 *= 0.298631336
        tmp5 *= 2.053119869
        tmp6 *= 3.072711026
        tmp7 *= 1.501321110
        z1 *= -0.899976223
        z2 *= -2.562915447
        z3 *= -1.961570560
        z4 *= -0.390180644
        z3 += z5
        z4 += z5
        o[7] = tmp4 + z1 + z3
        o[5] = tmp5 + z2 + z4
        o[3] = tmp6 + z2 + z3
        o[1] = tmp7 + z1 + z4
        return [x * 0.125 for x in o] if scale else o

    for r in range(8):
        d[8 * r: 8 * r + 8] = line(d[8 * r: 8 * r + 8], False)
    for c in range(8):
        col = line([d[c + 8 * i] for i in range(8)], True)
        for i in range(8):
            d[c + 8 * i] = col[i]
    return d
