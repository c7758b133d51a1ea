"""Integer model of the pseudo-Mersenne arithmetic the u64 kernels run on (csrc/ntt_core.h, csrc/behz.hip), checked with
Python integers: every intermediate the kernels keep in a 64-bit (or 32-bit) register stays inside it for the largest
operands the static range tracking admits, and every result is congruent to the exact product and below the bound the
tracking assumes.  No GPU, no library: this pins the FORMULAS and BOUNDS (the kernels themselves are compared with the
oracle bit for bit in tests/test_gpu_parity.py)."""
import random

import pytest

M64, M32 = (1 << 64) - 1, (1 << 32) - 1
SEAL_A = [0x7FFFFFFF380001, 0x7FFFFFFEF00001, 0x3FFFFFFF000001, 0x3FFFFFFEF40001]       # SEAL 2.3, 55/55/54/54 bits (class PmA)


def is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def aux_primes(count, bits=58):
    """csrc/behz.hip fhe_behz_build: the largest primes = 1 (mod 2^17) below 2^bits"""
    out, cand = [], (1 << bits) + 1 - (1 << 17)
    while len(out) < count:
        if is_prime(cand):
            out.append(cand)
        cand -= 1 << 17
    return out


AUX_B = aux_primes(9)
CLASSES = {"A": dict(LIM=2048, RQ=96, CS=3, XB=192), "B": dict(LIM=256, RQ=24, CS=1, XB=48)}     # ntt_core.h PmA / PmB, sixteenths of q
FOLDED = 17


class Pm:
    def __init__(self, q):
        self.q, self.b = q, q.bit_length()
        self.delta, self.sh, self.mb = (1 << self.b) - q, self.b - 32, (1 << (self.b - 32)) - 1
        assert self.delta < (1 << 31) and 1 <= self.sh <= 31


def u64(v):
    assert 0 <= v <= M64, "64-bit register overflow"
    return v


def u32(v):
    assert 0 <= v <= M32, "32-bit register overflow"
    return v


def mul_pm(x, w, m):
    """ntt_core.h mul_pm: x < 2^62, w canonical with w2 = w 2^31 mod q"""
    assert x < (1 << 62)
    w2 = (w << 31) % m.q
    xl, xh = x & 0x7FFFFFFF, u32(x >> 31)
    wl, wh, vl, vh = w & M32, w >> 32, w2 & M32, w2 >> 32
    A = u64(xh * vl + xl * wl)
    B = u64(xh * vh + u64(xl * wh + (A >> 32)))
    zh = u32(B >> m.sh)
    zl = ((B & m.mb) << 32) | (A & M32)
    return u64(zh * m.delta + zl)


def fold_pm(x, m):
    top = (x >> 32) >> m.sh
    lo = (((x >> 32) & m.mb) << 32) | (x & M32)
    return u64(top * m.delta + lo)


def mulvv_pm(a, b, m):
    """ntt_core.h mulvv_pm: a < 2^(b+1), b canonical"""
    assert a < (1 << (m.b + 1)) and b < m.q
    al, ah, bl, bh = a & M32, a >> 32, b & M32, b >> 32
    P0 = u64(al * bl)
    mid = u64(ah * bl + u64(al * bh + (P0 >> 32)))
    top = u64(ah * bh + (mid >> 32))
    z = (P0 & M32) | ((mid & M32) << 32) | (top << 64)
    assert z == a * b
    zh_lo, zh_hi = (z >> m.b) & M32, u32(z >> (m.b + 32))
    zl = (P0 & M32) | (((mid & M32) & m.mb) << 32)
    F = u64(zh_lo * m.delta + zl)
    G = u64(zh_hi * m.delta + (F >> 32))
    zh2 = u32(G >> m.sh)
    lo = (F & M32) | (((G & M32) & m.mb) << 32)
    return u64(zh2 * m.delta + lo)


def cls_of(q):
    return "A" if q.bit_length() <= 55 else "B"


@pytest.mark.parametrize("q", SEAL_A + AUX_B)
def test_products_and_fold_stay_in_their_registers_and_below_the_class_bounds(q):
    m, c = Pm(q), CLASSES[cls_of(q)]
    rng = random.Random(q)
    ws = [0, 1, q - 1, q // 2] + [rng.randrange(q) for _ in range(200)]
    xs = [0, 1, q - 1, (1 << 62) - 1, (1 << 31) - 1, 1 << 31, ((1 << 31) - 1) << 31] + [rng.randrange(1 << 62) for _ in range(200)]
    for w in ws:
        for x in xs[:7] + rng.sample(xs[7:], 20):
            r = mul_pm(x, w, m)
            assert r % q == x * w % q and 16 * r < c["RQ"] * q
    for x in [0, M64, q, q - 1, 1 << m.b, (1 << 63) + 12345] + [rng.randrange(1 << 64) for _ in range(500)]:
        r = fold_pm(x, m)
        assert r % q == x % q and 16 * r < FOLDED * q
        assert r - q < q                                   # canon_pm: fold + one conditional subtraction
    for _ in range(500):
        a, b = rng.choice([rng.randrange(1 << (m.b + 1)), (1 << (m.b + 1)) - 1, fold_pm(rng.randrange(1 << 64), m)]), rng.choice([q - 1, rng.randrange(q)])
        r = mulvv_pm(a, b, m)
        assert r % q == a * b % q and 16 * r <= c["RQ"] * q
    assert (1 << 62) * 16 >= c["LIM"] * q                  # LIM q is inside the product's operand range


def pass_lo(L, P):
    return max(L - 4 * P - 4, 0)


def pass_stages(L, P):
    return min(4, L - 4 * P)


def clog(bd16):
    s = 0
    while (16 << s) < bd16:
        s += 1
    return s


@pytest.mark.parametrize("L", [10, 11, 12, 13, 14])
@pytest.mark.parametrize("name", ["A", "B"])
def test_static_range_plans_never_pass_the_operand_limit(L, name):
    """ntt_core.h pm_fwd_bound / pm_inv_plan, replayed: the forward transform's product operands and the inverse
    transform's differences stay at or below LIM q; inverse outputs are below RQ q; nothing passes 2^64 (4 LIM q)."""
    c = CLASSES[name]
    bd = 16
    for s in range(L):                                    # forward: uniform bound, fold all when it would pass LIM
        if bd > c["LIM"]:
            bd = FOLDED
        assert bd <= c["LIM"] and bd + (16 << c["CS"]) <= 4 * c["LIM"]
        bd += 16 << c["CS"]
    NP = (L + 3) // 4
    for e0 in (16, FOLDED, c["RQ"], c["RQ"] + 2 * (16 << clog(c["RQ"]))):       # canonical, folded, product, rgb sums
        for P in range(NP - 1, -1, -1):
            b = [e0 if P == NP - 1 else c["XB"]] * 16
            LO, S = pass_lo(L, P), pass_stages(L, P)
            for u in range(S - 1, -1, -1):
                sigma = 4 * P + u
                rb = (L - 1 - sigma) - LO
                for r0 in range(16):
                    if r0 & (1 << rb):
                        continue
                    r1 = r0 | (1 << rb)
                    if b[r0] + (16 << clog(b[r1])) > c["LIM"]:
                        b[r1] = FOLDED
                    if b[r0] + (16 << clog(b[r1])) > c["LIM"]:
                        b[r0] = FOLDED
                    assert b[r0] + (16 << clog(b[r1])) <= c["LIM"]
                    assert b[r0] + b[r1] <= 4 * c["LIM"]
                    b[r0] = c["RQ"] if sigma == 0 else b[r0] + b[r1]
                    b[r1] = c["RQ"]
            if P > 0:
                b = [FOLDED if v > c["XB"] else v for v in b]
                assert max(b) <= c["XB"]
            else:
                assert max(b) <= c["RQ"]


@pytest.mark.parametrize("L", [10, 11, 12, 13, 14])
def test_twiddle_table_order_is_a_permutation_inside_every_stage(L):
    """ntt_core.h pm_tw_index: position of twiddle 2^sigma + th (8 >> rb) + i in the [i][th] order the kernels read"""
    n, seen = 1 << L, set()
    for idx in range(n):
        if idx < 2:
            pos = idx
        else:
            sigma = idx.bit_length() - 1
            P = sigma // 4
            rb = (L - 1 - sigma) - pass_lo(L, P)
            off, cnt = idx - (1 << sigma), 8 >> rb
            th, i = divmod(off, cnt)
            pos = (1 << sigma) + (i << (sigma - 3 + rb)) + th
            assert (1 << sigma) <= pos < (2 << sigma)
        seen.add(pos)
    assert len(seen) == n


def test_two_column_sums_of_the_base_conversions_fit_for_the_p8192_constants():
    """csrc/behz.hip pm_mac / pm_acc_reduce with worst-case variables and worst-case 58-bit / 55-bit constants: the columns
    stay below 2^64, the quotient part below 2^32 (behz_pm_tables checks the same with the actual constants)."""
    rng = random.Random(5)
    for mod, terms in ((AUX_B[0], [(Pm(AUX_B[0]).q * 17 // 16, 29)] + [(qi - 1, 28) for qi in SEAL_A]),     # fast floor: folded D_b + four y_i
                       (AUX_B[4], [(AUX_B[0] - 1, 29), (AUX_B[1] - 1, 29)]),                                # Shenoy-Kumaresan, two z_j
                       (SEAL_A[2], [(AUX_B[0] - 1, 29), (AUX_B[1] - 1, 29)])):                              # back conversion, two z_j
        m = Pm(mod)
        for trial in range(200):
            A = B = exact = 0
            for vmax, split in terms:
                x = vmax if trial % 2 == 0 else rng.randrange(vmax + 1)
                c = mod - 1 if trial < 2 else rng.randrange(mod)
                c2 = (c << split) % mod
                xl, xh = x & ((1 << split) - 1), u32(x >> split)
                A = u64(A + xl * (c & M32) + xh * (c2 & M32))
                B = u64(B + xl * (c >> 32) + xh * (c2 >> 32))
                exact += x * c
            Bc = u64(B + (A >> 32))
            zh = u32(Bc >> m.sh)
            r = u64(zh * m.delta + (((Bc & m.mb) << 32) | (A & M32)))
            assert r % mod == exact % mod and r < (1 << m.b) + (m.delta << 32)      # what the kernels fold or multiply next
