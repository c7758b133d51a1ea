"""CPU: algebraic invariants and decrypt-known-answer tests that pin the oracle without SEAL."""
import numpy as np
import pytest

from oracle import bigint_model as bm

Q3 = [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001]
T = 1 << 14


@pytest.fixture(scope="module")
def orc1k(oracle_mod):
    return oracle_mod.Oracle(1024, Q3, T)


def test_ntt_roundtrip_and_schoolbook(oracle_mod):
    orc = oracle_mod.Oracle(64, Q3, T)
    rng = np.random.default_rng(0)
    for i, q in enumerate(Q3):
        a = rng.integers(0, q, size=64, dtype=np.uint64)
        b = rng.integers(0, q, size=64, dtype=np.uint64)
        assert np.array_equal(orc.ntt_inv(orc.ntt_fwd(a, i), i), a)
        fa, fb = orc.ntt_fwd(a, i), orc.ntt_fwd(b, i)
        prod = orc.ntt_inv(np.array([(int(x) * int(y)) % q for x, y in zip(fa, fb)], dtype=np.uint64), i)
        ref = bm.polymul_negacyclic_schoolbook([int(x) for x in a], [int(x) for x in b])
        assert [int(x) for x in prod] == [x % q for x in ref]
    # auxiliary (BEHZ) base transforms too
    a = rng.integers(0, orc.aux[0], size=64, dtype=np.uint64)
    assert np.array_equal(orc.ntt_inv(orc.ntt_fwd(a, 0, base=1), 0, base=1), a)


def test_kronecker_product_equals_schoolbook():
    import random
    random.seed(3)
    a = [random.randrange(-10**40, 10**40) for _ in range(32)]
    b = [random.randrange(-2**13, 2**13) for _ in range(32)]
    assert bm.polymul_negacyclic(a, b) == bm.polymul_negacyclic_schoolbook(a, b)


def test_rns_is_crt(oracle_mod):
    orc = oracle_mod.Oracle(64, Q3, T)
    m = bm.Model(64, Q3, T)
    ct = orc.random_ct(1)[0]
    assert np.array_equal(np.array(m.to_rns(m.from_rns(ct)), dtype=np.uint64), ct)


def test_oracle_vs_model_direct_small(oracle_mod):
    """a fresh (non-golden) seed through both implementations"""
    n = 128
    orc, m = oracle_mod.Oracle(n, Q3, T), bm.Model(n, Q3, T)
    cts = orc.random_ct(3, seed=20260929)
    A, B, C = (m.from_rns(c) for c in cts)
    eq = lambda x, y: np.array_equal(x, np.array(m.to_rns(y), dtype=np.uint64))
    assert eq(orc.multiply(cts[0], cts[1]), m.multiply(A, B))
    assert eq(orc.multiply(orc.multiply(cts[0], cts[1]), cts[2]), m.multiply(m.multiply(A, B), C))
    pl = [int(x) for x in np.random.default_rng(1).integers(0, T, size=n)]
    assert eq(orc.multiply_plain(cts[0], np.array(pl, dtype=np.uint64)), m.multiply_plain(A, pl))


def test_encrypt_decrypt_and_noise_budget(orc1k):
    sk, pk = orc1k.keygen(5)
    for v in [0.0, 1.0, -1.0, 37.25, -255.75, 0.541196100]:
        plain, budget = orc1k.decrypt(sk, orc1k.encrypt(pk, orc1k.encode(v), seed=int(abs(v) * 7) + 3))
        assert orc1k.decode(plain) == v
        assert budget > 60


def test_model_decrypt_agrees_with_oracle(orc1k):
    sk, pk = orc1k.keygen(6)
    m = bm.Model(1024, Q3, T)
    ct = orc1k.multiply(orc1k.encrypt(pk, orc1k.encode(3.5), seed=1), orc1k.encrypt(pk, orc1k.encode(-2.25), seed=2))
    po, bo = orc1k.decrypt(sk, ct)
    skc = [int(x) if x < 2 else -1 for x in sk[0]]
    pm, bmd = m.decrypt(skc, m.from_rns(ct))
    assert [int(x) for x in po] == pm and bo == bmd
    assert orc1k.decode(po) == 3.5 * -2.25


def test_multiply_plain_and_add_plain_known_answer(orc1k):
    sk, pk = orc1k.keygen(7)
    ct = orc1k.encrypt(pk, orc1k.encode(200.0))
    got = orc1k.decode(orc1k.decrypt(sk, orc1k.multiply_plain(ct, orc1k.encode(0.541196100)))[0])
    assert abs(got - 200.0 * 0.541196100) < 1e-9
    got = orc1k.decode(orc1k.decrypt(sk, orc1k.sub_plain(ct, orc1k.encode(128.0)))[0])
    assert got == 72.0
    got = orc1k.decode(orc1k.decrypt(sk, orc1k.add_plain(ct, orc1k.encode(-4.71238898038469)))[0])
    assert abs(got - (200.0 - 4.71238898038469)) < 1e-9


def test_square_equals_multiply_and_relinearize_preserves_plaintext(orc1k):
    sk, pk = orc1k.keygen(8)
    ct = orc1k.encrypt(pk, orc1k.encode(-6.5))
    sq = orc1k.square(ct)
    assert np.array_equal(sq, orc1k.multiply(ct, ct))
    evk = orc1k.evk_gen(sk, dbc=16)
    rl = orc1k.relinearize(sq, evk, dbc=16)
    assert rl.shape[0] == 2
    plain, budget = orc1k.decrypt(sk, rl)
    assert orc1k.decode(plain) == 42.25 and budget > 0


def test_dct_circuit_known_answer_vs_plain_dct(orc1k, oracle_mod):
    """decrypt(encrypted_dct+quantize(encrypt(pixels))) == plaintext dct() of homo/fhe_image.h:400-484 / YQT"""
    sk, pk = orc1k.keygen(9)
    vals = [float((37 * x + 101 * y) % 256) - 128.0 for y in range(8) for x in range(8)]
    blk = np.stack([orc1k.encrypt(pk, orc1k.encode(v), seed=100 + i) for i, v in enumerate(vals)])
    out = orc1k.dct_quant(blk, oracle_mod.YQT)
    expect = bm.plain_dct(vals)
    for i in range(64):
        plain, budget = orc1k.decrypt(sk, out[i])
        assert budget > 0
        assert abs(orc1k.decode(plain) - expect[i] / oracle_mod.YQT[i]) < 1e-6


def test_rgb_to_ycc_known_answer(orc1k):
    sk, pk = orc1k.keygen(10)
    r, g, b = 200.0, 31.0, 77.0
    cr, cg, cb = (orc1k.encrypt(pk, orc1k.encode(v), seed=s) for v, s in ((r, 1), (g, 2), (b, 3)))
    y, u, v = orc1k.rgb_to_ycc(cr, cg, cb)
    dec = lambda c: orc1k.decode(orc1k.decrypt(sk, c)[0])
    assert abs(dec(y) - (0.299 * r + 0.587 * g + 0.114 * b - 128.0)) < 1e-9
    assert abs(dec(u) - (-0.168736 * r - 0.331264 * g + 0.5 * b)) < 1e-9
    assert abs(dec(v) - (0.5 * r - 0.418688 * g - 0.081312 * b)) < 1e-9


def test_cubic_and_linear_known_answer(oracle_mod):
    """Cubic / Linear of homo/fhe_resize.h:143-204 incl. the reference's t3 = t*t quirk (:175)"""
    orc = oracle_mod.Oracle(2048, Q3, T)   # three primes leave budget for depth 2
    sk, pk = orc.keygen(11)
    A, B, C, D, t = 10.0, 50.0, 90.0, 40.0, 0.25
    cA, cB, cC, cD, ct = (orc.encrypt(pk, orc.encode(v), seed=s) for s, v in enumerate((A, B, C, D, t)))
    res = orc.cubic(cA, cB, cC, cD, ct)
    assert res.shape[0] == 4
    a, b, c, d = -A + 3 * B - 3 * C + D, 2 * A - 5 * B + 4 * C - D, C - A, B
    t2 = t * t
    expect = 0.5 * (a * t2 + b * t2 + c * t) + d      # a*t3 with t3 == t*t, as the reference computes it
    plain, budget = orc.decrypt(sk, res)
    assert budget > 0 and abs(orc.decode(plain) - expect) < 1e-6
    lin = orc.linear(cA, cB, ct)
    assert lin.shape[0] == 3
    assert abs(orc.decode(orc.decrypt(sk, lin)[0]) - ((1 - t) * A + t * B)) < 1e-9


def test_transform_constant_product_equals_plain_mulmod(oracle_mod):
    """the oracle's butterflies multiply by table constants through a precomputed quotient (fhe_oracle.c mulmod_const);
    it must be the canonical residue a * w mod q for any 64-bit a, at the smallest and largest primes in use"""
    import random
    L = oracle_mod.lib()
    rng = random.Random(7)
    for q in (0xFFFFEE001, 0x1FFFFE0001, 0x3FFFFFFF000001, 0x7FFFFFFF380001, 0x1FFFFFFFFFE00001, 12289):
        cases = [(0, 0), (q - 1, q - 1), ((1 << 64) - 1, q - 1), (q, 1), (1, q - 1)]
        cases += [(rng.getrandbits(64), rng.randrange(q)) for _ in range(2000)]
        for a, w in cases:
            assert L.fo_mulmod_const_check(a, w, q) == (a * w) % q, (a, w, q)


def test_relinearised_compositions_decrypt_to_the_closed_forms(oracle_mod):
    """the checker of the relinearised mode (oracle.RelinOracle + the op-by-op restatements of Cubic / Linear / the samplers) pinned
    on the CPU: fo_cubic == oracle_cubic_calls on the plain oracle (the C restatement and the Python one agree bit for bit), and
    under RelinOracle every product comes back with two polynomials and the results decrypt to the closed forms with budget to
    spare (n = 4096, three 36/37-bit moduli, dbc 30 and 60; at this small q the key switch costs a few bits against the reference's
    size-4 result -- 30 against 33 at dbc 30 -- where at n = 8192 it gains them: tests/test_gpu_relin.py prints both)"""
    om = oracle_mod
    orc = om.Oracle(4096, Q3, T)
    sk, pk = orc.keygen(5)
    enc = lambda v, s: orc.encrypt(pk, orc.encode(v), seed=s)
    A, B, C_, D, t = 10.0, 50.0, 90.0, 40.0, 0.25
    cA, cB, cC, cD, ct = (enc(v, 20 + i) for i, v in enumerate((A, B, C_, D, t)))
    ref = orc.cubic(cA, cB, cC, cD, ct)
    assert np.array_equal(ref, om.oracle_cubic_calls(orc, cA, cB, cC, cD, ct)) and ref.shape[0] == 4
    assert np.array_equal(orc.linear(cA, cB, ct), om.oracle_linear_calls(orc, cA, cB, ct))
    a, b, c = -A + 3 * B - 3 * C_ + D, 2 * A - 5 * B + 4 * C_ - D, C_ - A
    cubic = lambda A_, B_, C__, D_, t_: 0.5 * ((-A_ + 3 * B_ - 3 * C__ + D_) * t_ * t_ + (2 * A_ - 5 * B_ + 4 * C__ - D_) * t_ * t_ + (C__ - A_) * t_) + B_
    expect = 0.5 * (a * t * t + b * t * t + c * t) + B
    p_ref, b_ref = orc.decrypt(sk, ref)
    assert orc.decode(p_ref) == expect
    for dbc in (30, 60):
        rorc = om.RelinOracle(orc, orc.evk_gen(sk, dbc=dbc), dbc)
        rel = om.oracle_cubic_calls(rorc, cA, cB, cC, cD, ct)
        assert rel.shape[0] == 2
        p_rel, b_rel = orc.decrypt(sk, rel)
        assert orc.decode(p_rel) == expect and b_rel > 0 and b_ref > 0, (dbc, b_rel, b_ref)
        lin = om.oracle_linear_calls(rorc, cA, cB, ct)
        assert lin.shape[0] == 2 and orc.decode(orc.decrypt(sk, lin)[0]) == (1 - t) * A + t * B
    # level 2 -- a column Cubic over four row results, as SampleBicubic composes them -- needs the room of n = 8192 (218-bit q)
    orc = om.Oracle.preset("P8192")
    sk, pk = orc.keygen(6)
    cA, cB, cC, cD, ct = (orc.encrypt(pk, orc.encode(v), seed=40 + i) for i, v in enumerate((A, B, C_, D, t)))
    rorc = om.RelinOracle(orc, orc.evk_gen(sk, dbc=30), 30)
    rows = [om.oracle_cubic_calls(rorc, cA, cB, cC, cD, ct), om.oracle_cubic_calls(rorc, cB, cC, cD, cA, ct),
            om.oracle_cubic_calls(rorc, cC, cD, cA, cB, ct), om.oracle_cubic_calls(rorc, cD, cA, cB, cC, ct)]
    col = om.oracle_sample_bicubic_calls(rorc, [cA, cB, cC, cD, cB, cC, cD, cA, cC, cD, cA, cB, cD, cA, cB, cC], ct, ct)
    assert np.array_equal(col, om.oracle_cubic_calls(rorc, rows[0], rows[1], rows[2], rows[3], ct))
    vals = [cubic(A, B, C_, D, t), cubic(B, C_, D, A, t), cubic(C_, D, A, B, t), cubic(D, A, B, C_, t)]
    plain, budget = orc.decrypt(sk, col)
    assert col.shape[0] == 2 and budget > 0 and abs(orc.decode(plain) - cubic(vals[0], vals[1], vals[2], vals[3], t)) < 1e-9
