"""CPU: the C oracle against the committed golden vectors (tests/golden/, produced by the
independent big-integer model in oracle/bigint_model.py via tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def exact(oracle_mod):
    d = np.load(os.path.join(G, "exact_ops_n256.npz"))
    return d, oracle_mod.Oracle(int(d["n"]), d["q"].tolist(), int(d["t"]))


def test_synthetic_inputs_match_generator(exact, oracle_mod):
    d, orc = exact
    assert np.array_equal(orc.random_ct(5, seed=oracle_mod.SEED), d["inputs"])
    assert np.array_equal(orc.random_ct(1, size=4, seed=oracle_mod.SEED, first_index=77777)[0], d["input4"])


def test_add_sub_negate_golden(exact):
    d, orc = exact
    a, b, c4 = d["inputs"][0], d["inputs"][1], d["input4"]
    assert np.array_equal(orc.add(a, b), d["add"])
    assert np.array_equal(orc.sub(a, b), d["sub"])
    assert np.array_equal(orc.negate(a), d["negate"])
    assert np.array_equal(orc.add(a, c4), d["add_2_4"])       # unequal sizes grow the destination
    assert np.array_equal(orc.sub(a, c4), d["sub_2_4"])
    assert np.array_equal(orc.sub(c4, a), d["sub_4_2"])


def test_encoder_and_plain_ops_golden(exact):
    d, orc = exact
    a = d["inputs"][0]
    for i, v in enumerate(d["consts"]):
        pl = orc.encode(float(v))
        assert np.array_equal(pl, d["encoded"][i]), v
        assert np.array_equal(orc.multiply_plain(a, pl), d["multiply_plain"][i]), v
        assert np.array_equal(orc.add_plain(a, pl), d["add_plain"][i]), v
        assert np.array_equal(orc.sub_plain(a, pl), d["sub_plain"][i]), v
    assert np.array_equal(orc.multiply_plain(d["input4"], d["dense_plain"]), d["multiply_plain_dense"])


def test_encoder_golden_nonzero_patterns(oracle_mod):
    d = np.load(os.path.join(G, "encoder.npz"), allow_pickle=True)
    cache = {}
    for n, v, idx, co in zip(d["n"], d["value"], d["nz_index"], d["nz_coeff"]):
        n = int(n)
        if n not in cache:
            cache[n] = oracle_mod.Oracle(n, [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], int(d["t"]))
        pl = cache[n].encode(float(v))
        exp = np.zeros(n, dtype=np.uint64)
        exp[np.asarray(idx, dtype=np.int64)] = np.asarray(co, dtype=np.uint64)
        assert np.array_equal(pl, exp), (n, v)


@pytest.mark.parametrize("name", ["behz_n64.npz", "behz_seal23_n64.npz"])
def test_behz_multiply_square_golden(oracle_mod, name):
    d = np.load(os.path.join(G, name))
    orc = oracle_mod.Oracle(int(d["n"]), d["q"].tolist(), int(d["t"]))
    x = d["inputs"]
    ab = orc.multiply(x[0], x[1])
    assert np.array_equal(ab, d["mul_2x2"])
    assert np.array_equal(orc.square(x[0]), d["square_2"])
    if "mul_3x2" in d:
        abc = orc.multiply(ab, x[2])
        assert np.array_equal(abc, d["mul_3x2"])
        assert np.array_equal(orc.square(ab), d["square_3"])
        assert np.array_equal(orc.multiply(abc, ab), d["mul_4x3"])


def test_dct_quant_golden_n256(oracle_mod):
    d = np.load(os.path.join(G, "dct_quant_n256.npz"))
    orc = oracle_mod.Oracle(int(d["n"]), d["q"].tolist(), int(d["t"]))
    blk = orc.random_ct(64, seed=oracle_mod.SEED)
    dct = orc.encrypted_dct(blk)
    assert _sha(dct) == str(d["sha256_dct"])
    assert np.array_equal(dct[63], d["dct_full_ct63"])
    out = orc.quantize(dct, d["quant"].tolist())
    assert np.array_equal(out, d["dct_quant_full"])
    assert _sha(out) == str(d["sha256_dct_quant"])


def test_dct_quant_golden_n4096_digest(oracle_mod):
    """full-size parameters (BASELINE.json configs[1]), one block: SHA-256 of all 64 output cts"""
    d = np.load(os.path.join(G, "dct_quant_n4096_digest.npz"))
    orc = oracle_mod.Oracle(int(d["n"]), d["q"].tolist(), int(d["t"]))
    blk = orc.random_ct(64, seed=oracle_mod.SEED)
    out = orc.dct_quant(blk, d["quant"].tolist())
    assert _sha(out) == str(d["sha256_dct_quant"])
    idx = d["sample_index"]
    assert np.array_equal(out[idx][:, :, :, :16], d["dct_quant_sample"])
