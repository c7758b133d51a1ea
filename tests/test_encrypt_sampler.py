"""CPU: the keyed sampler behind the servers' own encryptions (include/fhe_hip.h "server-side encryptions": fhe_encrypt_batch),
as restated in the oracle -- the thing the GPU parity tests (tests/test_gpu_encrypt.py) compare the HIP kernels with:
  * the ChaCha20 restatement against the published keystream blocks,
  * the noise table (both copies: oracle/fhe_oracle.c and the library's fhe_noise_cdt) against a 90-digit evaluation,
  * the distribution of the draws, and that encryptions made from them decrypt,
  * the oracle-backed C ABI (what the facade's CPU builds link) against the oracle's Python-facing entry point."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_chacha20_restatement_against_published_blocks(oracle_mod):
    om = oracle_mod
    zero = bytes(32)
    # D. J. Bernstein's ChaCha20 (64-bit counter, 64-bit nonce), all-zero key and nonce: the keystream every implementation document quotes
    assert om.chacha20_block(zero, 0, 0).hex() == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                                                   "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    assert om.chacha20_block(zero, 1, 0).hex().startswith("9f07e7be5551387a98ba977c732d080dcb0f29a048e3656912c6533e32ee7aed")
    # key = 00..01 and nonce = 1 (draft-strombergson-chacha-test-vectors TC2 / TC3, first blocks)
    assert om.chacha20_block(bytes([1] + [0] * 31), 0, 0).hex().startswith("c5d30a7ce1ec119378c84f487d775a8542f13ece238a9455e8229e888de85bbd")
    assert om.chacha20_block(zero, 0, 1).hex().startswith("ef3fdfd6c61578fbf5cf35bd3dd33b8009631634d21e42ac33960bd138e50d32")            # nonce bytes 01 00 .. 00
    assert om.chacha20_block(zero, 0, 1 << 56).hex().startswith("de9cba7bf3d69ef5e786dc63973f653a0b49e015adbff7134fcb7df137821031")      # nonce bytes 00 .. 00 01
    # counter and nonce are both 64-bit: the high words matter
    assert om.chacha20_block(zero, 1 << 32, 0) != om.chacha20_block(zero, 0, 0)
    assert om.chacha20_block(zero, 0, 1 << 32) != om.chacha20_block(zero, 0, 0)


def test_noise_table_against_90_digit_evaluation(oracle_mod, fhe):
    import noise_cdt
    want = noise_cdt.table()
    assert len(want) == 19 and all(a < b for a, b in zip(want, want[1:])) and want[-1] < 1 << 63
    assert oracle_mod.noise_cdt() == want
    out = (C.c_uint64 * 19)()
    fhe._lib.load().fhe_noise_cdt(out)                         # host-only entry point: no device needed
    assert list(out) == want


def test_draw_distribution_and_decryption(oracle_mod):
    om = oracle_mod
    orc = om.Oracle.preset("P4096")
    key = bytes(range(32))
    d = np.stack([orc.encrypt_draws(key, i) for i in range(64)])        # [64, 3, n]
    assert np.array_equal(d[5], orc.encrypt_draws(key, 5))
    assert not np.array_equal(d[5], d[6]) and not np.array_equal(d[5], orc.encrypt_draws(bytes(32), 5))
    u = d[:, 0].ravel()
    counts = np.array([(u == v).sum() for v in (-1, 0, 1)], dtype=np.float64)
    assert set(np.unique(u)) == {-1, 0, 1}
    assert ((counts - u.size / 3) ** 2 / (u.size / 3)).sum() < 20                 # chi-square, 2 degrees of freedom
    e = d[:, 1:].ravel().astype(np.int64)
    assert abs(e).max() <= 19 and abs(e.mean()) < 0.03
    assert abs(e.var() - (3.19 ** 2 + 1 / 12)) < 0.15
    cdt = [0] + om.noise_cdt() + [1 << 63]
    p_abs = np.diff(np.array(cdt, dtype=np.float64)) / 2.0 ** 63                    # P(|e| = m), m = 0..19
    chi = 0.0
    for m in range(12):                                                             # tails pooled
        exp = p_abs[m] * e.size
        chi += ((abs(e) == m).sum() - exp) ** 2 / exp
    exp = p_abs[12:].sum() * e.size
    chi += ((abs(e) >= 12).sum() - exp) ** 2 / exp
    assert chi < 45, chi                                                            # 12 degrees of freedom
    assert abs((e > 0).sum() - (e < 0).sum()) < 5 * np.sqrt(e.size)
    # an encryption formed from the draws decrypts, with the budget of a fresh ciphertext
    sk, pk = orc.keygen(seed=3)
    for v in (0.0, 0.71875, -3.5, 200.25):
        plain = orc.encode(v)
        ct = orc.encrypt_keyed(pk, plain, key, 9)
        assert np.array_equal(ct, orc.encrypt_with_draws(pk, plain, d[9]))
        got, budget = orc.decrypt(sk, ct)
        assert np.array_equal(got, plain) and budget > 60, budget
    assert not np.array_equal(orc.encrypt_keyed(pk, orc.encode(1.0), key, 1), orc.encrypt_keyed(pk, orc.encode(1.0), key, 2))


def test_oracle_backed_c_abi_encrypts_like_the_oracle(oracle_mod):
    """oracle/libfhe_cabi_oracle.so is what the facade's CPU builds link: its fhe_encrypt_batch / fhe_frac_encode_batch / fhe_encrypt_draws
    against the oracle entry points"""
    om = oracle_mod
    path = os.path.join(ROOT, "oracle", "libfhe_cabi_oracle.so")
    if not os.path.exists(path):
        pytest.skip("oracle/libfhe_cabi_oracle.so not built")
    L = C.CDLL(path)
    orc = om.Oracle.preset("SEAL23_2048")
    q = (C.c_uint64 * orc.k)(*[int(x) for x in orc.q])
    ctx = C.c_void_p()
    L.fhe_ctx_create.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
    assert L.fhe_ctx_create(orc.n, q, orc.k, orc.t, 0, C.byref(ctx)) == 0
    vp = C.c_void_p
    sk, pk = orc.keygen(seed=5)
    pk_ntt = np.ascontiguousarray(pk).copy()
    L.fhe_ntt_forward.argtypes = [vp, vp, vp, C.c_uint64, vp]
    assert L.fhe_ntt_forward(ctx, pk_ntt.ctypes.data_as(vp), pk_ntt.ctypes.data_as(vp), 2, None) == 0
    vals = np.array([0.0, 0.3125, -1.75, 17.0, 0.999], dtype=np.float64)
    plain = np.zeros((len(vals), orc.n), dtype=np.uint64)
    L.fhe_frac_encode_batch.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp]
    assert L.fhe_frac_encode_batch(ctx, vals.ctypes.data_as(vp), len(vals), 100, 100, plain.ctypes.data_as(vp), None) == 0
    for v, p in zip(vals, plain):
        assert np.array_equal(p, orc.encode(float(v)))
    key = bytes(range(7, 39))
    out = np.zeros((len(vals), 2, orc.k, orc.n), dtype=np.uint64)
    L.fhe_encrypt_scratch_bytes.restype = C.c_size_t
    L.fhe_encrypt_scratch_bytes.argtypes = [vp, C.c_uint64]
    nbytes = L.fhe_encrypt_scratch_bytes(ctx, len(vals))
    scratch = np.zeros(nbytes // 8 + 1, dtype=np.uint64)
    L.fhe_encrypt_batch.argtypes = [vp, vp, vp, C.c_uint64, C.c_char_p, C.c_uint64, vp, vp, C.c_size_t, vp]
    first = (1 << 40) + 3
    assert L.fhe_encrypt_batch(ctx, pk_ntt.ctypes.data_as(vp), plain.ctypes.data_as(vp), len(vals), key, first, out.ctypes.data_as(vp),
                               scratch.ctypes.data_as(vp), nbytes, None) == 0
    for i in range(len(vals)):
        assert np.array_equal(out[i], orc.encrypt_keyed(pk, plain[i], key, first + i))
        assert np.array_equal(orc.decrypt(sk, out[i])[0], plain[i])
    zero = np.zeros((2, 2, orc.k, orc.n), dtype=np.uint64)
    assert L.fhe_encrypt_batch(ctx, pk_ntt.ctypes.data_as(vp), None, 2, key, 0, zero.ctypes.data_as(vp), scratch.ctypes.data_as(vp), nbytes, None) == 0
    assert np.array_equal(zero[1], orc.encrypt_keyed(pk, np.zeros(orc.n, dtype=np.uint64), key, 1))
    assert L.fhe_encrypt_batch(ctx, pk_ntt.ctypes.data_as(vp), None, 2, key, 0, zero.ctypes.data_as(vp), scratch.ctypes.data_as(vp), 8, None) < 0
    # fhe_decrypt_batch of the shim = the oracle's decryption, with the raw noise figure
    sk_ntt = np.ascontiguousarray(sk).copy()
    assert L.fhe_ntt_forward(ctx, sk_ntt.ctypes.data_as(vp), sk_ntt.ctypes.data_as(vp), 1, None) == 0
    L.fhe_decrypt_scratch_bytes.restype = C.c_size_t
    L.fhe_decrypt_scratch_bytes.argtypes = [vp, C.c_uint32, C.c_uint64]
    L.fhe_decrypt_batch.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64, vp, vp, vp, C.c_size_t, vp]
    L.fhe_ctx_modulus_bits.restype = C.c_uint32
    L.fhe_ctx_modulus_bits.argtypes = [vp]
    dbytes = L.fhe_decrypt_scratch_bytes(ctx, 2, len(vals))
    dscr = np.zeros(dbytes // 8 + 1, dtype=np.uint64)
    got_plain = np.zeros((len(vals), orc.n), dtype=np.uint64)
    bits = np.zeros(len(vals), dtype=np.uint32)
    assert L.fhe_decrypt_batch(ctx, sk_ntt.ctypes.data_as(vp), out.ctypes.data_as(vp), 2, len(vals), got_plain.ctypes.data_as(vp), bits.ctypes.data_as(vp),
                               dscr.ctypes.data_as(vp), dbytes, None) == 0
    for i in range(len(vals)):
        want, nb, mb = orc.decrypt_noise_bits(sk, out[i])
        assert np.array_equal(got_plain[i], want) and np.array_equal(want, plain[i]) and bits[i] == nb and mb == L.fhe_ctx_modulus_bits(ctx)
    draws = np.zeros((3, 3, orc.n), dtype=np.int8)
    L.fhe_encrypt_draws.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_uint64, vp, vp]
    assert L.fhe_encrypt_draws(ctx, key, 11, 3, draws.ctypes.data_as(vp), None) == 0
    assert np.array_equal(draws[2], orc.encrypt_draws(key, 13))
    L.fhe_ctx_destroy.argtypes = [vp]
    L.fhe_ctx_destroy(ctx)
