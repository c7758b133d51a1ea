"""GPU: the servers' own encryptions as device batches (include/fhe_hip.h fhe_encrypt_batch / fhe_frac_encode_batch / fhe_encrypt_draws,
csrc/encrypt.hip) against the oracle's restatement of the keyed sampler (oracle/fhe_oracle.c fo_encrypt_draws / fo_encrypt_keyed; pinned
on the CPU by tests/test_encrypt_sampler.py) -- bit for bit -- and through the streaming servers that use them.  The reference makes these
encryptions one seal::Encryptor::encrypt at a time inside its loops: homo/fhe_resize.h:230,234,262,266, homo/fhe_decode.h:54,134,
homo/server_decode.cpp:121,126."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KEY = bytes(range(32))


def _pair(fhe, om, preset):
    return fhe.SEALContext.preset(preset), om.Oracle.preset(preset)


@pytest.mark.parametrize("preset", ["P4096", "SEAL23_2048", "P8192"])
def test_draws_equal_the_oracle_restatement(fhe, oracle_mod, preset):
    ctx, orc = _pair(fhe, oracle_mod, preset)
    kg = fhe.KeyGenerator(ctx, seed=1)
    for key, first in ((KEY, 0), (bytes(32), 7), (os.urandom(32), (1 << 40) + 5), (KEY, (1 << 64) - 4)):
        der = fhe.DeviceEncryptor(ctx, kg.public_key(), key=key)
        got = der.draws(first, 3)
        for i in range(3):
            assert np.array_equal(got[i], orc.encrypt_draws(key, first + i)), (preset, first, i)


def test_device_encoder_equals_host_encoder(fhe, oracle_mod):
    ctx = fhe.SEALContext.preset("P4096")
    rng = np.random.default_rng(3)
    vals = np.concatenate([rng.uniform(-1, 1, 200), rng.uniform(-300, 300, 200), rng.integers(-1000, 1000, 50).astype(np.float64),
                           [0.0, -0.0, 0.5, -0.5, 2.0 ** -60, -2.0 ** -99, 2.0 ** -100, 2.0 ** -101, 1 - 2.0 ** -53, 255.999999, 2.0 ** 62, -2.0 ** 62, 3.0e18,
                            0.1, 1 / 3, -1 / 3, 0.7071067811865476, 123456789.123456789, 5e-324, 2.0 ** -1074 * 3]])
    for ic, fc in ((100, 100), (64, 32), (10, 1200)):
        use = vals if ic >= 64 else vals[np.abs(vals) < 2.0 ** ic]
        fe = fhe.FractionalEncoder(ctx, ic, fc)
        want = np.stack([fe.encode(float(v)) for v in use])
        got = torch.empty((len(use), ctx.n), dtype=torch.int64, device=ctx.device)
        fhe._lib.call("fhe_frac_encode_batch", ctx.h, np.ascontiguousarray(use).ctypes.data_as(C.c_void_p), len(use), ic, fc, C.c_void_p(got.data_ptr()), None)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint64), want), (ic, fc)
    bad = torch.empty((1, ctx.n), dtype=torch.int64, device=ctx.device)
    for v, ic in ((float("nan"), 100), (float("inf"), 100), (1.0e19, 100), (1024.0, 10)):       # the host encoder's refusals
        arr = np.array([v])
        with pytest.raises(fhe.FheError):
            fhe._lib.call("fhe_frac_encode_batch", ctx.h, arr.ctypes.data_as(C.c_void_p), 1, ic, 10, C.c_void_p(bad.data_ptr()), None)
    # more values than one staging slot holds (32768 doubles): two slots, same result
    many = rng.uniform(-4, 4, 40000)
    small = fhe.SEALContext.preset("SEAL23_2048")
    got = torch.empty((len(many), small.n), dtype=torch.int64, device=small.device)
    fhe._lib.call("fhe_frac_encode_batch", small.h, many.ctypes.data_as(C.c_void_p), len(many), 100, 100, C.c_void_p(got.data_ptr()), None)
    fe = fhe.FractionalEncoder(small)
    for i in (0, 1, 32767, 32768, 39999):
        assert np.array_equal(got[i].cpu().numpy().view(np.uint64), fe.encode(float(many[i]))), i


@pytest.mark.parametrize("preset", ["P4096", "SEAL23_4096", "SEAL23_2048", "P8192", "SEAL3_8192", "SEAL23_16384"])
def test_fused_encryption_kernel_equals_the_five_launches(fhe, preset):
    """fhe_encrypt_batch as ONE launch (k_enc_fused: draws, forward transform, both key products, the pair of inverse transforms, noise
    and Delta m' per workgroup; pseudo-Mersenne arithmetic on SEAL's 54 / 55-bit primes, lazy Shoup arithmetic on the 36 / 43-bit sets),
    built for two and for four waves per SIMD (FHE_ENC_OCC=4), against the five launches of round 5 (FHE_ENC_UNFUSED=1): the same
    (key, index) stream, the same exact arithmetic, the same bits -- 70 ciphertexts (not a multiple of anything) with plaintexts in
    both halves of [0, t), and encryptions of zero."""
    ctx = fhe.SEALContext.preset(preset)
    kg = fhe.KeyGenerator(ctx, seed=5)
    vals = np.linspace(-300.0, 300.0, 70)
    outs, zeros = [], []
    for sw in ({}, {"FHE_ENC_OCC": 4}, {"FHE_ENC_UNFUSED": 1}):
        c2 = fhe.SEALContext(ctx.n, ctx.q, ctx.t, switches=sw) if sw else ctx
        der = fhe.DeviceEncryptor(c2, kg.public_key(), key=KEY, reproducible=True)
        der.seek(1 << 40)
        outs.append(der.encrypt_values(vals))
        zeros.append(der.encrypt_zeros(3))
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])
    assert torch.equal(zeros[0], zeros[2]) and torch.equal(zeros[1], zeros[2])
    dec, fe = fhe.Decryptor(ctx, kg.secret_key()), fhe.FractionalEncoder(ctx)
    plains, budgets = dec.decrypt_batch(outs[0][:4], with_budget=True)
    assert [fe.decode(p) for p in plains] == [fe.decode(fe.encode(v)) for v in vals[:4]] and min(budgets) > 20


@pytest.mark.parametrize("preset", ["P4096", "SEAL23_4096", "SEAL23_2048", "P8192", "SEAL3_8192", "SEAL23_16384"])
def test_encrypt_batch_equals_the_oracle_bit_for_bit(fhe, oracle_mod, preset):
    ctx, orc = _pair(fhe, oracle_mod, preset)
    kg = fhe.KeyGenerator(ctx, seed=11)
    pk, sk = fhe.to_host(kg.public_key()), fhe.to_host(kg.secret_key())
    vals = [0.0, 0.40625, -0.40625, 0.9990234375, 77.0, -200.75, 1 / 3, 2.0 ** -30]
    first = (1 << 33) + 9
    der = fhe.DeviceEncryptor(ctx, kg.public_key(), key=KEY, reproducible=True)
    der.seek(first)
    got = fhe.to_host(der.encrypt_values(vals))
    assert der.next == first + len(vals)
    dec = fhe.Decryptor(ctx, kg.secret_key())
    fe = fhe.FractionalEncoder(ctx)
    for i, v in enumerate(vals):
        plain = orc.encode(v)
        assert np.array_equal(got[i], orc.encrypt_keyed(pk, plain, KEY, first + i)), (preset, i)
        p, budget = orc.decrypt(sk, got[i])
        assert np.array_equal(p, plain) and budget > 20
        assert fe.decode(dec.decrypt(fhe.to_device(got[i]))) == fe.decode(plain)
    # batches are independent of how they are cut, and of what else ran in between
    der.seek(first + 2)
    again = fhe.to_host(der.encrypt_values(vals[2:5]))
    assert np.array_equal(again, got[2:5])
    # encryptions of zero (no plaintext array) and explicit plaintext arrays
    der.seek(5)
    z = fhe.to_host(der.encrypt_zeros(3))
    zero = np.zeros(ctx.n, dtype=np.uint64)
    for i in range(3):
        assert np.array_equal(z[i], orc.encrypt_keyed(pk, zero, KEY, 5 + i))
    rng = np.random.default_rng(5)
    plains = rng.integers(0, ctx.t, size=(2, ctx.n), dtype=np.uint64)                   # dense plaintexts, both halves of [0, t)
    der.seek(100)
    d = fhe.to_host(der.encrypt_plains(torch.from_numpy(plains.view(np.int64)).to(ctx.device)))
    for i in range(2):
        assert np.array_equal(d[i], orc.encrypt_keyed(pk, plains[i], KEY, 100 + i))
        assert np.array_equal(orc.decrypt(sk, d[i])[0], plains[i])


def test_encrypt_argument_errors_and_key_discipline(fhe):
    ctx = fhe.SEALContext.preset("SEAL23_2048")
    kg = fhe.KeyGenerator(ctx, seed=2)
    der = fhe.DeviceEncryptor(ctx, kg.public_key())                 # key from the OS generator
    with pytest.raises(RuntimeError):
        der.seek(0)                                                  # would allow a (key, index) pair to repeat
    a, b = der.encrypt_zeros(2), der.encrypt_zeros(2)
    assert der.next == 4 and not torch.equal(a, b) and not torch.equal(a[0], a[1])
    assert fhe.DeviceEncryptor(ctx, kg.public_key()).key != der.key
    keyed = fhe.DeviceEncryptor(ctx, kg.public_key(), key=KEY)      # a caller's own key WITHOUT reproducible=True: counts upwards only
    keyed.encrypt_zeros(3)
    with pytest.raises(RuntimeError):
        keyed.seek(0)
    assert keyed.next == 3
    with pytest.raises(ValueError):
        fhe.DeviceEncryptor(ctx, kg.public_key(), reproducible=True)    # reproducible needs an explicit key
    with pytest.raises(ValueError):
        fhe.DeviceEncryptor(ctx, kg.public_key(), key=b"short")
    out = ctx.empty(1)
    pkn = der._pk_ntt
    small = torch.empty(8, dtype=torch.int64, device=ctx.device)
    with pytest.raises(fhe.FheError):                                # scratch too small
        fhe._lib.call("fhe_encrypt_batch", ctx.h, C.c_void_p(pkn.data_ptr()), None, 1, der.key, 0, C.c_void_p(out.data_ptr()), C.c_void_p(small.data_ptr()), 64, None)
    with pytest.raises(fhe.FheError):                                # the index must not wrap
        big = torch.empty(2 * ctx.k * ctx.n, dtype=torch.int64, device=ctx.device)
        fhe._lib.call("fhe_encrypt_batch", ctx.h, C.c_void_p(pkn.data_ptr()), None, 2, der.key, (1 << 64) - 1, C.c_void_p(ctx.empty(2).data_ptr()),
                      C.c_void_p(big.data_ptr()), big.numel() * 8, None)


def test_servers_with_device_encryptions(fhe, oracle_mod, tmp_path):
    """server_resize with the device encryptor: a seeded run equals the circuit called directly on the same encryptions, two row shards write
    the whole run's bytes, and the result decrypts to the interpolated image; server_decode the same with its zeros"""
    ctx = fhe.SEALContext.preset("P4096")
    kg = fhe.KeyGenerator(ctx, seed=21)
    dec, fe, ev = fhe.Decryptor(ctx, kg.secret_key()), fhe.FractionalEncoder(ctx), fhe.Evaluator(ctx)
    W, H, w, h = 6, 5, 4, 3
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(H, W, 3))
    client = fhe.DeviceEncryptor(ctx, kg.public_key(), key=bytes(32))
    pix = client.encrypt_values(img.reshape(-1).astype(np.float64))             # [H * W * 3, 2, k, n], the stream's order
    fin, fout, fsh = str(tmp_path / "in.ct"), str(tmp_path / "out.ct"), str(tmp_path / "shards.ct")
    with open(fin, "wb") as f:
        for c in fhe.to_host(pix):
            fhe.server.write_ciphertext(f, c)
    enc = fhe.server.make_fraction_encryptor(ctx, kg.public_key(), seed=9, device=True)
    assert hasattr(enc, "seek")
    assert fhe.server.server_resize(ctx, fin, fout, W, H, w, h, False, enc, rows_per_step=2) == w * h
    whole = open(fout, "rb").read()
    for rows in ((0, 2), (2, 3)):
        fhe.server.server_resize(ctx, fin, fsh, W, H, w, h, False, fhe.server.make_fraction_encryptor(ctx, kg.public_key(), seed=9, device=True), rows_per_step=1, rows=rows)
    assert open(fsh, "rb").read() == whole
    # the circuit called directly with the same encryptions
    taps, xs, ys = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=False)
    direct = fhe.server.make_fraction_encryptor(ctx, kg.public_key(), seed=9, device=True)
    fr = direct([v for pair in zip(xs, ys) for v in pair])
    pc = fhe.circuits.PlainCache(ctx)
    rec = fhe.server.RECORD_HEADER + 4 * ctx.k * ctx.n * 8
    for ch in range(3):
        out = fhe.to_host(fhe.circuits.sample_linear(ev, pc, pix, np.asarray(taps, dtype=np.uint32) * 3 + ch, fr[0::2].contiguous(), fr[1::2].contiguous()))
        for p in (0, 5, w * h - 1):
            off = (p * 3 + ch) * rec + fhe.server.RECORD_HEADER
            assert np.array_equal(np.frombuffer(whole[off:off + rec - fhe.server.RECORD_HEADER], dtype=np.uint64).reshape(4, ctx.k, ctx.n), out[p])
            # and the value: bilinear interpolation of the plain image at this pixel
            flat = img.reshape(-1, 3)
            q = [[float(flat[taps[p][0], ch]), float(flat[taps[p][1], ch])], [float(flat[taps[p][2], ch]), float(flat[taps[p][3], ch])]]   # p00 p10 / p01 p11
            fx, fy = float(xs[p]), float(ys[p])
            want = (q[0][0] * (1 - fx) + q[0][1] * fx) * (1 - fy) + (q[1][0] * (1 - fx) + q[1][1] * fx) * fy
            assert abs(fe.decode(dec.decrypt(fhe.to_device(out[p]))) - want) < 1e-3
    # a production encryptor (key from the OS) has no seek and never repeats itself
    prod = fhe.server.make_fraction_encryptor(ctx, kg.public_key())
    assert not hasattr(prod, "seek")
    assert fhe.server.server_resize(ctx, fin, fsh, W, H, w, h, False, prod, rows_per_step=2) == w * h
    assert open(fsh, "rb").read() != whole
    zeros = fhe.server.make_zero_encryptor(ctx, kg.public_key(), seed=4, device=True)
    zeros.seek(3)
    z = zeros(2)
    assert [fe.decode(dec.decrypt(c)) for c in z] == [0.0, 0.0] and not torch.equal(z[0], z[1])
    zeros.seek(4)
    assert torch.equal(zeros(1)[0], z[1])


@pytest.mark.parametrize("preset", ["P4096", "SEAL23_2048", "SEAL23_4096", "P8192", "SEAL3_8192"])
def test_decrypt_batch_equals_the_oracle_big_integer_decryption(fhe, oracle_mod, preset):
    """fhe_decrypt_batch (phase by Horner per slot + the exact rounding with multi-word integers on the device, csrc/encrypt.hip k_dec_round)
    against the oracle's CRT composition and big-integer rounding (oracle/fhe_oracle.c fo_decrypt): plaintexts AND the bit length of the
    invariant noise, for fresh ciphertexts, products (sizes 3 and 5) and uniformly random residues (far beyond the noise budget: the
    rounding is exact there too); k = 1 .. 5 primes"""
    ctx, orc = _pair(fhe, oracle_mod, preset)
    kg = fhe.KeyGenerator(ctx, seed=31)
    sk = fhe.to_host(kg.secret_key())
    dec, ev = fhe.Decryptor(ctx, kg.secret_key()), fhe.Evaluator(ctx)
    der = fhe.DeviceEncryptor(ctx, kg.public_key(), key=KEY)
    vals = [0.0, 1.5, -2.25, 0.40625, 100.0, -0.001953125]
    fresh = der.encrypt_values(vals)
    qbits = int(fhe._lib.load().fhe_ctx_modulus_bits(ctx.h))

    def check(cts, label):
        plains, budgets = dec.decrypt_batch(cts, with_budget=True)
        host = fhe.to_host(cts)
        for i in range(host.shape[0]):
            want, nb, mb = orc.decrypt_noise_bits(sk, host[i])
            assert mb == qbits
            assert np.array_equal(plains[i], want), (preset, label, i)
            assert budgets[i] == max(0, mb - nb - 1), (preset, label, i, budgets[i], nb)
        return plains, budgets

    plains, budgets = check(fresh, "fresh")
    fe = fhe.FractionalEncoder(ctx)
    assert [fe.decode(p) for p in plains] == vals and min(budgets) > 20
    one, b1 = dec.decrypt(fresh[2], with_budget=True)                       # the single-ciphertext form and the host big-integer form agree
    hostp, hb = dec.decrypt_host(fresh[2], with_budget=True)
    assert np.array_equal(one, plains[2]) and np.array_equal(hostp, one) and b1 == hb == budgets[2]
    if ctx.k >= 2:                                                          # a product needs room: not the one-prime set
        prod = ev.multiply(fresh[:3].contiguous(), fresh[3:6].contiguous())
        assert prod.shape[1] == 3
        p3, _ = check(prod, "size 3")
        assert [fe.decode(p) for p in p3] == [vals[i] * vals[i + 3] for i in range(3)]
        check(ev.multiply(prod, prod), "size 5")                           # whatever the budget says, the same bits as the oracle
    noise = ctx.random_ct(3, size=2, seed=77)                              # uniformly random residues: no budget at all
    _, b = check(noise, "random")
    assert b == [0, 0, 0]
    check(ctx.random_ct(2, size=4, seed=78), "random size 4")
    assert dec.decrypt_batch(ctx.empty(0)).shape == (0, ctx.n)
