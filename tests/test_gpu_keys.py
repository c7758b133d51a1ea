"""GPU: the product's own KeyGenerator / Encryptor / Decryptor (C ABI underneath) interoperate with
the CPU oracle, and the relinearised Cubic mode decrypts to the same value as the reference path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
Q3 = [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001]


def _pair(fhe, om, n=2048):
    return fhe.SEALContext(n, Q3, 1 << 14), om.Oracle(n, Q3, 1 << 14)


def test_product_keys_interoperate_with_oracle(fhe, oracle_mod):
    ctx, orc = _pair(fhe, oracle_mod)
    kg = fhe.KeyGenerator(ctx, seed=1)
    enc, dec = fhe.Encryptor(ctx, kg.public_key(), seed=2), fhe.Decryptor(ctx, kg.secret_key())
    fe = fhe.FractionalEncoder(ctx)
    sk_host, pk_host = fhe.to_host(kg.secret_key()), fhe.to_host(kg.public_key())
    for v in (37.25, -255.75, 0.541196100, 0.0):
        ct = enc.encrypt(fe.encode(v))
        plain, budget = dec.decrypt(ct, with_budget=True)
        assert fe.decode(plain) == v and budget > 50
        # oracle decrypts the product's ciphertext with the product's key
        p2, b2 = orc.decrypt(sk_host, fhe.to_host(ct))
        assert np.array_equal(plain, p2) and budget == b2
        # and the product decrypts the oracle's ciphertext under the same public key
        oc = orc.encrypt(pk_host, orc.encode(v), seed=9)
        assert fe.decode(dec.decrypt(fhe.to_device(oc))) == v


def test_product_evaluation_keys_relinearize(fhe, oracle_mod):
    ctx, orc = _pair(fhe, oracle_mod)
    kg = fhe.KeyGenerator(ctx, seed=3)
    enc, dec = fhe.Encryptor(ctx, kg.public_key(), seed=4), fhe.Decryptor(ctx, kg.secret_key())
    fe, ev = fhe.FractionalEncoder(ctx), fhe.Evaluator(ctx)
    evk = kg.generate_evaluation_keys(30)
    a, b = enc.encrypt(fe.encode(3.5))[None], enc.encrypt(fe.encode(-2.25))[None]
    prod = ev.multiply(a.contiguous(), b.contiguous())
    assert prod.shape[-3] == 3
    rl = ev.relinearize(prod, evk, 30)
    assert rl.shape[-3] == 2
    assert fe.decode(dec.decrypt(rl[0])) == 3.5 * -2.25
    assert dec.invariant_noise_budget(rl[0]) > 0


def test_relinearised_cubic_mode(fhe, oracle_mod):
    """Cubic with every product relinearised under the PRODUCT's own evaluation keys: the library's relinearised mode
    (fhe_circuits_create_relin) equals the oracle's op-by-op composition multiply -> relinearize BIT FOR BIT (the keys go to
    the oracle through coefficient form: each side transforms them into its own slot order), decrypts to the closed form,
    and keeps more noise budget than the reference's size-4 result"""
    import torch
    ctx, orc = _pair(fhe, oracle_mod, n=4096)
    kg = fhe.KeyGenerator(ctx, seed=5)
    enc, dec = fhe.Encryptor(ctx, kg.public_key(), seed=6), fhe.Decryptor(ctx, kg.secret_key())
    fe, ev = fhe.FractionalEncoder(ctx), fhe.Evaluator(ctx)
    pc = fhe.circuits.PlainCache(ctx)
    evk = kg.generate_evaluation_keys(30)
    coeff = fhe.to_host(ev.ntt_inverse(evk))
    evk_orc = np.zeros_like(coeff)
    for idx in np.ndindex(coeff.shape[:3]):
        for i in range(ctx.k):
            evk_orc[idx + (i,)] = orc.ntt_fwd(coeff[idx + (i,)], i)
    rorc = oracle_mod.RelinOracle(orc, evk_orc, 30)
    A, B, C_, D, t = 10.0, 50.0, 90.0, 40.0, 0.25
    cA, cB, cC, cD, ct = (enc.encrypt(fe.encode(v))[None].contiguous() for v in (A, B, C_, D, t))
    ref = fhe.circuits.cubic(ev, pc, cA, cB, cC, cD, ct)
    rel = fhe.circuits.cubic(ev, pc, cA, cB, cC, cD, ct, relin=(evk, 30))
    assert ref.shape[-3] == 4 and rel.shape[-3] == 2
    h = fhe.to_host
    assert np.array_equal(h(rel)[0], oracle_mod.oracle_cubic_calls(rorc, h(cA)[0], h(cB)[0], h(cC)[0], h(cD)[0], h(ct)[0]))
    assert torch.equal(rel, fhe.circuits.cubic_evaluator_calls(ev, pc, cA, cB, cC, cD, ct, (evk, 30)))
    a, b, c = -A + 3 * B - 3 * C_ + D, 2 * A - 5 * B + 4 * C_ - D, C_ - A
    expect = 0.5 * (a * t * t + b * t * t + c * t) + B
    assert fe.decode(dec.decrypt(ref[0])) == expect == fe.decode(dec.decrypt(rel[0]))      # dyadic rationals: exact
    b_ref, b_rel = dec.invariant_noise_budget(ref[0]), dec.invariant_noise_budget(rel[0])
    print("\n[relin n=4096 Q3 dbc=30] Cubic noise budget left: reference mode %d bits, relinearised %d bits" % (b_ref, b_rel))
    assert b_rel > 0 and b_ref > 0


def test_keys_of_another_context_are_refused(fhe):
    """Encryptor / DeviceEncryptor / Decryptor read their key with the context's strides: a key of another degree or another number of
    primes, a host tensor or a wrong polynomial count is an error at construction, not a read behind the allocation"""
    ctx, other = fhe.SEALContext.preset("SEAL23_4096"), fhe.SEALContext.preset("SEAL23_2048")
    kg, kg2 = fhe.KeyGenerator(ctx, seed=1), fhe.KeyGenerator(other, seed=1)
    for make in (lambda: fhe.Encryptor(ctx, kg2.public_key()), lambda: fhe.DeviceEncryptor(ctx, kg2.public_key()), lambda: fhe.Decryptor(ctx, kg2.secret_key()),
                 lambda: fhe.DeviceEncryptor(ctx, kg.secret_key()), lambda: fhe.Decryptor(ctx, kg.public_key()), lambda: fhe.DeviceEncryptor(ctx, kg.public_key().cpu())):
        with pytest.raises(ValueError, match="another context"):
            make()
    fhe.DeviceEncryptor(ctx, kg.public_key()), fhe.Encryptor(ctx, kg.public_key()), fhe.Decryptor(ctx, kg.secret_key())
