"""GPU: the threading contract of include/fhe_hip.h -- a context is immutable after fhe_ctx_create (ct x ct tables
included) and may be shared by host threads that issue calls on their own streams."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_threads_share_one_context_on_their_own_streams(fhe, oracle_mod):
    import torch
    ctx = fhe.SEALContext.preset("P4096")
    orc = oracle_mod.Oracle.preset("P4096")
    plan = fhe.DctPlan(ctx, fhe.YQT)
    n_threads, rounds = 4, 3
    inputs = [(ctx.random_ct(4, size=2, seed=100 + i), ctx.random_ct(4, size=2, seed=200 + i), ctx.random_ct(1, 64, seed=300 + i)) for i in range(n_threads)]
    torch.cuda.synchronize()
    results, errors = [None] * n_threads, []
    start = threading.Barrier(n_threads)            # the first multiply / rgb_to_ycc of every thread coincide

    def worker(i):
        try:
            ev = fhe.Evaluator(ctx)                 # per-thread evaluator (scratch buffers), shared context
            a, b, blk = inputs[i]
            stream = torch.cuda.Stream()
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(rounds):
                    prod = ev.multiply(a, b)
                    dct = ev.dct8x8_quant(plan, blk)
                    r, g, bl = a.clone(), b.clone(), a.clone()
                    ev.rgb_to_ycc(r, g, bl)
                stream.synchronize()
            results[i] = (fhe.to_host(prod), fhe.to_host(dct), fhe.to_host(r))
        except Exception as exc:                    # surfaced in the main thread
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, (prod, dct, y) in enumerate(results):
        a, b, blk = (fhe.to_host(x) for x in inputs[i])
        assert np.array_equal(prod[0], orc.multiply(a[0], b[0])), i
        assert np.array_equal(dct[0], orc.dct_quant(blk[0], fhe.YQT)), i
        assert np.array_equal(y[0], orc.rgb_to_ycc(a[0], b[0], a[0])[0]), i


def test_encryptions_use_fresh_randomness(fhe, oracle_mod):
    """keys.py without a seed samples from the OS CSPRNG: two encryptions of one plaintext differ, both decrypt"""
    ctx = fhe.SEALContext.preset("P4096")
    kg = fhe.KeyGenerator(ctx)
    enc = fhe.FractionalEncoder(ctx)
    er, dr = fhe.Encryptor(ctx, kg.public_key()), fhe.Decryptor(ctx, kg.secret_key())
    c1, c2 = er.encrypt(enc.encode(3.25)), er.encrypt(enc.encode(3.25))
    assert not np.array_equal(fhe.to_host(c1[None]), fhe.to_host(c2[None]))
    assert enc.decode(dr.decrypt(c1)) == 3.25 and enc.decode(dr.decrypt(c2)) == 3.25
    kg2 = fhe.KeyGenerator(ctx)
    assert not np.array_equal(fhe.to_host(kg.secret_key()[None]), fhe.to_host(kg2.secret_key()[None]))
