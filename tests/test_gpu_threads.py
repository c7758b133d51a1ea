"""GPU: the threading contract of include/fhe_hip.h -- a context may be shared by host threads that issue calls on their own
streams; the only state built after fhe_ctx_create (the ct x ct tables, under std::call_once; the rgb constant cache, under
a mutex) is safe to race."""
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


def test_threads_race_the_first_multiply_on_a_fresh_context(fhe, oracle_mod):
    """the ct x ct tables are built by the first call that needs them (std::call_once inside the library): eight threads hit
    fhe_multiply / fhe_square on a context that has never multiplied, all at once; every product equals the oracle's, and a
    context that only runs the linear circuits never builds the tables at all"""
    import torch
    L = fhe._lib.load()
    orc = oracle_mod.Oracle.preset("P4096")
    for attempt in range(3):                                    # three fresh contexts: the race window is the first call only
        ctx = fhe.SEALContext(4096, fhe.PRESETS["P4096"]["q"], 1 << 14)
        assert L.fhe_ctx_has_ctct_tables(ctx.h) == 0
        plan = fhe.DctPlan(ctx, fhe.YQT)
        blk = ctx.random_ct(1, 64, seed=5)
        fhe.Evaluator(ctx).dct8x8_quant(plan, blk)
        r, g, b = (ctx.random_ct(2, size=2, seed=6 + i) for i in range(3))
        fhe.Evaluator(ctx).rgb_to_ycc(r, g, b)
        torch.cuda.synchronize()
        assert L.fhe_ctx_has_ctct_tables(ctx.h) == 0            # DCT + colour conversion: no auxiliary base was ever searched for
        n_threads = 8
        ins = [(ctx.random_ct(2, size=2, seed=400 + i), ctx.random_ct(2, size=2, seed=500 + i)) for i in range(n_threads)]
        torch.cuda.synchronize()
        out, errors, gate = [None] * n_threads, [], threading.Barrier(n_threads)

        def worker(i):
            try:
                ev = fhe.Evaluator(ctx)
                stream = torch.cuda.Stream()
                gate.wait()
                with torch.cuda.stream(stream):
                    p = ev.multiply(ins[i][0], ins[i][1]) if i % 2 == 0 else ev.square(ins[i][0])
                    stream.synchronize()
                out[i] = fhe.to_host(p)
            except Exception as exc:
                errors.append((i, repr(exc)))
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert L.fhe_ctx_has_ctct_tables(ctx.h) == 1
        for i in range(n_threads):
            a, b = fhe.to_host(ins[i][0]), fhe.to_host(ins[i][1])
            want = orc.multiply(a[0], b[0]) if i % 2 == 0 else orc.square(a[0])
            assert np.array_equal(out[i][0], want), (attempt, i)


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
