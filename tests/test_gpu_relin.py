"""GPU: the RELINEARISED mode of the circuits (include/fhe_circuits.h fhe_circuits_create_relin).

The reference multiplies without ever relinearising (homo/fhe_resize.h:174-179, homo/fhe_decode.h:67-98,235,239) but carries
the decomposition bit count it would need (homo/client_resize.cpp:26,47,72; DBC = 30, homo/fhe_image.h:28).  The mode is the
reference's Evaluator call sequence with evaluator.relinearize after every multiply / square; the checker is the oracle's
op-by-op composition fo_multiply -> fo_relinearize3 (oracle/oracle.py RelinOracle), compared bit for bit (integer work:
array_equal, no tolerance), plus decrypt known answers with the remaining noise budget printed beside the reference mode's.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SMALL = dict(n=1024, q=[0xFFFFEE001, 0xFFFFC4001], t=1 << 14)


def _setup(fhe, om, name, dbc, key_seed=77):
    """context + oracle + ONE set of evaluation keys in both libraries' NTT slot orders"""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    p = SMALL if name == "SMALL" else om.PRESETS[name]
    ctx, orc = fhe.SEALContext(p["n"], p["q"], p["t"]), om.Oracle(p["n"], p["q"], p["t"])
    sk, pk = orc.keygen(key_seed)
    evk = orc.evk_gen(sk, dbc=dbc)                       # oracle NTT form [k][nd][2][k][n]
    coeff = np.zeros_like(evk)
    for idx in np.ndindex(evk.shape[:3]):
        for i in range(ctx.k):
            coeff[idx + (i,)] = orc.ntt_inv(evk[idx + (i,)], i)
    evk_dev = fhe.Evaluator(ctx).ntt_forward(fhe.to_device(coeff)).contiguous()      # library slot order
    return ctx, orc, om.RelinOracle(orc, evk, dbc), (evk_dev, dbc), sk, pk


@pytest.mark.parametrize("preset,dbc", [("SMALL", 16), ("P8192", 30), ("SEAL23_4096", 30), ("P8192", 60)])
def test_cubic_and_linear_relin_vs_oracle(fhe, oracle_mod, preset, dbc):
    """Cubic / Linear with every product relinearised: level 1 on random size-2 ciphertexts, level 2 on level-1 results
    (what SampleBicubic / SampleLinear feed their column evaluation), a batch that is not a multiple of anything, one
    operand at q - 1 everywhere; fhe_cubic (library) == cubic_evaluator_calls (one C-ABI call per Evaluator call) == oracle"""
    import torch
    ctx, orc, rorc, relin, _, _ = _setup(fhe, oracle_mod, preset, dbc)
    ev, pc, h = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx), fhe.to_host
    A, B, C, D = (ctx.random_ct(3, size=2, seed=400 + i) for i in range(4))
    t = ctx.random_ct(3, size=2, seed=410)
    A[1] = torch.tensor([q - 1 for q in ctx.q], dtype=torch.int64, device=A.device).view(1, ctx.k, 1).expand(2, ctx.k, ctx.n)
    r1 = fhe.circuits.cubic(ev, pc, A, B, C, D, t, relin=relin)
    assert r1.shape[-3] == 2
    assert torch.equal(r1, fhe.circuits.cubic_evaluator_calls(ev, pc, A, B, C, D, t, relin))
    want1 = [oracle_mod.oracle_cubic_calls(rorc, h(A)[i], h(B)[i], h(C)[i], h(D)[i], h(t)[i]) for i in range(3)]
    for i in range(3):
        assert np.array_equal(h(r1)[i], want1[i]), i
    # level 2: the column Cubic of SampleBicubic takes four row results
    r2 = fhe.circuits.cubic(ev, pc, r1, B, r1, D, t, relin=relin)
    assert r2.shape[-3] == 2
    assert np.array_equal(h(r2)[2], oracle_mod.oracle_cubic_calls(rorc, want1[2], h(B)[2], want1[2], h(D)[2], h(t)[2]))
    l1 = fhe.circuits.linear(ev, pc, A, B, t, relin=relin)
    l2 = fhe.circuits.linear(ev, pc, l1, r1, t, relin=relin)
    assert l1.shape[-3] == 2 and l2.shape[-3] == 2
    for i in (0, 1):
        o1 = oracle_mod.oracle_linear_calls(rorc, h(A)[i], h(B)[i], h(t)[i])
        assert np.array_equal(h(l1)[i], o1)
        assert np.array_equal(h(l2)[i], oracle_mod.oracle_linear_calls(rorc, o1, want1[i], h(t)[i]))
    # the reference's mode on the same handle family is untouched by the existence of a relinearising handle
    assert fhe.circuits.cubic(ev, pc, A, B, C, D, t).shape[-3] == 4
    # argument errors: the relinearised Cubic takes size-2 operands only
    A4 = ctx.random_ct(3, size=4, seed=1)
    with pytest.raises(fhe._lib.FheError):
        fhe.circuits.cubic(ev, pc, A4, A4, A4, A4, t, relin=relin)


@pytest.mark.parametrize("preset,dbc", [("SMALL", 16), ("P8192", 30)])
def test_samplers_relin_vs_oracle_and_decrypt(fhe, oracle_mod, preset, dbc):
    """SampleBicubic / SampleLinear / the shared-offset ResizeImage in the relinearised mode on an 8x8 -> 4x4 image of real
    encryptions: library == oracle composition bit for bit on sampled pixels, every output decrypts to the closed form, and
    the remaining noise budget is reported beside the reference mode's"""
    ctx, orc, rorc, relin, sk, pk = _setup(fhe, oracle_mod, preset, dbc, key_seed=21)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    W = H = 8
    w = h_ = 4
    vals = [float((29 * x + 53 * y) % 256) for y in range(H) for x in range(W)]
    pix = np.stack([orc.encrypt(pk, orc.encode(v), seed=500 + i) for i, v in enumerate(vals)])
    d_pix = fhe.to_device(pix)

    def plain_cubic(A, B, C, D, t):     # closed form of homo/fhe_resize.h:149-185 with t3 = t*t
        a, b, c = -A + 3 * B - 3 * C + D, 2 * A - 5 * B + 4 * C - D, C - A
        return 0.5 * (a * t * t + b * t * t + c * t) + B

    taps, fx, fy = fhe.circuits.resize_sample_plan(W, H, w, h_, bicubic=True)
    xf = np.stack([orc.encrypt(pk, orc.encode(f), seed=600 + i) for i, f in enumerate(fx)])
    yf = np.stack([orc.encrypt(pk, orc.encode(f), seed=700 + i) for i, f in enumerate(fy)])
    out = fhe.to_host(fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, fhe.to_device(xf), fhe.to_device(yf), relin=relin))
    ref_mode = fhe.to_host(fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, fhe.to_device(xf), fhe.to_device(yf)))
    assert out.shape == (w * h_, 2, ctx.k, ctx.n) and ref_mode.shape[1] == 6
    picks = (0, 5, 15) if preset == "SMALL" else (5,)
    for o in picks:
        assert np.array_equal(out[o], oracle_mod.oracle_sample_bicubic_calls(rorc, [pix[i] for i in taps[o]], xf[o], yf[o])), o
    budgets = []
    decrypts = preset != "SMALL"           # n = 1024 with a 72-bit q has no noise budget for two levels of products: bits only there
    for o in range(w * h_ if decrypts else 0):
        v = [vals[i] for i in taps[o]]
        cols = [plain_cubic(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3], fx[o]) for r in range(4)]
        expect = plain_cubic(cols[0], cols[1], cols[2], cols[3], fy[o])
        plain, budget = orc.decrypt(sk, out[o])
        p_ref, b_ref = orc.decrypt(sk, ref_mode[o])
        assert budget > 0 and abs(orc.decode(plain) - expect) < 1e-6 and abs(orc.decode(p_ref) - expect) < 1e-6
        budgets.append((budget, b_ref))
    if decrypts:
        print("\n[relin %s dbc=%d] SampleBicubic noise budget left: relinearised min %d bits, reference mode min %d bits"
              % (preset, dbc, min(b for b, _ in budgets), min(b for _, b in budgets)))
    # shared offsets (one ciphertext per output column / row): every output equals the per-pixel sampler with xfract[x], yfract[y]
    import torch
    xs, ys = fhe.to_device(xf[:w].copy()), fhe.to_device(yf[::w].copy())
    shared = fhe.circuits.resize_bicubic_shared(ev, pc, d_pix, W, H, w, h_, xs, ys, batch=8, band_rows=2, relin=relin)
    per_px = fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, xs.repeat(h_, 1, 1, 1).contiguous(), ys.repeat_interleave(w, dim=0).contiguous(), relin=relin)
    assert shared.shape[-3] == 2 and torch.equal(shared, per_px)
    # a shard of the destination rows == the same rows of the whole image
    first, cnt = fhe.circuits.resize_source_rows(H, h_, 1, 3)
    part = fhe.circuits.resize_bicubic_shared(ev, pc, d_pix[first * W:(first + cnt) * W].contiguous(), W, H, w, h_, xs, ys[1:3].contiguous(), batch=8, band_rows=2,
                                              rows=(1, 3), src_rows=(first, cnt), relin=relin)
    assert torch.equal(part, shared[w:3 * w])
    # bilinear
    taps4, fx4, fy4 = fhe.circuits.resize_sample_plan(W, H, w, h_, bicubic=False)
    xf4 = np.stack([orc.encrypt(pk, orc.encode(f), seed=800 + i) for i, f in enumerate(fx4)])
    yf4 = np.stack([orc.encrypt(pk, orc.encode(f), seed=900 + i) for i, f in enumerate(fy4)])
    lin = fhe.to_host(fhe.circuits.sample_linear(ev, pc, d_pix, taps4, fhe.to_device(xf4), fhe.to_device(yf4), relin=relin))
    assert lin.shape[1] == 2
    for o in picks:
        assert np.array_equal(lin[o], oracle_mod.oracle_sample_linear_calls(rorc, [pix[i] for i in taps4[o]], xf4[o], yf4[o])), o
    for o in range(w * h_ if decrypts else 0):
        v = [vals[i] for i in taps4[o]]
        c0, c1 = (1 - fx4[o]) * v[0] + fx4[o] * v[1], (1 - fx4[o]) * v[2] + fx4[o] * v[3]
        plain, budget = orc.decrypt(sk, lin[o])
        assert budget > 0 and abs(orc.decode(plain) - ((1 - fy4[o]) * c0 + fy4[o] * c1)) < 1e-6


@pytest.mark.parametrize("preset,dbc", [("SMALL", 16), ("P8192", 30), ("SEAL23_4096", 30)])
def test_sincos_relin_vs_oracle(fhe, oracle_mod, preset, dbc):
    """homomorphic_sin / cos with every power relinearised where it is formed: sizes stay 2 (11 in the reference's mode)"""
    ctx, orc, rorc, relin, _, _ = _setup(fhe, oracle_mod, preset, dbc)
    ev, pc, h = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx), fhe.to_host
    x, z = ctx.random_ct(3, size=2, seed=800), ctx.random_ct(3, size=2, seed=801)
    s = fhe.circuits.homomorphic_sin(ev, pc, x, z, relin=relin)
    c = fhe.circuits.homomorphic_cos(ev, pc, x, z, relin=relin)
    assert s.shape[-3] == 2 and c.shape[-3] == 2
    for i in ((0, 2) if preset != "SMALL" else (0, 1, 2)):
        assert np.array_equal(h(s)[i], oracle_mod.oracle_homomorphic_sin(rorc, h(x)[i], h(z)[i])), i
        assert np.array_equal(h(c)[i], oracle_mod.oracle_homomorphic_cos(rorc, h(x)[i], h(z)[i])), i


@pytest.mark.parametrize("preset,dbc,npos,degree", [("SMALL", 16, 3, 2), ("P8192", 30, 2, 2)])
def test_approximated_step_and_decode_channel_relin_vs_oracle(fhe, oracle_mod, preset, dbc, npos, degree):
    """approximated_step in the relinearised mode (the sin x cos product is 2 x 2 instead of 11 x 11, results have 2 polynomials
    instead of 22) against the oracle; a shard of the positions == the whole run; decode_channel over two runs == the
    composition of its steps; degree 0 (no harmonics) as well"""
    import torch
    ctx, orc, rorc, relin, _, _ = _setup(fhe, oracle_mod, preset, dbc)
    ev, pc, h = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx), fhe.to_host
    amp, idx, cnt = (ctx.random_ct(1, size=2, seed=900 + i) for i in range(3))
    zeros = ctx.random_ct(npos * degree * 2, size=2, seed=920).reshape(npos, degree, 2, 2, ctx.k, ctx.n)
    hz = h(zeros)
    run = fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, order=64, degree=degree, delta=0.5, width=npos, height=1, zeros=zeros, relin=relin)
    ref = oracle_mod.oracle_approximated_step(rorc, h(amp)[0], h(idx)[0], h(cnt)[0], 64, degree, 0.5, npos, 1,
                                              lambda i, j, wh: hz[i, j - 1, int(wh == "cos")])
    assert len(run) == npos
    for g, r in zip(run, ref):
        assert g.shape[-3] == 2 and np.array_equal(h(g)[0], r)
    part = fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, order=64, degree=degree, delta=0.5, width=npos, height=1,
                                          zeros=zeros[1:].contiguous(), positions=(1, npos), relin=relin)
    for g, w in zip(part, run[1:]):
        assert torch.equal(g, w)
    flat = fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, order=64, degree=0, delta=0.5, width=npos, height=1, zeros=None, relin=relin)
    ref0 = oracle_mod.oracle_approximated_step(rorc, h(amp)[0], h(idx)[0], h(cnt)[0], 64, 0, 0.5, npos, 1, None)
    assert flat[0].shape[-3] == 2 and np.array_equal(h(flat[0])[0], ref0[0])
    # the driver loop of homo/server_decode.cpp:120-137 over two runs
    runs = torch.stack([torch.cat([amp, cnt]), torch.cat([ctx.random_ct(1, size=2, seed=930), ctx.random_ct(1, size=2, seed=931)])]).contiguous()
    acc0 = ctx.random_ct(npos, size=2, seed=940)
    z2 = torch.stack([zeros, ctx.random_ct(npos * degree * 2, size=2, seed=950).reshape(npos, degree, 2, 2, ctx.k, ctx.n)]).contiguous()
    index = idx.clone()
    got = fhe.circuits.decode_channel(ev, pc, runs, index, acc0, z2, 64, degree, 0.5, npos, 1, relin=relin)
    assert got.shape[-3] == 2
    want, index2 = acc0.clone(), idx.clone()
    for p in range(2):
        step = fhe.circuits.approximated_step(ev, pc, runs[p, 0:1].contiguous(), index2, runs[p, 1:2].contiguous(), 64, degree, 0.5, npos, 1, z2[p], relin=relin)
        want = ev.add(want, torch.cat(step))
        index2 = ev.add(index2, runs[p, 1:2].contiguous())
    assert torch.equal(got, want) and torch.equal(index, index2)


def _ring_mul(a, b, t):
    """exact product in Z_t[x] / (x^n + 1) (coefficients < 2^14 and n <= 8192: every partial sum stays below 2^41)"""
    n = len(a)
    c = np.convolve(a.astype(np.int64), b.astype(np.int64))
    r = c[:n].copy()
    r[:n - 1] -= c[n:]
    return r % t


@pytest.mark.parametrize("dbc", [30, 60])
def test_homomorphic_sin_known_answers_with_noise_budgets(fhe, oracle_mod, dbc):
    """the reference's own check (tests/test_decode.cpp:39-48: x in {1..7}, n = 8192, print the remaining noise budget and the
    decoded value beside sin x) in both modes.  The known answer is the PLAINTEXT polynomial: a correct BFV evaluation decrypts
    to exactly the product / sum of the encoded polynomials in Z_t[x]/(x^n + 1) whenever its noise budget is positive (for
    x = 1, 2, 3 the coefficients of the tenth power wrap modulo t = 2^14 and the decoded number is not sin x -- in the reference
    too; that is the encoder's limit, not the evaluation's).  Budgets of the two modes are printed side by side."""
    ctx, orc, rorc, relin, sk, pk = _setup(fhe, oracle_mod, "P8192", dbc, key_seed=5)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    t = ctx.t
    xs = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]
    x = fhe.to_device(np.stack([orc.encrypt(pk, orc.encode(v), seed=40 + i) for i, v in enumerate(xs)]))
    z = fhe.to_device(np.stack([orc.encrypt(pk, orc.encode(0.0), seed=60 + i) for i in range(len(xs))]))
    ref = fhe.to_host(fhe.circuits.homomorphic_sin(ev, pc, x, z))
    rel = fhe.to_host(fhe.circuits.homomorphic_sin(ev, pc, x, z, relin=relin))
    assert ref.shape[1] == 11 and rel.shape[1] == 2
    E = lambda v: orc.encode(v).astype(np.int64)
    print()
    ok_rel = 0
    for i, v in enumerate(xs):
        s = (E(v) + E(-3 * math.pi / 2)) % t                      # homo/fhe_decode.h:57
        s2 = _ring_mul(s, s, t)
        s4 = _ring_mul(s2, s2, t)
        s8 = _ring_mul(s4, s4, t)
        s6 = _ring_mul(_ring_mul(s4, s, t), s, t)
        s10 = _ring_mul(_ring_mul(s8, s, t), s, t)
        want = E(-1.0)
        for pw, cf in ((s2, 0.5), (s4, -1.0 / 24.0), (s6, 1.0 / 720.0), (s8, -1.0 / 40320.0), (s10, 1.0 / 3628800.0)):      # :66-118
            want = (want + _ring_mul(pw, E(cf), t)) % t
        p_ref, b_ref = orc.decrypt(sk, ref[i])
        p_rel, b_rel = orc.decrypt(sk, rel[i])
        print("[relin P8192 dbc=%d] homomorphic_sin(%g): plaintext model decodes to %.9f (sin = %.9f); noise budget left: reference mode %d bits, relinearised %d bits"
              % (dbc, v, orc.decode(want.astype(np.uint64)), math.sin(v), b_ref, b_rel))
        if b_ref > 0:
            assert np.array_equal(p_ref.astype(np.int64), want), ("reference mode", v)
        if b_rel > 0:
            assert np.array_equal(p_rel.astype(np.int64), want), ("relinearised mode", v)
            ok_rel += 1
    assert ok_rel == len(xs), "the relinearised evaluation must keep a positive noise budget at n = 8192"


def test_relin_handle_argument_errors(fhe, oracle_mod):
    """fhe_circuits_create_relin: keys and bit count come together, dbc in 1..60; the handle reports its mode and its output sizes"""
    import ctypes as C
    ctx, orc, rorc, relin, _, _ = _setup(fhe, oracle_mod, "SMALL", 16)
    L = fhe._lib.load()
    h = C.c_void_p()
    evk = C.c_void_p(relin[0].data_ptr())
    assert L.fhe_circuits_create_relin(ctx.h, 100, 100, None, 30, C.byref(h)) < 0 and b"come together" in L.fhe_last_error()
    assert L.fhe_circuits_create_relin(ctx.h, 100, 100, evk, 0, C.byref(h)) < 0
    assert L.fhe_circuits_create_relin(ctx.h, 100, 100, evk, 61, C.byref(h)) < 0 and b"out of range" in L.fhe_last_error()
    assert L.fhe_circuits_create_relin(ctx.h, 100, 100, evk, 16, C.byref(h)) == 0
    K = fhe.circuits
    assert L.fhe_circuits_relin_dbc(h) == 16
    assert [L.fhe_circuits_out_size(h, c, a) for c, a in ((K.CUBIC, 2), (K.LINEAR, 2), (K.SAMPLE_BICUBIC, 0), (K.SAMPLE_LINEAR, 0), (K.SINCOS, 0), (K.STEP, 12), (K.STEP, 0),
                                                        (K.DECODE, 3))] == [2] * 8
    assert L.fhe_circuits_out_size(h, 99, 0) == 0
    L.fhe_circuits_destroy(h)
    plain = K.circuits_of(K.PlainCache(ctx))
    assert L.fhe_circuits_relin_dbc(plain.h) == 0
    assert [plain.out_size(c, a) for c, a in ((K.CUBIC, 2), (K.CUBIC, 4), (K.LINEAR, 3), (K.SAMPLE_BICUBIC, 0), (K.SAMPLE_LINEAR, 0), (K.SINCOS, 0), (K.STEP, 12), (K.STEP, 0))] \
        == [4, 6, 4, 6, 4, 11, 22, 3]


def test_streaming_servers_in_the_relinearised_mode(fhe, oracle_mod, tmp_path):
    """server.server_resize / server.server_decode with relin=(evk, dbc): the streaming loops of homo/server_resize.cpp and
    homo/server_decode.cpp with every product relinearised -- records of size 2, each equal to the oracle's relinearised
    composition on the same stream (sampled pixels; every decode position)"""
    import sys
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from refrun import clamp, parse_stream, read_records, sample_origins, write_record
    ctx, orc, rorc, relin, _, _ = _setup(fhe, oracle_mod, "SEAL23_4096", 30)
    W = H = 10
    w = h = 6
    pix = orc.random_ct(W * H * 3, seed=21).reshape(W * H, 3, 2, orc.k, orc.n)
    fin, fout = tmp_path / "in.ct", tmp_path / "out.ct"
    with open(fin, "wb") as f:
        for p in range(W * H):
            for c in range(3):
                write_record(f, pix[p, c])
    bank = orc.random_ct(w * h * 2, seed=22)
    pos = [0]

    def encrypt(values):
        out = bank[pos[0]:pos[0] + len(values)]
        pos[0] += len(values)
        return fhe.to_device(np.ascontiguousarray(out))
    for bicubic in (True, False):
        pos[0] = 0
        assert fhe.server.server_resize(ctx, str(fin), str(fout), W, H, w, h, bicubic, encrypt, rows_per_step=3, relin=relin) == w * h
        out = read_records(str(fout), 2, orc.k, orc.n, w * h * 3)
        origins = sample_origins(W, H, w, h)
        for o in (0, 7, w * h - 1):
            xi, yi = origins[o]
            P = lambda dx, dy, ch: pix[clamp(yi + dy, 0, H - 1) * W + clamp(xi + dx, 0, W - 1), ch]
            for ch in (0, 2):
                if bicubic:
                    want = oracle_mod.oracle_sample_bicubic_calls(rorc, [P(dx, dy, ch) for dy in (-1, 0, 1, 2) for dx in (-1, 0, 1, 2)], bank[2 * o], bank[2 * o + 1])
                else:
                    want = oracle_mod.oracle_sample_linear_calls(rorc, [P(0, 0, ch), P(1, 0, ch), P(0, 1, ch), P(1, 1, ch)], bank[2 * o], bank[2 * o + 1])
                assert np.array_equal(out[o * 3 + ch], want), (bicubic, o, ch)
    # decode: two channels with runs, one without
    from refrun import oracle_server_decode
    pairs, width, height, order, degree, delta = (1, 0, 1), 2, 1, 64, 1, 0.5
    runs = orc.random_ct(2 * sum(pairs), seed=31).reshape(sum(pairs), 2, 2, orc.k, orc.n)
    hook = orc.random_ct(sum(1 + width * height + p * width * height * degree * 2 for p in pairs), seed=32)
    din, dout = tmp_path / "dec_in.ct", tmp_path / "dec_out.ct"
    with open(din, "wb") as f:
        for r in range(runs.shape[0]):
            write_record(f, runs[r, 0])
            write_record(f, runs[r, 1])
    at = [0]

    def zeros(count):
        out = hook[at[0]:at[0] + count]
        at[0] += count
        return fhe.to_device(np.ascontiguousarray(out))
    zeros.seek = lambda i: at.__setitem__(0, i)
    fhe.server.server_decode(ctx, str(din), str(dout), width, height, pairs, zeros, order=order, degree=degree, delta=delta, relin=relin)
    got = parse_stream(open(dout, "rb").read(), orc.k, orc.n)
    want = oracle_server_decode(rorc, oracle_mod, runs, pairs, width, height, hook, order, degree, delta)
    assert len(got) == 3 * width * height
    for i in range(width * height):
        for ch in range(3):
            assert got[i * 3 + ch].shape[0] == 2 and np.array_equal(got[i * 3 + ch], want[ch][i]), (i, ch)


# ------------------------------------------------------------------------------------------------------------------------------------
# the second placement: the reference's Cubic / Linear unchanged, ONE evaluator.relinearize at the end of each (FHE_RELIN_PER_CUBIC)
# ------------------------------------------------------------------------------------------------------------------------------------
def _setup_tail(fhe, om, name, dbc, key_seed=77):
    """context + oracle + the keys for s^2 AND s^3 in both libraries' NTT slot orders"""
    p = SMALL if name == "SMALL" else om.PRESETS[name]
    ctx, orc = fhe.SEALContext(p["n"], p["q"], p["t"]), om.Oracle(p["n"], p["q"], p["t"])
    sk, pk = orc.keygen(key_seed)
    evks = orc.evk_gen_powers(sk, dbc=dbc, count=2)                 # [2][k][nd][2][k][n], oracle NTT form
    coeff = np.zeros_like(evks)
    for idx in np.ndindex(evks.shape[:4]):
        for i in range(ctx.k):
            coeff[idx + (i,)] = orc.ntt_inv(evks[idx + (i,)], i)
    evk_dev = fhe.Evaluator(ctx).ntt_forward(fhe.to_device(coeff)).contiguous()
    return ctx, orc, om.TailRelinOracle(orc, evks, dbc), (evk_dev, dbc, "cubic"), evks, sk, pk


@pytest.mark.parametrize("preset,dbc", [("SMALL", 16), ("P8192", 30), ("SEAL23_4096", 30), ("P8192", 60)])
def test_relinearize_any_size_and_per_cubic_placement_vs_oracle(fhe, oracle_mod, preset, dbc):
    """fhe_relinearize_n (sizes 3, 4, 5 -> 2: key switches for s^4, s^3, s^2, the top polynomial first, as SEAL's relinearize) and the
    per-Cubic placement: fhe_cubic / fhe_linear == the op-by-op Evaluator calls + one relinearize == the oracle's composition
    `oracle_cubic_calls -> relinearize_n` (TailRelinOracle), bit for bit, levels 1 and 2, an operand at q - 1."""
    import torch
    ctx, orc, torc, relin, evks, sk, pk = _setup_tail(fhe, oracle_mod, preset, dbc)
    ev, pc, h = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx), fhe.to_host
    for size in (3, 4):
        x = ctx.random_ct(3, size=size, seed=900 + size)
        got = h(ev.relinearize(x, relin[0] if size > 3 else relin[0][0].contiguous(), dbc))
        for i in range(3):
            assert np.array_equal(got[i], orc.relinearize_n(h(x)[i], evks, dbc)), (size, i)
    # the one-pass form (all key switches as one sum, the default where the lazy sums have room) == the sequential steps (FHE_RELIN_STEPS=1)
    x4 = ctx.random_ct(5, size=4, seed=950)
    steps = fhe.Evaluator(fhe.SEALContext(ctx.n, ctx.q, ctx.t, switches={"FHE_RELIN_STEPS": 1}))
    assert torch.equal(ev.relinearize(x4, relin[0], dbc), steps.relinearize(x4, relin[0], dbc))
    with pytest.raises(ValueError):
        ev.relinearize(ctx.random_ct(1, size=5, seed=1), relin[0], dbc)           # needs the keys for s^4 as well
    A, B, C, D = (ctx.random_ct(3, size=2, seed=400 + i) for i in range(4))
    t = ctx.random_ct(3, size=2, seed=410)
    A[1] = torch.tensor([q - 1 for q in ctx.q], dtype=torch.int64, device=A.device).view(1, ctx.k, 1).expand(2, ctx.k, ctx.n)
    r1 = fhe.circuits.cubic(ev, pc, A, B, C, D, t, relin=relin)
    assert r1.shape[-3] == 2
    assert torch.equal(r1, fhe.circuits.cubic_evaluator_calls(ev, pc, A, B, C, D, t, relin))
    # the reference's result (4 polynomials) relinearised by one call is the same thing
    assert torch.equal(r1, ev.relinearize(fhe.circuits.cubic(ev, pc, A, B, C, D, t), relin[0], dbc))
    want1 = [oracle_mod.oracle_cubic_calls(torc, h(A)[i], h(B)[i], h(C)[i], h(D)[i], h(t)[i]) for i in range(3)]
    for i in range(3):
        assert want1[i].shape[0] == 2 and np.array_equal(h(r1)[i], want1[i]), i
    r2 = fhe.circuits.cubic(ev, pc, r1, B, r1, D, t, relin=relin)
    assert np.array_equal(h(r2)[2], oracle_mod.oracle_cubic_calls(torc, want1[2], h(B)[2], want1[2], h(D)[2], h(t)[2]))
    l1 = fhe.circuits.linear(ev, pc, A, B, t, relin=relin)
    l2 = fhe.circuits.linear(ev, pc, l1, r1, t, relin=relin)
    assert l1.shape[-3] == 2 and l2.shape[-3] == 2
    for i in (0, 1):
        o1 = oracle_mod.oracle_linear_calls(torc, h(A)[i], h(B)[i], h(t)[i])
        assert np.array_equal(h(l1)[i], o1)
        assert np.array_equal(h(l2)[i], oracle_mod.oracle_linear_calls(torc, o1, want1[i], h(t)[i]))
    # the decode circuits have no Cubic to end: refused for such a handle
    with pytest.raises(fhe._lib.FheError):
        fhe.circuits.homomorphic_sin(ev, pc, ctx.random_ct(1, size=2, seed=3), ctx.random_ct(1, size=2, seed=4), relin=relin)


@pytest.mark.parametrize("preset,dbc", [("SMALL", 16), ("P8192", 30)])
def test_samplers_per_cubic_placement_vs_oracle_and_decrypt(fhe, oracle_mod, preset, dbc):
    """SampleBicubic / SampleLinear / the shared-offset ResizeImage (+ a row shard) with ONE relinearize per Cubic / Linear on an
    8x8 -> 4x4 image of real encryptions: library == oracle composition bit for bit, every output decrypts to the closed form, and
    the remaining noise budget is printed beside the every-product placement's and the reference mode's"""
    import torch
    ctx, orc, torc, relin, evks, sk, pk = _setup_tail(fhe, oracle_mod, preset, dbc, key_seed=21)
    each = (relin[0][0].contiguous(), dbc)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    W = H = 8
    w = h_ = 4
    vals = [float((29 * x + 53 * y) % 256) for y in range(H) for x in range(W)]
    pix = np.stack([orc.encrypt(pk, orc.encode(v), seed=500 + i) for i, v in enumerate(vals)])
    d_pix = fhe.to_device(pix)

    def plain_cubic(A, B, C, D, t):
        a, b, c = -A + 3 * B - 3 * C + D, 2 * A - 5 * B + 4 * C - D, C - A
        return 0.5 * (a * t * t + b * t * t + c * t) + B

    taps, fx, fy = fhe.circuits.resize_sample_plan(W, H, w, h_, bicubic=True)
    xf = np.stack([orc.encrypt(pk, orc.encode(f), seed=600 + i) for i, f in enumerate(fx)])
    yf = np.stack([orc.encrypt(pk, orc.encode(f), seed=700 + i) for i, f in enumerate(fy)])
    dx, dy = fhe.to_device(xf), fhe.to_device(yf)
    out = fhe.to_host(fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, dx, dy, relin=relin))
    assert out.shape == (w * h_, 2, ctx.k, ctx.n)
    for o in ((0, 5, 15) if preset == "SMALL" else (5,)):
        assert np.array_equal(out[o], oracle_mod.oracle_sample_bicubic_calls(torc, [pix[i] for i in taps[o]], xf[o], yf[o])), o
    if preset != "SMALL":
        other = fhe.to_host(fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, dx, dy, relin=each))
        ref_mode = fhe.to_host(fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, dx, dy))
        b_tail, b_each, b_ref = [], [], []
        for o in range(w * h_):
            v = [vals[i] for i in taps[o]]
            cols = [plain_cubic(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3], fx[o]) for r in range(4)]
            expect = plain_cubic(cols[0], cols[1], cols[2], cols[3], fy[o])
            plain, budget = orc.decrypt(sk, out[o])
            assert budget > 0 and abs(orc.decode(plain) - expect) < 1e-6
            b_tail.append(budget)
            b_each.append(orc.decrypt(sk, other[o])[1])
            b_ref.append(orc.decrypt(sk, ref_mode[o])[1])
        print("\n[relin %s dbc=%d] SampleBicubic noise budget left (min over 16 pixels): per-Cubic placement %d bits, every-product placement %d bits, "
              "reference mode %d bits" % (preset, dbc, min(b_tail), min(b_each), min(b_ref)))
    xs, ys = fhe.to_device(xf[:w].copy()), fhe.to_device(yf[::w].copy())
    shared = fhe.circuits.resize_bicubic_shared(ev, pc, d_pix, W, H, w, h_, xs, ys, batch=8, band_rows=2, relin=relin)
    per_px = fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, xs.repeat(h_, 1, 1, 1).contiguous(), ys.repeat_interleave(w, dim=0).contiguous(), relin=relin)
    assert shared.shape[-3] == 2 and torch.equal(shared, per_px)
    first, cnt = fhe.circuits.resize_source_rows(H, h_, 1, 3)
    part = fhe.circuits.resize_bicubic_shared(ev, pc, d_pix[first * W:(first + cnt) * W].contiguous(), W, H, w, h_, xs, ys[1:3].contiguous(), batch=8, band_rows=2,
                                              rows=(1, 3), src_rows=(first, cnt), relin=relin)
    assert torch.equal(part, shared[w:3 * w])
    tl, lx, ly = fhe.circuits.resize_sample_plan(W, H, w, h_, bicubic=False)
    xl = np.stack([orc.encrypt(pk, orc.encode(f), seed=800 + i) for i, f in enumerate(lx)])
    yl = np.stack([orc.encrypt(pk, orc.encode(f), seed=850 + i) for i, f in enumerate(ly)])
    lin = fhe.to_host(fhe.circuits.sample_linear(ev, pc, d_pix, tl, fhe.to_device(xl), fhe.to_device(yl), relin=relin))
    assert lin.shape[1] == 2
    for o in (0, 9):
        assert np.array_equal(lin[o], oracle_mod.oracle_sample_linear_calls(torc, [pix[i] for i in tl[o]], xl[o], yl[o])), o


def test_streaming_server_resize_with_the_per_cubic_placement(fhe, oracle_mod, tmp_path):
    """server.server_resize(relin=(keys, dbc, "cubic")), per-pixel offsets and shared offsets: records of size 2, each equal to the oracle's
    composition `reference sequence -> relinearize` on the same stream (TailRelinOracle), bicubic and bilinear"""
    import sys
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from refrun import clamp, read_records, sample_origins, write_record
    ctx, orc, torc, relin, _, _, _ = _setup_tail(fhe, oracle_mod, "SEAL23_4096", 30)
    W = H = 10
    w = h = 6
    pix = orc.random_ct(W * H * 3, seed=21).reshape(W * H, 3, 2, orc.k, orc.n)
    fin, fout = tmp_path / "in.ct", tmp_path / "out.ct"
    with open(fin, "wb") as f:
        for p in range(W * H):
            for c in range(3):
                write_record(f, pix[p, c])
    bank = orc.random_ct(w * h * 2, seed=22)
    pos = [0]

    def encrypt(values):
        out = bank[pos[0]:pos[0] + len(values)]
        pos[0] += len(values)
        return fhe.to_device(np.ascontiguousarray(out))
    origins = sample_origins(W, H, w, h)
    for bicubic in (True, False):
        pos[0] = 0
        assert fhe.server.server_resize(ctx, str(fin), str(fout), W, H, w, h, bicubic, encrypt, rows_per_step=3, relin=relin) == w * h
        out = read_records(str(fout), 2, orc.k, orc.n, w * h * 3)
        for o in (0, 7, w * h - 1):
            xi, yi = origins[o]
            P = lambda dx, dy, ch: pix[clamp(yi + dy, 0, H - 1) * W + clamp(xi + dx, 0, W - 1), ch]
            for ch in (0, 2):
                if bicubic:
                    want = oracle_mod.oracle_sample_bicubic_calls(torc, [P(dx, dy, ch) for dy in (-1, 0, 1, 2) for dx in (-1, 0, 1, 2)], bank[2 * o], bank[2 * o + 1])
                else:
                    want = oracle_mod.oracle_sample_linear_calls(torc, [P(0, 0, ch), P(1, 0, ch), P(0, 1, ch), P(1, 1, ch)], bank[2 * o], bank[2 * o + 1])
                assert np.array_equal(out[o * 3 + ch], want), (bicubic, o, ch)
    # shared offsets: the bank is read as one ciphertext per output column, then one per output row
    pos[0] = 0
    assert fhe.server.server_resize(ctx, str(fin), str(fout), W, H, w, h, True, encrypt, rows_per_step=3, relin=relin, shared_offsets=True) == w * h
    out = read_records(str(fout), 2, orc.k, orc.n, w * h * 3)
    for o in (0, 7, w * h - 1):
        xi, yi = origins[o]
        P = lambda dx, dy, ch: pix[clamp(yi + dy, 0, H - 1) * W + clamp(xi + dx, 0, W - 1), ch]
        want = oracle_mod.oracle_sample_bicubic_calls(torc, [P(dx, dy, 1) for dy in (-1, 0, 1, 2) for dx in (-1, 0, 1, 2)], bank[o % w], bank[w + o // w])
        assert np.array_equal(out[o * 3 + 1], want), o


# ------------------------------------------------------------------------------------------------------------------------------------
# the third placement: the samplers unchanged, ONE evaluator.relinearize of every output pixel (FHE_RELIN_PER_SAMPLE)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("preset,dbc", [("SMALL", 16), ("P8192", 30), ("P8192", 60), ("SEAL23_4096", 30)])
def test_per_sample_placement_vs_oracle_and_decrypt(fhe, oracle_mod, preset, dbc):
    """SampleBicubic / SampleLinear / the shared-offset resize (+ a row shard) / a stand-alone Cubic and Linear with ONE relinearize of every
    result (6 / 4 -> 2: keys for s^2 .. s^5; one pass of key switches at dbc 60, two at dbc 30, sequential steps on the general kernels):
    library == oracle composition `reference sequence -> relinearize_n` (SampleRelinOracle) bit for bit; equal to the reference mode's result
    relinearised by one call; decrypts to the closed form with the budget printed beside the other modes'."""
    import torch
    p = SMALL if preset == "SMALL" else oracle_mod.PRESETS[preset]
    ctx, orc = fhe.SEALContext(p["n"], p["q"], p["t"]), oracle_mod.Oracle(p["n"], p["q"], p["t"])
    sk, pk = orc.keygen(21)
    evks = orc.evk_gen_powers(sk, dbc=dbc, count=4)
    coeff = np.zeros_like(evks)
    for idx in np.ndindex(evks.shape[:4]):
        for i in range(ctx.k):
            coeff[idx + (i,)] = orc.ntt_inv(evks[idx + (i,)], i)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    relin = (ev.ntt_forward(fhe.to_device(coeff)).contiguous(), dbc, "sample")
    sorc = oracle_mod.SampleRelinOracle(orc, evks, dbc)
    W = H = 8
    w = h_ = 4
    vals = [float((29 * x + 53 * y) % 256) for y in range(H) for x in range(W)]
    pix = np.stack([orc.encrypt(pk, orc.encode(v), seed=500 + i) for i, v in enumerate(vals)])
    d_pix = fhe.to_device(pix)
    taps, fx, fy = fhe.circuits.resize_sample_plan(W, H, w, h_, bicubic=True)
    xf = np.stack([orc.encrypt(pk, orc.encode(f), seed=600 + i) for i, f in enumerate(fx)])
    yf = np.stack([orc.encrypt(pk, orc.encode(f), seed=700 + i) for i, f in enumerate(fy)])
    dx, dy = fhe.to_device(xf), fhe.to_device(yf)
    got = fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, dx, dy, relin=relin)
    assert got.shape[-3] == 2
    # the reference mode's size-6 pixels relinearised by ONE call are the same thing
    ref6 = fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, dx, dy)
    assert ref6.shape[-3] == 6 and torch.equal(got, ev.relinearize(ref6, relin[0], dbc))
    out = fhe.to_host(got)
    for o in ((0, 5, 15) if preset == "SMALL" else (5,)):
        assert np.array_equal(out[o], oracle_mod.oracle_sample_bicubic_calls(sorc, [pix[i] for i in taps[o]], xf[o], yf[o])), o
    if preset == "P8192":
        def plain_cubic(A, B, C, D, t):
            a, b, c = -A + 3 * B - 3 * C + D, 2 * A - 5 * B + 4 * C - D, C - A
            return 0.5 * (a * t * t + b * t * t + c * t) + B
        budgets, ref_b = [], []
        for o in range(w * h_):
            v = [vals[i] for i in taps[o]]
            cols = [plain_cubic(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3], fx[o]) for r in range(4)]
            plain, budget = orc.decrypt(sk, out[o])
            assert budget > 0 and abs(orc.decode(plain) - plain_cubic(cols[0], cols[1], cols[2], cols[3], fy[o])) < 1e-6
            budgets.append(budget)
            ref_b.append(orc.decrypt(sk, fhe.to_host(ref6[o:o + 1])[0])[1])
        print("\n[relin %s dbc=%d] SampleBicubic noise budget left (min over 16 pixels): per-sample placement %d bits, reference mode %d bits" % (preset, dbc, min(budgets), min(ref_b)))
    xs, ys = fhe.to_device(xf[:w].copy()), fhe.to_device(yf[::w].copy())
    shared = fhe.circuits.resize_bicubic_shared(ev, pc, d_pix, W, H, w, h_, xs, ys, batch=8, band_rows=2, relin=relin)
    per_px = fhe.circuits.sample_bicubic(ev, pc, d_pix, taps, xs.repeat(h_, 1, 1, 1).contiguous(), ys.repeat_interleave(w, dim=0).contiguous(), relin=relin)
    assert shared.shape[-3] == 2 and torch.equal(shared, per_px)
    first, cnt = fhe.circuits.resize_source_rows(H, h_, 1, 3)
    part = fhe.circuits.resize_bicubic_shared(ev, pc, d_pix[first * W:(first + cnt) * W].contiguous(), W, H, w, h_, xs, ys[1:3].contiguous(), batch=8, band_rows=2,
                                              rows=(1, 3), src_rows=(first, cnt), relin=relin)
    assert torch.equal(part, shared[w:3 * w])
    tl, lx, ly = fhe.circuits.resize_sample_plan(W, H, w, h_, bicubic=False)
    xl = np.stack([orc.encrypt(pk, orc.encode(f), seed=800 + i) for i, f in enumerate(lx)])
    yl = np.stack([orc.encrypt(pk, orc.encode(f), seed=850 + i) for i, f in enumerate(ly)])
    lin = fhe.to_host(fhe.circuits.sample_linear(ev, pc, d_pix, tl, fhe.to_device(xl), fhe.to_device(yl), relin=relin))
    assert lin.shape[1] == 2
    for o in (0, 9):
        assert np.array_equal(lin[o], oracle_mod.oracle_sample_linear_calls(sorc, [pix[i] for i in tl[o]], xl[o], yl[o])), o
    # stand-alone Cubic (operands of 2 and of 4 polynomials) and Linear: the reference's result, relinearised
    A, B, C, D = (ctx.random_ct(3, size=2, seed=400 + i) for i in range(4))
    t = ctx.random_ct(3, size=2, seed=410)
    c2 = fhe.circuits.cubic(ev, pc, A, B, C, D, t, relin=relin)
    r4 = fhe.circuits.cubic(ev, pc, A, B, C, D, t)
    assert c2.shape[-3] == 2 and torch.equal(c2, ev.relinearize(r4, relin[0], dbc)) and torch.equal(c2, fhe.circuits.cubic_evaluator_calls(ev, pc, A, B, C, D, t, relin))
    c6 = fhe.circuits.cubic(ev, pc, r4, r4, r4, r4, t, relin=relin)
    assert c6.shape[-3] == 2 and torch.equal(c6, ev.relinearize(fhe.circuits.cubic(ev, pc, r4, r4, r4, r4, t), relin[0], dbc))
    h = fhe.to_host
    assert np.array_equal(h(c2)[1], sorc.sample_tail(oracle_mod.oracle_cubic_calls(orc, h(A)[1], h(B)[1], h(C)[1], h(D)[1], h(t)[1])))
    l2 = fhe.circuits.linear(ev, pc, A, B, t, relin=relin)
    assert l2.shape[-3] == 2 and np.array_equal(h(l2)[2], sorc.sample_tail(oracle_mod.oracle_linear_calls(orc, h(A)[2], h(B)[2], h(t)[2])))
    with pytest.raises(fhe._lib.FheError):
        fhe.circuits.homomorphic_sin(ev, pc, ctx.random_ct(1, size=2, seed=3), ctx.random_ct(1, size=2, seed=4), relin=relin)


def test_relinearize_n_argument_errors_and_empty_batches(fhe, oracle_mod):
    """fhe_relinearize_n / fhe_relinearize_poly / fhe_circuits_create_relin_at: sizes below 3 and above FHE_MAX_POLYS, null pointers, strides
    smaller than the ciphertext, a decomposition bit count out of range, an unknown placement and partially overlapping ranges return
    FHE_ERR_PARAM with a message; an empty batch is a no-op; nothing is launched on a refusal (the output keeps its bytes)."""
    import ctypes as C
    import torch
    ctx = fhe.SEALContext.preset("SEAL23_4096")
    L, kn = fhe._lib.load(), ctx.k * ctx.n
    kg = fhe.KeyGenerator(ctx, seed=3)
    evk = kg.generate_evaluation_keys(30, 2).contiguous()
    ct = ctx.random_ct(2, size=4, seed=1)
    out = torch.full((2, 2, ctx.k, ctx.n), 7, dtype=torch.int64, device=ct.device)
    nbytes = L.fhe_relinearize_n_scratch_bytes(ctx.h, 4, 30, 2)
    assert nbytes > L.fhe_relinearize_scratch_bytes(ctx.h, 30, 2) > 0 and L.fhe_relinearize_n_scratch_bytes(ctx.h, 2, 30, 2) == 0
    scr = torch.empty(nbytes // 8 + 1, dtype=torch.int64, device=ct.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    args = lambda size=4, stride=4 * kn, o=out, ostride=2 * kn, count=2, dbc=30, e=evk: (ctx.h, p(ct), size, stride, p(o) if o is not None else None, ostride, count, p(e) if e is not None else None, dbc, p(scr), nbytes, None)
    assert L.fhe_relinearize_n(*args(size=2)) == -1 and b"polynomials" in L.fhe_last_error()
    assert L.fhe_relinearize_n(*args(size=65)) == -1
    assert L.fhe_relinearize_n(*args(dbc=61)) == -1 and L.fhe_relinearize_n(*args(dbc=0)) == -1
    assert L.fhe_relinearize_n(*args(o=None)) == -1 and L.fhe_relinearize_n(*args(e=None)) == -1
    assert L.fhe_relinearize_n(*args(stride=3 * kn)) == -1 and b"stride" in L.fhe_last_error()
    assert L.fhe_relinearize_n(*args(ostride=kn)) == -1
    inside = C.c_void_p(ct.data_ptr() + 8 * kn)                       # output inside the input range with another stride
    assert L.fhe_relinearize_n(ctx.h, p(ct), 4, 4 * kn, inside, 2 * kn, 2, p(evk), 30, p(scr), nbytes, None) == -1 and b"overlaps" in L.fhe_last_error()
    assert L.fhe_relinearize_poly(ctx.h, p(ct), 4 * kn, 1, p(out), 2 * kn, 2, p(evk), 30, p(scr), nbytes, None) == -1           # source polynomial below 2
    assert L.fhe_relinearize_poly(ctx.h, p(ct), 4 * kn, 4, p(out), 2 * kn, 2, p(evk), 30, p(scr), nbytes, None) == -1           # ... beyond the ciphertext
    torch.cuda.synchronize()
    assert bool((out == 7).all())                                        # no refusal launched anything
    assert L.fhe_relinearize_n(*args(count=0)) == 0 and bool((out == 7).all())
    h = C.c_void_p()
    assert L.fhe_circuits_create_relin_at(ctx.h, 100, 100, p(evk), 30, 3, C.byref(h)) == -1 and b"placement" in L.fhe_last_error()
    assert L.fhe_circuits_create_relin_at(ctx.h, 100, 100, None, 30, 1, C.byref(h)) == -1
    # and the good call: the same bits as the Evaluator wrapper
    assert L.fhe_relinearize_n(*args()) == 0
    assert torch.equal(out, fhe.Evaluator(ctx).relinearize(ctx.random_ct(2, size=4, seed=1), evk, 30))


def test_host_checks_evaluation_keys_before_the_library_reads_behind_them(fhe, tmp_path):
    """The C ABI takes evaluation keys as a bare pointer (include/fhe_hip.h): Evaluator.relinearize, Circuits(relin=...) and server.read_evaluation_keys
    refuse keys made for another decomposition bit count (fewer digits than the call will read), another context, too few powers, or a stream that is
    short / unreduced / foreign; write_evaluation_keys -> read_evaluation_keys returns the tensor and the bit count it was given."""
    import io
    import struct
    import torch
    ctx = fhe.SEALContext.preset("SEAL23_4096")
    other = fhe.SEALContext.preset("SEAL23_2048")
    ev, kg = fhe.Evaluator(ctx), fhe.KeyGenerator(ctx, seed=3)
    keys60 = kg.generate_evaluation_keys(60, 2).contiguous()           # one digit per prime
    keys30 = kg.generate_evaluation_keys(30, 2).contiguous()           # two
    ct = ctx.random_ct(2, size=4, seed=1)
    with pytest.raises(ValueError, match="another decomposition bit count"):
        ev.relinearize(ct, keys60, 30)                                 # the call at dbc 30 reads twice what keys60 holds
    with pytest.raises(ValueError, match="shape"):
        ev.relinearize(ct, keys30, 60)                                 # enough words, but not this context's layout at dbc 60
    with pytest.raises(ValueError, match="needs the keys"):
        ev.relinearize(ctx.random_ct(1, size=5, seed=2), keys30, 30)
    with pytest.raises(ValueError, match="1 .. 60"):
        ev.relinearize(ct, keys30, 61)
    with pytest.raises(ValueError, match="contiguous int64"):
        ev.relinearize(ct, keys30.cpu(), 30)
    with pytest.raises(ValueError, match="another decomposition bit count"):
        fhe.circuits.Circuits(ctx, relin=(keys60, 30, "cubic"))
    with pytest.raises(ValueError, match="Circuits"):
        fhe.circuits.Circuits(ctx, relin=(keys30[:1].contiguous().view(-1), 30, "cubic"))      # one power where the placement takes two
    good = ev.relinearize(ct, keys30, 30)
    # the key file of the C++ hosts, read back
    f = io.BytesIO()
    fhe.server.write_evaluation_keys(f, keys30, 30)
    raw = f.getvalue()
    back, dbc = fhe.server.read_evaluation_keys(ctx, io.BytesIO(raw))
    assert dbc == 30 and torch.equal(back, keys30) and back.device == keys30.device
    assert torch.equal(ev.relinearize(ct, back, dbc), good)
    hdr = lambda **kw: raw[:8] + struct.pack("<6I", *[kw.get(name, v) for name, v in zip(("dbc", "digits", "count", "k", "n", "r"), struct.unpack("<6I", raw[8:32]))])
    for bad, exc, what in ((raw[:-8], EOFError, "truncated"), (b"FHEHIP1\0" + raw[8:], ValueError, "does not hold"), (raw[:20], ValueError, "does not hold"),
                           (hdr(dbc=60) + raw[32:], ValueError, "digit"), (hdr(digits=1) + raw[32:], ValueError, "digit"), (hdr(count=0) + raw[32:], ValueError, "out of range"),
                           (hdr(count=63) + raw[32:], ValueError, "out of range"), (hdr(count=3) + raw[32:], EOFError, "truncated"), (hdr(n=8192) + raw[32:], ValueError, "context has"),
                           (raw[:32] + b"\xff" * 8 + raw[40:], ValueError, "not reduced")):
        with pytest.raises(exc, match=what):
            fhe.server.read_evaluation_keys(ctx, io.BytesIO(bad))
    with pytest.raises(ValueError, match="context has"):
        fhe.server.read_evaluation_keys(other, io.BytesIO(raw))
