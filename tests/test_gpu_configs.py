"""GPU: BASELINE.json configs[2] and configs[3] at their STATED parameters -- n = 8192, the four 54/55-bit
SEAL 2.3 moduli (preset P8192: general u64 kernels + 61-bit auxiliary base, a different kernel path from
the 36-bit sets the small circuit tests use) -- checked, not just timed.

configs[2]  bicubic resize 128x128 -> 64x64 (homo/fhe_resize.h:254-392), one colour channel: the full sample
            plan in batches; sampled output pixels bit-equal to the oracle; the whole 4096-pixel output
            independent of how it is batched (order-independent digest over all 18 GiB of results).
configs[3]  approximated_step (homo/fhe_decode.h:202-242), W*H = 16, degree 12, size-22 results: one
            position bit-equal to the oracle (24 Taylor polynomials + 12 products of sizes 11x11 on the
            host, ~1 min), all 16 positions bit-equal to the REFERENCE's own code run op by op through the
            facade (oracle/_ref/ref_decode_circuit), when that binary was built."""
import numpy as np
import pytest

from refrun import oracle_sample, ref_bin, run_decode_circuit, sample_origins

pytestmark = pytest.mark.gpu

# configs[2]: the sixteen output pixels (x, y) of the 64 x 64 result that are compared with the ORACLE (about 0.6 s of host time each):
# the four corners (both taps clamped), one pixel on each edge away from the corners (one tap clamped: rows only / columns only),
# and eight interior pixels -- among them both sides of a source-window slide (output rows 31 | 32 read source rows 61..64 | 63..66,
# output columns 31 | 32 likewise) and the rows next to the clamped borders
PICKS_XY = [(0, 0), (63, 0), (0, 63), (63, 63),
            (20, 0), (41, 63), (0, 25), (63, 40),
            (17, 31), (17, 32), (31, 45), (32, 45), (40, 10), (5, 50), (62, 1), (1, 62)]


def _oracle_pixel(fhe, orc, pixels, W, H, origin, xf_ct, yf_ct, cache):
    """one output pixel of ResizeImage / SampleBicubic (homo/fhe_resize.h:254-305) through the oracle's Cubic, from the device image"""
    xi, yi = origin
    need = sorted({min(max(yi + dy, 0), H - 1) * W + min(max(xi + dx, 0), W - 1) for dx in (-1, 0, 1, 2) for dy in (-1, 0, 1, 2)})
    for i in need:
        if i not in cache:
            cache[i] = fhe.to_host(pixels[i:i + 1])[0]
    pix = {i: cache[i][None] for i in need}                                    # oracle_sample indexes pix[idx, ch]

    class View:
        def __getitem__(self, key):
            return pix[key[0]][key[1]]
    return oracle_sample(orc, View(), W, H, xi, yi, 0, xf_ct, yf_ct, True)


def test_config2_bicubic_128_to_64_at_n8192(fhe, oracle_mod):
    import torch
    ctx = fhe.SEALContext.preset("P8192")
    orc = oracle_mod.Oracle.preset("P8192")
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    W = H = 128
    w = h = 64
    taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
    pixels = ctx.random_ct(W * H, size=2, seed=fhe.SEED)                     # 8 GiB, resident
    n_out = w * h
    xf_all, yf_all = ctx.random_ct(n_out, size=2, seed=11), ctx.random_ct(n_out, size=2, seed=12)
    origins = sample_origins(W, H, w, h)
    picks = {y * w + x: None for x, y in PICKS_XY}                            # corners, edges, interior incl. both sides of a window slide

    def run(batch):
        total = 0
        for s in range(0, n_out, batch):
            e = min(s + batch, n_out)
            out = fhe.circuits.sample_bicubic(ev, pc, pixels, taps[s:e], xf_all[s:e].contiguous(), yf_all[s:e].contiguous())
            assert out.shape == (e - s, 6, ctx.k, ctx.n)
            total = (total + ctx.digest(out, index0=s * 6 * ctx.k * ctx.n)) % (1 << 64)
            for o in picks:
                if s <= o < e and picks[o] is None:
                    picks[o] = fhe.to_host(out[o - s:o - s + 1])[0]
            del out
        torch.cuda.synchronize()
        return total

    d256 = run(256)
    d192 = run(192)                                                            # other chunk boundaries, other batch shapes
    assert d256 == d192
    hp = {}
    assert len(picks) == 16 and origins[31 * w + 17][1] + 2 == origins[32 * w + 17][1]      # rows 31 | 32 sit on either side of a slide of the 4-row window
    for o, got in picks.items():
        ref = _oracle_pixel(fhe, orc, pixels, W, H, origins[o], fhe.to_host(xf_all[o:o + 1])[0], fhe.to_host(yf_all[o:o + 1])[0], hp)
        assert np.array_equal(got, ref), (o % w, o // w)


def test_config2_shared_offsets_full_size_equals_per_pixel_sampling(fhe, oracle_mod):
    """configs[2] at its stated size and parameters with one offset ciphertext per output column / row:
    the shared-row evaluation (12,288 Cubics) and the per-pixel evaluation (20,480 Cubics) give the same 4096 size-6
    ciphertexts -- compared through the position-dependent digest of all 6 GiB, and sixteen pixels of the shared-offset
    evaluation (corners, edges, both sides of a window slide) against the ORACLE directly, not only against sample_bicubic."""
    import torch
    ctx = fhe.SEALContext.preset("P8192")
    orc = oracle_mod.Oracle.preset("P8192")
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    W = H = 128
    w = h = 64
    pixels = ctx.random_ct(W * H, size=2, seed=fhe.SEED)
    xf, yf = ctx.random_ct(w, size=2, seed=11), ctx.random_ct(h, size=2, seed=12)
    words = 6 * ctx.k * ctx.n
    keep, acc = {}, [0]

    def consume(first, t):
        acc[0] = (acc[0] + ctx.digest(t, index0=first * words)) % (1 << 64)
        for o in [y * w + x for x, y in PICKS_XY]:
            if first <= o < first + t.shape[0]:
                keep[o] = t[o - first].clone()
    assert fhe.circuits.resize_bicubic_shared(ev, pc, pixels, W, H, w, h, xf, yf, consume=consume) is None
    taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
    xs = torch.as_tensor([x for y in range(h) for x in range(w)], device=pixels.device)
    ys = torch.as_tensor([y for y in range(h) for x in range(w)], device=pixels.device)
    total = 0
    for s in range(0, w * h, 256):
        e = s + 256
        out = fhe.circuits.sample_bicubic(ev, pc, pixels, taps[s:e], xf[xs[s:e]].contiguous(), yf[ys[s:e]].contiguous())
        total = (total + ctx.digest(out, index0=s * words)) % (1 << 64)
        for o, t in keep.items():
            if s <= o < e:
                assert torch.equal(out[o - s], t), o
    assert len(keep) == 16 and total == acc[0]
    origins, hp = sample_origins(W, H, w, h), {}
    hx, hy = fhe.to_host(xf), fhe.to_host(yf)
    for o, t in keep.items():
        ref = _oracle_pixel(fhe, orc, pixels, W, H, origins[o], hx[o % w], hy[o // w], hp)
        assert np.array_equal(fhe.to_host(t[None])[0], ref), (o % w, o // w)


@pytest.mark.parametrize("preset,W,H,w,h,band,batch", [("SMALL", 16, 12, 8, 7, 3, 16), ("SMALL", 9, 9, 17, 17, 4, 64), ("P8192", 24, 24, 12, 12, 4, 48)])
def test_shared_row_cubics_equal_per_pixel_sampling(fhe, preset, W, H, w, h, band, batch):
    """circuits.resize_bicubic_shared (one offset ciphertext per output column / row: every row Cubic, square and
    prepared operand formed once) == circuits.sample_bicubic pixel by pixel with those ciphertexts, bit for bit:
    down- and up-scaling, ragged last band, windows that skip and that repeat source rows, clamped borders."""
    import torch
    ctx = fhe.SEALContext(1024, [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], 1 << 14, 0) if preset == "SMALL" else fhe.SEALContext.preset(preset)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    pixels = ctx.random_ct(W * H, size=2, seed=5)
    xf, yf = ctx.random_ct(w, size=2, seed=11), ctx.random_ct(h, size=2, seed=12)
    got = fhe.circuits.resize_bicubic_shared(ev, pc, pixels, W, H, w, h, xf, yf, batch=batch, band_rows=band)
    assert got.shape == (w * h, 6, ctx.k, ctx.n)
    taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
    xs = torch.as_tensor([x for y in range(h) for x in range(w)], device=pixels.device)
    ys = torch.as_tensor([y for y in range(h) for x in range(w)], device=pixels.device)
    step = 64
    for s in range(0, w * h, step):
        e = min(s + step, w * h)
        want = fhe.circuits.sample_bicubic(ev, pc, pixels, taps[s:e], xf[xs[s:e]].contiguous(), yf[ys[s:e]].contiguous())
        assert torch.equal(got[s:e], want), s
    # streamed form: the bands handed to a consumer are the same tensor in pieces
    seen = []
    assert fhe.circuits.resize_bicubic_shared(ev, pc, pixels, W, H, w, h, xf, yf, batch=batch, band_rows=band,
                                              consume=lambda first, t: seen.append((first, t.clone()))) is None      # the band buffer is reused
    assert [f for f, _ in seen] == sorted(f for f, _ in seen) and sum(t.shape[0] for _, t in seen) == w * h
    assert all(torch.equal(t, got[f:f + t.shape[0]]) for f, t in seen)


def test_config3_approximated_step_16_positions_degree_12_at_n8192(fhe, oracle_mod, tmp_path):
    ctx = fhe.SEALContext.preset("P8192")
    orc = oracle_mod.Oracle.preset("P8192")
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    npos, deg, order, delta = 16, 12, 64, 0.5                                  # homo/server_decode.cpp:37-39
    run_in = orc.random_ct(3, seed=900)
    zs = orc.random_ct(npos * deg * 2, seed=1000)                              # Enc(0) accumulators in the reference's call order
    pick = lambda i, j, which: zs[(i * deg + j - 1) * 2 + (which == "cos")]
    mine = fhe.circuits.approximated_step(ev, pc, *(fhe.to_device(run_in[i:i + 1]) for i in range(3)), order=order, degree=deg, delta=delta,
                                          width=npos, height=1, zeros=lambda i, j, which: fhe.to_device(pick(i, j, which)[None]))
    assert len(mine) == npos and all(m.shape == (1, 22, ctx.k, ctx.n) for m in mine)
    mine = [fhe.to_host(m)[0] for m in mine]

    # one position against the oracle: the offset chain (add_plain only) is walked for every position, the
    # polynomial evaluation only for `probe`
    probe = 5
    import math
    b = orc.multiply_plain(run_in[2], orc.encode(0.5))
    offset = orc.negate(orc.add_plain(orc.add(run_in[1], b), orc.encode(-0.5)))
    b = orc.add_plain(b, orc.encode(delta - 0.5))
    c = None
    for i in range(probe + 1):
        if i == probe:
            c = orc.multiply_plain(b, orc.encode(1.0 / float(order)))
        for j in range(1, deg + 1):
            f = float(np.float32(j)) * math.pi / float(order)
            cos_arg = offset.copy()
            offset = orc.add_plain(offset, orc.encode(float(i)))               # homo/fhe_decode.h:229 (inside the j loop)
            if i == probe:
                s = oracle_mod.oracle_homomorphic_sin(orc, orc.multiply_plain(b, orc.encode(f)), pick(i, j, "sin"))
                co = oracle_mod.oracle_homomorphic_cos(orc, orc.multiply_plain(cos_arg, orc.encode(f)), pick(i, j, "cos"))
                term = orc.multiply_plain(orc.multiply(s, co), orc.encode(2.0 / (math.pi * float(np.float32(j)))))
                c = orc.add(c, term)
    assert np.array_equal(mine[probe], orc.multiply(c, run_in[0]))

    if ref_bin("ref_decode_circuit", True):
        got = run_decode_circuit(str(tmp_path), orc, "step", run_in, zs, gpu=True, n_arg=8192, extra=(order, deg, delta, npos, 1),
                                 env_extra={"FHE_SEAL23_MODULI": "1"}, sizes=(22,) * npos)
        for i in range(npos):
            assert np.array_equal(got[i], mine[i]), i
