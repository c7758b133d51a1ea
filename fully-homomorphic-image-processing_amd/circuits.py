"""The reference's homomorphic circuits, re-expressed over the batched Evaluator.

Each function follows the Evaluator call sequence of the cited reference lines, but on whole
batches: the leading dimension of every ciphertext tensor ([B, size, k, n]) runs over independent
pixels / output positions, which the reference visits in serial loops
(homo/server_jpeg.cpp:113, homo/fhe_resize.h:350,381, homo/server_decode.cpp:120-137).

Server-side fresh encryptions inside the reference's circuits (the fractional offsets in
SampleLinear/SampleBicubic, homo/fhe_resize.h:230-266, and the Enc(0) accumulators of
homomorphic_sin/cos, homo/fhe_decode.h:54,134) are randomised; here they are explicit inputs so that
results are reproducible.
"""
import math

import torch

from .evaluator import FractionalEncoder, PreparedPlain


class PlainCache:
    """encode + lift + NTT each distinct constant once (the reference redoes it on every call)."""

    def __init__(self, ctx, encoder=None):
        self.ctx = ctx
        self.enc = encoder or FractionalEncoder(ctx)
        self._prepared = {}
        self._plain = {}

    def plain(self, v):
        v = float(v)
        if v not in self._plain:
            self._plain[v] = self.enc.encode(v)
        return self._plain[v]

    def prepared(self, v):
        v = float(v)
        if v not in self._prepared:
            self._prepared[v] = PreparedPlain(self.ctx, self.plain(v))
        return self._prepared[v]


# ------------------------------------------------------------------------------------------------
# JPEG path, op-at-a-time (the fused kernel is Evaluator.dct8x8_quant)
# ------------------------------------------------------------------------------------------------
def rgb_to_ycc(ev, pc, r, g, b):
    """homo/fhe_image.h:310-325 as separate Evaluator calls (fused form: Evaluator.rgb_to_ycc)."""
    M, P = ev.multiply_plain, pc.prepared
    y = ev.sub_plain(ev.add(ev.add(M(r, P(0.299)), M(g, P(0.587))), M(b, P(0.114))), pc.plain(128.0))
    u = ev.add(ev.sub(M(r, P(-0.168736)), M(g, P(0.331264))), M(b, P(0.5)))
    v = ev.sub(ev.sub(M(r, P(0.5)), M(g, P(0.418688))), M(b, P(0.081312)))
    return y, u, v


# ------------------------------------------------------------------------------------------------
# resize path
# ------------------------------------------------------------------------------------------------
def _base2_cubic_constants(pc):
    """True if the encoder writes Cubic's constants the way the fused passes assume (base 2):
    encode(3) = x+1, encode(2) = x, encode(5) = x^2+1, encode(4) = x^2, encode(0.5) = -x^(n-1)."""
    if getattr(pc, "_cubic_ok", None) is None:
        import numpy as np
        t, n = pc.ctx.t, pc.ctx.n

        def nz(v):
            p = np.asarray(pc.plain(v), dtype=np.uint64)
            return {int(i): int(p[i]) for i in np.nonzero(p)[0]}

        pc._cubic_ok = (nz(3) == {0: 1, 1: 1} and nz(2) == {1: 1} and nz(5) == {0: 1, 2: 1} and nz(4) == {2: 1}
                        and nz(0.5) == {n - 1: t - 1})
    return pc._cubic_ok


def cubic_powers(ev, t, relin=None):
    """t2 = square(t) and t3 = multiply(t, t) of Cubic (homo/fhe_resize.h:174-175).  Both are the same
    ring tensor (the library's square IS multiply(t, t)), so one product serves both, and a caller
    that evaluates several Cubics at the same t (SampleBicubic: four rows share xfract) passes the
    pair in instead of recomputing it -- the ciphertext bits are the same either way."""
    t2 = ev.square(t)
    if relin is not None and t2.shape[-3] == 3:
        t2 = ev.relinearize(t2, relin[0], relin[1])
    return t2, t2


def cubic(ev, pc, A, B, C, D, t, relin=None, powers=None, prepared=None):
    """Cubic(result, A,B,C,D,t): homo/fhe_resize.h:143-189.  Note t3 = t*t exactly as the reference
    computes it (:175).  powers = cubic_powers(ev, t) may be shared between calls with the same t, and
    prepared = (prepare_operand(t3), prepare_operand(t2), prepare_operand(t)) for the same batch shape
    saves their base extension and transforms in every call.

    relin=(evk_ntt, dbc) switches on the relinearised mode (NOT what the
    reference does): every product is brought back to size 2, so the result has size 2 instead of
    s+2 and memory stays flat; ciphertext bits then differ from the reference path by construction
    (key-switching noise), the decrypted value does not."""
    M, P = ev.multiply_plain, pc.prepared

    def mul(x, y):
        z = ev.multiply(x, y)
        return ev.relinearize(z, relin[0], relin[1]) if relin is not None and z.shape[-3] == 3 else z

    fused = _base2_cubic_constants(pc) and A.shape == B.shape == C.shape == D.shape
    if fused:       # the linear parts in one pass each (same ring elements as the calls below)
        a, b, c = ev.cubic_coeffs(A.contiguous(), B.contiguous(), C.contiguous(), D.contiguous())
    else:
        a = ev.add(ev.sub(ev.sub(M(B, P(3)), A), M(C, P(3))), D)
        b = ev.sub(ev.add(ev.sub(M(A, P(2)), M(B, P(5))), M(C, P(4))), D)
        c = ev.sub(C, A)
    t2, t3 = powers if powers is not None else cubic_powers(ev, t, relin)
    if prepared is not None:
        a, b, c = mul(a, prepared[0]), mul(b, prepared[1]), mul(c, prepared[2])
    else:
        a, b, c = mul(a, t3), mul(b, t2), mul(c, t)
    if fused and a.shape == b.shape and c.shape[-3] <= a.shape[-3] and B.shape[-3] <= a.shape[-3]:
        if c.shape[-3] < a.shape[-3]:                # c * t is one polynomial shorter than a * t3: pad with zeros
            c = torch.cat([c, torch.zeros_like(a[..., c.shape[-3]:, :, :])], dim=-3)
        return ev.cubic_combine(a.contiguous(), b.contiguous(), c.contiguous(), B.contiguous())
    a = ev.add(ev.add(a, b), c)
    a = M(a, P(0.5))
    return ev.add(a, B)


def linear(ev, pc, A, B, t):
    """Linear(result, A,B,t): homo/fhe_resize.h:191-204: (1 - t) A + t B."""
    omt = ev.add_plain(ev.negate(t), pc.plain(1.0))
    return ev.add(ev.multiply(omt, A), ev.multiply(B, t))


def _clamp(v, lo, hi):
    return lo if v < lo else hi if v > hi else v


def resize_sample_plan(src_w, src_h, dst_w, dst_h, bicubic=True):
    """The index arithmetic of ResizeImage / SampleBicubic / SampleLinear / GetPixelClamped
    (homo/fhe_resize.h:350-351, 381-382, 260-290, 215-220), in float32 like the reference.
    Returns per output pixel: the clamped source pixel indices (16 for bicubic, row-major 4x4;
    4 for bilinear) and the fractional offsets (xfract, yfract)."""
    import numpy as np
    f32 = np.float32
    taps, fx, fy = [], [], []
    for y in range(dst_h):
        v = f32(f32(y) / f32(dst_h - 1) * f32(src_h)) - f32(0.5)
        for x in range(dst_w):
            u = f32(f32(x) / f32(dst_w - 1) * f32(src_w)) - f32(0.5)
            xi, yi = int(u), int(v)
            fx.append(float(u - f32(math.floor(u))))
            fy.append(float(v - f32(math.floor(v))))
            offs = [(-1, -1), (0, -1), (1, -1), (2, -1), (-1, 0), (0, 0), (1, 0), (2, 0),
                    (-1, 1), (0, 1), (1, 1), (2, 1), (-1, 2), (0, 2), (1, 2), (2, 2)] if bicubic else \
                   [(0, 0), (1, 0), (0, 1), (1, 1)]
            taps.append([_clamp(yi + dy, 0, src_h - 1) * src_w + _clamp(xi + dx, 0, src_w - 1) for dx, dy in offs])
    return taps, fx, fy


def sample_bicubic(ev, pc, pixels, taps, xfract, yfract):
    """SampleBicubic for a batch of output pixels and one colour channel (homo/fhe_resize.h:254-305).
    pixels: [src_pixels, 2, k, n]; taps: [B][16] source indices; xfract/yfract: [B, 2, k, n]
    ciphertexts of the fractional offsets.  Returns [B, 6, k, n]."""
    idx = torch.as_tensor(taps, dtype=torch.long, device=pixels.device)        # [B, 16]
    p = [pixels[idx[:, i]].contiguous() for i in range(16)]
    px = cubic_powers(ev, xfract)                       # shared by the four row Cubics, prepared once
    p2 = ev.prepare_operand(px[0])
    prep = (p2, p2, ev.prepare_operand(xfract.contiguous()))
    cols = [cubic(ev, pc, p[4 * r + 0], p[4 * r + 1], p[4 * r + 2], p[4 * r + 3], xfract, powers=px, prepared=prep) for r in range(4)]
    return cubic(ev, pc, cols[0], cols[1], cols[2], cols[3], yfract)


def resize_bicubic_shared(ev, pc, pixels, src_w, src_h, dst_w, dst_h, xfract, yfract, batch=256, band_rows=4, consume=None):
    """ResizeImage with SampleBicubic (homo/fhe_resize.h:254-392) for one colour channel when the fractional
    offsets arrive as ONE ciphertext per output column (xfract [dst_w, 2, k, n]) and ONE per output row
    (yfract [dst_h, 2, k, n]) -- SURVEY.md section 8(d), config 3: "xfract/yfract ciphertexts are inputs generated
    per distinct fractional value".  frac(u) depends on x only and frac(v) on y only (:351,382), so with shared
    ciphertexts the reference's per-pixel work repeats itself and this function forms every repeated ring element
    once; each output equals sample_bicubic(..., xfract[x], yfract[y]) bit for bit:

      * a row Cubic (:296-299) is a function of (output column x, source row r) only; consecutive output rows'
        4-row windows overlap, so the 4 * dst_h row Cubics of a column collapse to one per source row touched
        (128 instead of 256 for 128 -> 64);
      * xfract^2 and the prepared (extended + transformed) forms of xfract, xfract^2 are formed once per column,
        yfract^2 and its prepared forms once per row; the products index them in place (entry c % dst_w resp.
        first_row + c // dst_w of the prepared batch: fhe_multiply_prepared_shared), nothing is copied.

    The reference's server encrypts fresh offsets for every sample (:262,266); its results are therefore
    randomised per pixel and only sample_bicubic with per-pixel ciphertexts reproduces that run bit for bit
    (server.server_resize does).  Source rows are visited as a sliding window (`band_rows` output rows at a
    time; row Cubics no output row needs any more are dropped), like the reference's loader (:352-379).

    Returns [dst_h * dst_w, 6, k, n] (row-major), or None when `consume(first_pixel, tensor)` takes the bands."""
    taps, _, _ = resize_sample_plan(src_w, src_h, dst_w, dst_h, bicubic=True)
    colx = [[t % src_w for t in taps[x][0:4]] for x in range(dst_w)]                       # clamped xi-1 .. xi+2
    rows_of = [[taps[y * dst_w][4 * j] // src_w for j in range(4)] for y in range(dst_h)]  # clamped yi-1 .. yi+2
    xfract, yfract = xfract.contiguous(), yfract.contiguous()
    x2, y2 = ev.square(xfract), ev.square(yfract)                                          # t2 (= t3, :174-175) per column / row
    px2, px1 = ev.prepare_operand(x2), ev.prepare_operand(xfract)
    py2, py1 = ev.prepare_operand(y2), ev.prepare_operand(yfract)
    sx2, sx1 = px2.shared(1), px1.shared(1)
    dev = pixels.device
    cache = {}                                                                             # source row -> [dst_w, 4, k, n]

    def row_cubics(new_rows):
        if not new_rows:
            return
        res = []
        rows_per_call = max(1, batch // dst_w)
        for s0 in range(0, len(new_rows), rows_per_call):
            part = [(r, x) for r in new_rows[s0:s0 + rows_per_call] for x in range(dst_w)]
            tap = [torch.as_tensor([r * src_w + colx[x][i] for r, x in part], dtype=torch.long, device=dev) for i in range(4)]
            A, B, C, D = (pixels.index_select(0, t) for t in tap)
            # pairs are ordered (row, column) with the column fastest and whole rows per call: pair c multiplies the
            # column's xfract / xfract^2, entry c % dst_w of the prepared batches -- no copy (fhe_multiply_prepared_shared)
            res.append(cubic(ev, pc, A, B, C, D, None, powers=(None, None), prepared=(sx2, sx2, sx1)))
        allr = torch.cat(res, dim=0) if len(res) > 1 else res[0]
        for i, r in enumerate(new_rows):
            cache[r] = allr[i * dst_w:(i + 1) * dst_w]

    outs = []
    for y0 in range(0, dst_h, band_rows):
        ys = list(range(y0, min(y0 + band_rows, dst_h)))
        need = sorted({r for y in ys for r in rows_of[y]})
        row_cubics([r for r in need if r not in cache])
        for r in [r for r in cache if r < need[0]]:                                        # the window only moves down
            del cache[r]
        rows_per_call = max(1, batch // dst_w)
        for s0 in range(0, len(ys), rows_per_call):
            yy = ys[s0:s0 + rows_per_call]
            A, B, C, D = (torch.cat([cache[rows_of[y][j]] for y in yy], dim=0) for j in range(4))
            # pixel c of the call sits in output row yy[c // dst_w]: entry yy[0] + c // dst_w of the prepared yfract batches
            sy2, sy1 = py2.shared(dst_w, yy[0]), py1.shared(dst_w, yy[0])
            o = cubic(ev, pc, A, B, C, D, None, powers=(None, None), prepared=(sy2, sy2, sy1))
            if consume is not None:
                consume(yy[0] * dst_w, o)
            else:
                outs.append(o)
    return None if consume is not None else torch.cat(outs, dim=0)


def sample_linear(ev, pc, pixels, taps, xfract, yfract):
    """SampleLinear for one channel (homo/fhe_resize.h:222-252).  Returns [B, 4, k, n]."""
    idx = torch.as_tensor(taps, dtype=torch.long, device=pixels.device)        # [B, 4]
    p00, p10, p01, p11 = (pixels[idx[:, i]].contiguous() for i in range(4))
    col0 = linear(ev, pc, p00, p10, xfract)
    col1 = linear(ev, pc, p01, p11, xfract)
    return linear(ev, pc, col0, col1, yfract)


# ------------------------------------------------------------------------------------------------
# decode path
# ------------------------------------------------------------------------------------------------
def _taylor_terms(ev, pc, x, coeffs):
    """the five power terms of homomorphic_sin / homomorphic_cos (homo/fhe_decode.h:56-113 / :136-193):
    even Taylor polynomial of degree 10 in (x - 3 pi / 2); only the coefficients differ."""
    M, P = ev.multiply_plain, pc.prepared
    sx = ev.add_plain(x, pc.plain(-3 * math.pi / 2.0))
    # The reference rebuilds every power from a fresh copy of sx (11 squares, 4 multiplies); the
    # repeated squares are the same ring elements bit for bit, so each is formed once here.
    s2 = ev.square(sx)
    s4 = ev.square(s2)
    s8 = ev.square(s4)
    p2 = M(s2, P(coeffs[0]))
    p4 = M(s4, P(coeffs[1]))
    psx = ev.prepare_operand(sx.contiguous())            # the four products below share this operand
    p6 = M(ev.multiply(ev.multiply(s4, psx), psx), P(coeffs[2]))
    p8 = M(s8, P(coeffs[3]))
    p10 = M(ev.multiply(ev.multiply(s8, psx), psx), P(coeffs[4]))
    return (p2, p4, p6, p8, p10)


def _taylor_sum(ev, pc, zero, constant, terms):
    """res = Enc(0) + constant, then the terms in the reference's order (homo/fhe_decode.h:114-119)."""
    res = ev.add_plain(zero, pc.plain(constant))
    for term in terms:
        res = ev.add(res, term)
    return res


SIN_COEFFS = (0.5, -1.0 / 24.0, 1.0 / 720.0, -1.0 / 40320.0, 1.0 / 3628800.0)
COS_COEFFS = (-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0)


def homomorphic_sin(ev, pc, x, zero):
    """homo/fhe_decode.h:48-120; `zero` plays the role of encrypt(encode(0.0)) (:54)."""
    return _taylor_sum(ev, pc, zero, -1.0, _taylor_terms(ev, pc, x, SIN_COEFFS))


def homomorphic_cos(ev, pc, x, zero):
    """homo/fhe_decode.h:128-200 (the reference shifts by -3pi/2 here too, :137, and falls off the end
    without a return statement, :200; the value it leaves in `res` is what is returned here)."""
    return _taylor_sum(ev, pc, zero, 1.0, _taylor_terms(ev, pc, x, COS_COEFFS))


def approximated_step(ev, pc, amplitude, index, count, order, degree, delta, width, height, zeros):
    """The homomorphic overload of approximated_step (homo/fhe_decode.h:202-242) for ONE run.
    amplitude/index/count: [1, 2, k, n].  zeros: callable (i, j, which) -> [1, 2, k, n] encryption of
    zero for position i, harmonic j, which in {"sin", "cos"}.  Returns a list of width*height
    ciphertexts [1, 22, k, n].

    Faithful to the reference's quirk: `offset` is advanced by add_plain(offset, encode(i)) INSIDE
    the harmonic loop (:229), after cos_arg was copied from it.

    The reference walks positions x harmonics serially with one ciphertext per call; here only the
    (cheap) offset chain is serial.  All width*height*degree cosine polynomials are evaluated as ONE
    batch, the sine polynomial once per harmonic (its argument b * f_j does not depend on the
    position) -- the same ring operations on the same operands, so the same bits, but launches that
    fill the GPU."""
    import numpy as np
    M, P = ev.multiply_plain, pc.prepared
    npos = width * height
    b = M(count, P(0.5))
    offset = ev.negate(ev.add_plain(ev.add(index, b), pc.plain(-0.5)))
    b = ev.add_plain(b, pc.plain(delta - 0.5))
    factors = [float(np.float32(j)) * math.pi / float(order) for j in range(1, degree + 1)]
    pre = []                                    # pre[i][j-1] = offset as copied into cos_arg (:226)
    for i in range(npos):
        row = []
        for _ in range(degree):
            row.append(offset)
            offset = ev.add_plain(offset, pc.plain(float(i)))
        pre.append(row)
    # batch index = (j-1) * npos + i
    cos_arg = torch.cat([M(torch.cat([pre[i][j] for i in range(npos)]), P(factors[j])) for j in range(degree)])
    zc = torch.cat([zeros(i, j + 1, "cos") for j in range(degree) for i in range(npos)])
    zs = torch.cat([zeros(i, j + 1, "sin") for j in range(degree) for i in range(npos)])
    co = _taylor_sum(ev, pc, zc, 1.0, _taylor_terms(ev, pc, cos_arg, COS_COEFFS))
    sin_arg = torch.cat([M(b, P(factors[j])) for j in range(degree)])                  # [degree, 2, k, n]
    sin_terms = [t.repeat_interleave(npos, dim=0) for t in _taylor_terms(ev, pc, sin_arg, SIN_COEFFS)]
    s = _taylor_sum(ev, pc, zs, -1.0, sin_terms)
    del sin_terms
    prod = ev.multiply(s, co)                                                           # [degree * npos, 21, k, n]
    c = M(b, P(1.0 / float(order))).repeat(npos, 1, 1, 1)
    for j in range(degree):
        term = M(prod[j * npos:(j + 1) * npos], P(2.0 / (math.pi * float(np.float32(j + 1)))))
        c = ev.add(c, term)
    out = ev.multiply(c, amplitude.repeat(npos, 1, 1, 1))
    return [out[i:i + 1] for i in range(npos)]
