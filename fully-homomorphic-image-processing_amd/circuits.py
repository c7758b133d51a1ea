"""The reference's homomorphic circuits on whole batches: ctypes callers of include/fhe_circuits.h.

The circuits themselves -- tap gathers, the t^2 reuse of Cubic, prepared operands, the growth of ciphertext
sizes, every temporary -- live inside libfhe_hip.so (csrc/circuits.hip); this module only allocates the
output and scratch tensors and passes pointers, so a C++ host gets exactly the same batched path
(seal/hip_circuits.h).  The leading dimension of every ciphertext tensor ([B, size, k, n]) runs over
independent pixels / output positions, which the reference visits in serial loops
(homo/server_jpeg.cpp:113, homo/fhe_resize.h:350,381, homo/server_decode.cpp:120-137).

Server-side fresh encryptions inside the reference's circuits (the fractional offsets in
SampleLinear/SampleBicubic, homo/fhe_resize.h:230-266, and the Enc(0) accumulators of
homomorphic_sin/cos, homo/fhe_decode.h:54,134) are randomised; here they are explicit inputs so that
results are reproducible.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .evaluator import FractionalEncoder, PreparedPlain, _ptr, _stream, check_evaluation_keys


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
# the circuits handle and scratch
# ------------------------------------------------------------------------------------------------
CUBIC, LINEAR, SAMPLE_BICUBIC, SAMPLE_LINEAR, SINCOS, STEP, DECODE = range(7)      # FHE_CIRC_* of include/fhe_circuits.h


class Circuits:
    """fhe_circuits: the constants of the resize / decode circuits for one context and encoder.
    relin=(evk_ntt, dbc): the RELINEARISED mode (fhe_circuits_create_relin; SURVEY.md section 8(f) #4, not what the
    reference does): evaluator.relinearize after every multiply / square, so every ciphertext of every circuit has two
    polynomials.  evk_ntt as KeyGenerator.generate_evaluation_keys(dbc) returns it; the handle keeps it alive.
    relin=(evk_ntt, dbc, "cubic"): the second placement (FHE_RELIN_PER_CUBIC, include/fhe_circuits.h): the reference's Cubic /
    Linear sequences unchanged and ONE relinearize at the end of each (size 4 / 3 -> 2; two key switches per Cubic where the
    first placement spends five); evk_ntt = generate_evaluation_keys(dbc, 2): the keys for s^2 and s^3.  Resize circuits only.
    relin=(evk_ntt, dbc, "sample"): the third (FHE_RELIN_PER_SAMPLE): the samplers exactly as the reference evaluates them and ONE relinearize of
    every output pixel (6 -> 2 / 4 -> 2); evk_ntt = generate_evaluation_keys(dbc, 4): the keys for s^2 .. s^5.  Resize circuits only."""

    def __init__(self, ctx, int_coeffs=100, frac_coeffs=100, relin=None):
        self.ctx = ctx
        h = C.c_void_p()
        if relin is None:
            _lib.call("fhe_circuits_create", ctx.h, int_coeffs, frac_coeffs, C.byref(h))
        else:
            self._evk, dbc = relin[0], relin[1]
            placement = relin_placement(relin)
            check_evaluation_keys(ctx, self._evk, dbc, (1, 2, 4)[placement], "Circuits")
            if placement == 1:
                assert self._evk.dim() == 6 and self._evk.shape[0] >= 2, "per-Cubic placement: keys for s^2 and s^3 (generate_evaluation_keys(dbc, 2))"
            if placement == 2:
                assert self._evk.dim() == 6 and self._evk.shape[0] >= 4, "per-sample placement: keys for s^2 .. s^5 (generate_evaluation_keys(dbc, 4))"
            _lib.call("fhe_circuits_create_relin_at", ctx.h, int_coeffs, frac_coeffs, _ptr(self._evk), int(dbc), placement, C.byref(h))
        self.h = h
        self._scratch = None

    def out_size(self, circuit, arg=0):
        return int(_lib.load().fhe_circuits_out_size(self.h, circuit, arg))

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                _lib.load().fhe_circuits_destroy(h)
            except Exception:
                pass
            self.h = None

    def scratch(self, nbytes):
        if not nbytes:
            raise _lib.FheError(-1, _lib.load().fhe_last_error().decode("utf-8", "replace") or "scratch query failed")
        if self._scratch is None or self._scratch.numel() < nbytes:
            self._scratch = None
            self._scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.ctx.device)
        return self._scratch


def relin_placement(relin):
    """0: after every product (relin=(evk, dbc)), 1: once per Cubic / Linear (relin=(evk, dbc, "cubic"))"""
    if relin is None or len(relin) < 3 or relin[2] in (0, None, "product", "every"):
        return 0
    if relin[2] in (1, "cubic"):
        return 1
    if relin[2] in (2, "sample"):
        return 2
    raise ValueError("relin placement must be 'product', 'cubic' or 'sample', got %r" % (relin[2],))


def circuits_of(pc, relin=None):
    """the fhe_circuits handle that goes with a PlainCache (same context and encoder), created on first use;
    relin=(evk_ntt, dbc): the relinearising handle for those keys"""
    if relin is not None:
        cache = pc.__dict__.setdefault("_circuits_relin", {})
        key = (relin[0].data_ptr(), int(relin[1]), relin_placement(relin))
        if key not in cache:
            cache[key] = Circuits(pc.ctx, pc.enc.int_coeffs, pc.enc.frac_coeffs, relin=relin)
        return cache[key]
    h = getattr(pc, "_circuits", None)
    if h is None:
        h = pc._circuits = Circuits(pc.ctx, pc.enc.int_coeffs, pc.enc.frac_coeffs)
    return h


def _count(t, size):
    assert t.dtype == torch.int64 and t.is_contiguous() and t.shape[-3] == size, (t.shape, size)
    c = 1
    for d in t.shape[:-3]:
        c *= d
    return c


def _taps_array(taps, width):
    a = np.ascontiguousarray(taps, dtype=np.uint32)
    assert a.ndim == 2 and a.shape[1] == width, a.shape
    return a


# ------------------------------------------------------------------------------------------------
# resize path
# ------------------------------------------------------------------------------------------------
def _base2_cubic_constants(pc):
    """True if the encoder writes Cubic's constants the way the fused passes assume (base 2):
    encode(3) = x+1, encode(2) = x, encode(5) = x^2+1, encode(4) = x^2, encode(0.5) = -x^(n-1)."""
    t, n = pc.ctx.t, pc.ctx.n

    def nz(v):
        p = np.asarray(pc.plain(v), dtype=np.uint64)
        return {int(i): int(p[i]) for i in np.nonzero(p)[0]}

    return (nz(3) == {0: 1, 1: 1} and nz(2) == {1: 1} and nz(5) == {0: 1, 2: 1} and nz(4) == {2: 1}
            and nz(0.5) == {n - 1: t - 1})


def cubic_evaluator_calls(ev, pc, A, B, C, D, t, relin=None):
    """Cubic(result, A,B,C,D,t) as the reference's Evaluator call sequence, one (batched) call per line of
    homo/fhe_resize.h:149-188 -- the op-by-op form the fused circuit (cubic below) is tested against, and the
    carrier of the relinearised mode: relin=(evk_ntt, dbc) brings every product back to size 2 (SURVEY.md
    section 8(f) #4, NOT what the reference does), so the result has size 2 instead of s+2; ciphertext bits
    then differ from the reference path by construction (key-switching noise), the decrypted value does not."""
    M, P = ev.multiply_plain, pc.prepared
    each = relin is not None and relin_placement(relin) == 0     # relinearize after every product
    tail = relin is not None and relin_placement(relin) in (1, 2)    # the reference's sequence, ONE relinearize of the result (relin=(evk, dbc, "cubic"); a
    #                                                                  stand-alone Cubic of the "sample" placement is relinearised by its caller the same way)

    def mul(x, y):
        z = ev.multiply(x, y)
        return ev.relinearize(z, relin[0], relin[1]) if each and z.shape[-3] == 3 else z

    a = ev.add(ev.sub(ev.sub(M(B, P(3)), A), M(C, P(3))), D)
    b = ev.sub(ev.add(ev.sub(M(A, P(2)), M(B, P(5))), M(C, P(4))), D)
    c = ev.sub(C, A)
    t2 = ev.square(t)
    if each and t2.shape[-3] == 3:
        t2 = ev.relinearize(t2, relin[0], relin[1])
    t3 = t2 if each else ev.multiply(t, t)                       # t3 = t * t exactly as the reference computes it (:175)
    a, b, c = mul(a, t3), mul(b, t2), mul(c, t)
    a = ev.add(ev.add(a, b), c)
    a = M(a, P(0.5))
    a = ev.add(a, B)
    return ev.relinearize(a, relin[0], relin[1]) if tail else a


def cubic(ev, pc, A, B, C, D, t, relin=None):
    """Cubic(result, A,B,C,D,t): homo/fhe_resize.h:143-189 for a batch (fhe_cubic).  Note t3 = t*t exactly as the
    reference computes it (:175).  relin=(evk_ntt, dbc) selects the relinearised mode of the library (every product
    relinearised: result of size 2); cubic_evaluator_calls(..., relin) is the same thing one Evaluator call at a time."""
    cc = circuits_of(pc, relin)
    size = A.shape[-3]
    count = _count(A, size)
    assert A.shape == B.shape == C.shape == D.shape and _count(t, 2) == count
    out = ev.ctx.empty(*A.shape[:-3], size=cc.out_size(CUBIC, size))
    nbytes = _lib.load().fhe_cubic_scratch_bytes(cc.h, size, count)
    scr = cc.scratch(nbytes)
    _lib.call("fhe_cubic", cc.h, _ptr(A), _ptr(B), _ptr(C), _ptr(D), size, _ptr(t), _ptr(out), count, _ptr(scr), nbytes, _stream())
    return out


def linear(ev, pc, A, B, t, relin=None):
    """Linear(result, A,B,t): homo/fhe_resize.h:191-204: (1 - t) A + t B (fhe_linear)."""
    cc = circuits_of(pc, relin)
    size = A.shape[-3]
    count = _count(A, size)
    assert A.shape == B.shape and _count(t, 2) == count
    out = ev.ctx.empty(*A.shape[:-3], size=cc.out_size(LINEAR, size))
    nbytes = _lib.load().fhe_linear_scratch_bytes(cc.h, size, count)
    scr = cc.scratch(nbytes)
    _lib.call("fhe_linear", cc.h, _ptr(A), _ptr(B), size, _ptr(t), _ptr(out), count, _ptr(scr), nbytes, _stream())
    return out


def resize_sample_plan(src_w, src_h, dst_w, dst_h, bicubic=True):
    """The index arithmetic of ResizeImage / SampleBicubic / SampleLinear / GetPixelClamped
    (homo/fhe_resize.h:350-351, 381-382, 260-290, 215-220), in float32 like the reference
    (fhe_resize_sample_plan).  Returns per output pixel: the clamped source pixel indices (uint32 array
    [dst_w * dst_h, 16] for bicubic, row-major 4x4; [.., 4] for bilinear) and the fractional offsets
    (xfract, yfract) as lists of floats."""
    npx = dst_w * dst_h
    taps = np.zeros((npx, 16 if bicubic else 4), dtype=np.uint32)
    fx, fy = np.zeros(npx), np.zeros(npx)
    _lib.call("fhe_resize_sample_plan", src_w, src_h, dst_w, dst_h, int(bool(bicubic)), taps.ctypes.data_as(C.c_void_p),
              fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p))
    return taps, [float(v) for v in fx], [float(v) for v in fy]


def _sample(name, width, kind, ev, pc, pixels, taps, xfract, yfract, relin=None):
    cc = circuits_of(pc, relin)
    out_size = cc.out_size(kind)
    taps = _taps_array(taps, width)
    count = taps.shape[0]
    n_pixels = _count(pixels, 2)
    assert _count(xfract, 2) == count and _count(yfract, 2) == count
    out = ev.ctx.empty(count, size=out_size)
    L = _lib.load()
    nbytes = getattr(L, name + "_scratch_bytes")(cc.h, count)
    scr = cc.scratch(nbytes)
    _lib.call(name, cc.h, _ptr(pixels), n_pixels, taps.ctypes.data_as(C.c_void_p), _ptr(xfract), _ptr(yfract), _ptr(out), count,
              _ptr(scr), nbytes, _stream())
    return out


def sample_bicubic(ev, pc, pixels, taps, xfract, yfract, relin=None):
    """SampleBicubic for a batch of output pixels and one colour channel (homo/fhe_resize.h:254-305;
    fhe_sample_bicubic).  pixels: [src_pixels, 2, k, n]; taps: [B][16] source indices; xfract/yfract: [B, 2, k, n]
    ciphertexts of the fractional offsets.  Returns [B, 6, k, n] ([B, 2, k, n] with relin=(evk_ntt, dbc))."""
    return _sample("fhe_sample_bicubic", 16, SAMPLE_BICUBIC, ev, pc, pixels, taps, xfract, yfract, relin)


def sample_linear(ev, pc, pixels, taps, xfract, yfract, relin=None):
    """SampleLinear for one channel (homo/fhe_resize.h:222-252; fhe_sample_linear).  Returns [B, 4, k, n] ([B, 2, k, n] relinearised)."""
    return _sample("fhe_sample_linear", 4, SAMPLE_LINEAR, ev, pc, pixels, taps, xfract, yfract, relin)


BAND_CONSUMER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p)


def resize_source_rows(src_h, dst_h, row0, row1, bicubic=True):
    """(first, count) of the source rows destination rows [row0, row1) read (fhe_resize_source_rows): the shard's rows plus
    the sampler's halo, clamped -- what a GPU that owns those destination rows has to load."""
    first, count = C.c_uint32(), C.c_uint32()
    _lib.call("fhe_resize_source_rows", src_h, dst_h, row0, row1, int(bool(bicubic)), C.byref(first), C.byref(count))
    return int(first.value), int(count.value)


def resize_bicubic_shared(ev, pc, pixels, src_w, src_h, dst_w, dst_h, xfract, yfract, batch=256, band_rows=4, consume=None, rows=None, src_rows=None,
                          relin=None):
    """ResizeImage with SampleBicubic (homo/fhe_resize.h:254-392) for one colour channel when the fractional
    offsets arrive as ONE ciphertext per output column (xfract [dst_w, 2, k, n]) and ONE per output row
    (yfract [dst_h, 2, k, n]) -- SURVEY.md section 8(d), config 3: "xfract/yfract ciphertexts are inputs generated
    per distinct fractional value" (fhe_resize_bicubic_shared).  frac(u) depends on x only and frac(v) on y only
    (:351,382), so with shared ciphertexts the reference's per-pixel work repeats itself and the library forms every
    repeated ring element once; each output equals sample_bicubic(..., xfract[x], yfract[y]) bit for bit:

      * a row Cubic (:296-299) is a function of (output column x, source row r) only; consecutive output rows'
        4-row windows overlap, so the 4 * dst_h row Cubics of a column collapse to one per source row touched
        (128 instead of 256 for 128 -> 64);
      * xfract^2 and the prepared (extended + transformed) forms of xfract, xfract^2 are formed once per column,
        yfract^2 and its prepared forms once per row; the products index them in place.

    The reference's server encrypts fresh offsets for every sample (:262,266); its results are therefore
    randomised per pixel and only sample_bicubic with per-pixel ciphertexts reproduces that run bit for bit
    (server.server_resize does).  Source rows are visited as a sliding window (`band_rows` output rows at a
    time), like the reference's loader (:352-379).

    rows=(y0, y1) evaluates a SHARD of the destination rows (fhe_resize_bicubic_shared_rows; the multi-GPU partition of
    the outer loop, :350): `pixels` then holds the source rows src_rows=(first, count) only (resize_source_rows: the
    shard's rows +- the halo), yfract the offsets of rows [y0, y1) only, and the result the pixels of those rows -- each
    bit-identical to the whole-image call's.

    relin=(evk_ntt, dbc): the relinearised mode (every product relinearised; outputs of size 2 instead of 6).

    Returns [dst_h * dst_w, 6, k, n] (row-major; [(y1 - y0) * dst_w, ...] for a shard), or None when
    `consume(first_pixel, tensor)` takes the bands (first_pixel is the global index y * dst_w; the tensor is a view of a
    buffer the library reuses: clone what must outlive the callback)."""
    cc = circuits_of(pc, relin)
    so = cc.out_size(SAMPLE_BICUBIC)
    ctx = ev.ctx
    y0, y1 = rows if rows is not None else (0, dst_h)
    s0, sc = src_rows if src_rows is not None else ((0, src_h) if rows is None else resize_source_rows(src_h, dst_h, y0, y1))
    assert _count(pixels, 2) == src_w * sc and _count(xfract, 2) == dst_w and _count(yfract, 2) == y1 - y0, (pixels.shape, sc, yfract.shape, rows)
    L = _lib.load()
    out = None if consume is not None else ctx.empty(dst_w * (y1 - y0), size=so)
    nbytes = L.fhe_resize_bicubic_shared_rows_scratch_bytes(cc.h, src_w, src_h, dst_w, dst_h, y0, y1, s0, sc, batch, band_rows, int(out is not None))
    scr = cc.scratch(nbytes)
    words = so * ctx.k * ctx.n
    err = []

    def on_band(_user, first, d_band, npx, _stream_):
        try:
            off = d_band - scr.data_ptr()                              # the band buffer lies inside the scratch tensor
            assert 0 <= off and off + npx * words * 8 <= scr.numel() and off % 8 == 0
            consume(int(first), scr[off:off + npx * words * 8].view(torch.int64).view(npx, so, ctx.k, ctx.n))
            return 0
        except BaseException as e:                                     # must not propagate through the C frame
            err.append(e)
            return -1

    cb = BAND_CONSUMER(on_band) if consume is not None else None
    try:
        _lib.call("fhe_resize_bicubic_shared_rows", cc.h, _ptr(pixels), src_w, src_h, dst_w, dst_h, y0, y1, s0, sc, _ptr(xfract), _ptr(yfract),
                  _ptr(out) if out is not None else C.c_void_p(None), batch, band_rows, cb, None, _ptr(scr), nbytes, _stream())
    except _lib.FheError:
        if err:
            raise err[0]
        raise
    return out


# ------------------------------------------------------------------------------------------------
# decode path
# ------------------------------------------------------------------------------------------------
def _sincos(cosine, ev, pc, x, zero, relin=None):
    cc = circuits_of(pc, relin)
    count = _count(x, 2)
    assert _count(zero, 2) == count
    out = ev.ctx.empty(*x.shape[:-3], size=cc.out_size(SINCOS))
    nbytes = _lib.load().fhe_homomorphic_sincos_scratch_bytes(cc.h, count)
    scr = cc.scratch(nbytes)
    _lib.call("fhe_homomorphic_sincos", cc.h, cosine, _ptr(x), _ptr(zero), _ptr(out), count, _ptr(scr), nbytes, _stream())
    return out


def homomorphic_sin(ev, pc, x, zero, relin=None):
    """homo/fhe_decode.h:48-120; `zero` plays the role of encrypt(encode(0.0)) (:54).  relin=(evk_ntt, dbc): every power
    relinearised where it is formed (result of size 2 instead of 11)."""
    return _sincos(0, ev, pc, x, zero, relin)


def homomorphic_cos(ev, pc, x, zero, relin=None):
    """homo/fhe_decode.h:128-200 (the reference shifts by -3pi/2 here too, :137, and falls off the end
    without a return statement, :200; the value it leaves in `res` is what is returned here)."""
    return _sincos(1, ev, pc, x, zero, relin)


def stack_zeros(zeros, npos, degree, pos0=0):
    """zeros: callable (i, j, which) -> [1, 2, k, n] -> one tensor [npos, degree, 2, 2, k, n] in the reference's
    call order (position i = pos0 .. pos0 + npos - 1, harmonic j = 1..degree, homomorphic_sin's Enc(0) then homomorphic_cos's)."""
    return torch.cat([zeros(i, j, w) for i in range(pos0, pos0 + npos) for j in range(1, degree + 1) for w in ("sin", "cos")]).contiguous()


def approximated_step(ev, pc, amplitude, index, count, order, degree, delta, width, height, zeros, positions=None, relin=None):
    """The homomorphic overload of approximated_step (homo/fhe_decode.h:202-242) for ONE run
    (fhe_approximated_step).  amplitude/index/count: [1, 2, k, n].  zeros: a tensor [npos, degree, 2, 2, k, n]
    (stack_zeros order) or a callable (i, j, which) -> [1, 2, k, n] encryption of zero for position i, harmonic j,
    which in {"sin", "cos"}.  Returns a list of width*height ciphertexts [1, 22, k, n].

    Faithful to the reference's quirk: `offset` is advanced by add_plain(offset, encode(i)) INSIDE
    the harmonic loop (:229), after cos_arg was copied from it.

    The reference walks positions x harmonics serially with one ciphertext per call; in the library only the
    (cheap) offset chain is serial.  All width*height*degree cosine polynomials are evaluated as ONE
    batch, the sine polynomial once per harmonic (its argument b * f_j does not depend on the
    position) -- the same ring operations on the same operands, so the same bits, but launches that
    fill the GPU.

    positions=(p0, p1) evaluates a SHARD of the output positions (fhe_approximated_step_range; the multi-GPU partition of
    the position loop, :224): zeros (tensor form) and the result then hold positions [p0, p1) only, each bit-identical
    to the whole-run call's.

    relin=(evk_ntt, dbc): the relinearised mode -- every power of the Taylor polynomials, the sin x cos product and the
    final product by the amplitude are relinearised, so the 11 x 11 -> 21 product of the reference's evaluation is a 2 x 2
    one and the results have 2 polynomials instead of 22."""
    cc = circuits_of(pc, relin)
    npos = width * height
    p0, p1 = positions if positions is not None else (0, npos)
    if callable(zeros):
        zeros = stack_zeros(zeros, p1 - p0, degree, p0) if degree > 0 else None
    L = _lib.load()
    so = cc.out_size(STEP, degree)
    out = ev.ctx.empty(p1 - p0, size=so)
    nbytes = L.fhe_approximated_step_range_scratch_bytes(cc.h, degree, npos, p0, p1)
    scr = cc.scratch(nbytes)
    _lib.call("fhe_approximated_step_range", cc.h, _ptr(amplitude), _ptr(index), _ptr(count), order, degree, float(delta), width, height, p0, p1,
              _ptr(zeros) if zeros is not None else C.c_void_p(None), _ptr(out), _ptr(scr), nbytes, _stream())
    return [out[i:i + 1] for i in range(p1 - p0)]


def decode_channel(ev, pc, runs, index, acc0, zeros, order, degree, delta, width, height, positions=None, relin=None):
    """One colour channel of the server_decode driver loop (homo/server_decode.cpp:120-137; fhe_decode_channel).
    runs: [pairs, 2, 2, k, n] (elem, count per run); index: [1, 2, k, n] or [2, k, n], UPDATED IN PLACE (index += count
    per run, :137); acc0: [npos, 2, k, n], the channel's Enc(0) accumulators (:126); zeros: [pairs, npos, degree, 2, 2, k, n].
    Returns [npos, S, k, n], S = 22 for degree >= 1 and pairs > 0.

    positions=(p0, p1): a SHARD of the channel's positions (fhe_decode_channel_range); acc0, zeros and the result hold
    positions [p0, p1) only, and `index` is this shard's own copy of the chain (it ends at the same value on every shard)."""
    cc = circuits_of(pc, relin)
    npos = width * height
    p0, p1 = positions if positions is not None else (0, npos)
    pairs = int(runs.shape[0]) if runs is not None else 0
    assert _count(acc0, 2) == p1 - p0 and index.is_contiguous()
    L = _lib.load()
    so = cc.out_size(DECODE, degree) if pairs else 2
    out = ev.ctx.empty(p1 - p0, size=so)
    nbytes = L.fhe_decode_channel_range_scratch_bytes(cc.h, degree, npos, p0, p1, pairs)
    scr = cc.scratch(nbytes)
    null = C.c_void_p(None)
    _lib.call("fhe_decode_channel_range", cc.h, _ptr(runs) if pairs else null, pairs, _ptr(index), _ptr(acc0),
              _ptr(zeros) if (pairs and degree > 0) else null, order, degree, float(delta), width, height, p0, p1, _ptr(out), _ptr(scr), nbytes, _stream())
    return out
