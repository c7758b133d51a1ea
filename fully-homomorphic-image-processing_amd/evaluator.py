"""Host-side mirror of the SEAL 2.3 objects the reference's circuits use, over the C ABI.

Names follow seal::{EncryptionParameters, SEALContext, FractionalEncoder, Evaluator} as used at
homo/server_jpeg.cpp:74-100 and homo/fhe_image.h:196-325.  Ciphertexts are torch int64 tensors on
the HIP device holding u64 bit patterns, shaped [..., size, k, n] (SEAL-logical order); every
Evaluator method works on whole batches (leading dimensions) in one launch.  PyTorch is used for
device memory and streams only -- all arithmetic happens in libfhe_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

# Parameter presets
PRESETS = {
    "P4096": dict(n=4096, q=[0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], t=1 << 14),
    "P8192": dict(n=8192, q=[0x7FFFFFFF380001, 0x7FFFFFFEF00001, 0x3FFFFFFF000001, 0x3FFFFFFEF40001], t=1 << 14),
    "SEAL23_4096": dict(n=4096, q=[0x7FFFFFFF380001, 0x3FFFFFFF000001], t=1 << 14),
    "SEAL23_2048": dict(n=2048, q=[0x3FFFFFFF000001], t=1 << 14),
    "SEAL23_16384": dict(n=16384, q=[0x7FFFFFFF380001, 0x7FFFFFFEF00001, 0x7FFFFFFEAC0001, 0x7FFFFFFE700001, 0x7FFFFFFE600001, 0x7FFFFFFE4C0001,
                                     0x3FFFFFFF000001, 0x3FFFFFFEF40001], t=1 << 14),      # six 55-bit + two 54-bit primes, 438 bits
    "SEAL3_8192": dict(n=8192, q=[0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001, 0xFFFFFEBC001], t=1 << 14),
}
YQT = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
       14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
       49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]  # homo/fhe_image.h:99
SEED = 0x5EA12026


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_EMPTY = {}


def _ptr(t):
    """device pointer of a tensor for the C ABI.  An EMPTY tensor has no storage (data_ptr() == 0) while the library refuses null pointers before it
    looks at the count: an empty batch (a rank whose shard of a small image holds no block, a channel without runs) gets the address of a
    one-word placeholder on the same device instead -- the call is then the no-op the C ABI defines for count == 0."""
    if t.numel() == 0:
        key = (t.device.type, t.device.index)
        if key not in _EMPTY:
            _EMPTY[key] = torch.zeros(2, dtype=torch.int64, device=t.device)
        return C.c_void_p(_EMPTY[key].data_ptr())
    return C.c_void_p(t.data_ptr())


def to_device(arr, device="cuda"):
    """numpy uint64 -> device int64 tensor with the same bits."""
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t):
    """device int64 tensor -> numpy uint64."""
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


class SEALContext:
    """EncryptionParameters + SEALContext: poly_modulus_degree n, coeff_modulus q[], plain_modulus t."""

    def __init__(self, n, q, t, device=0, switches=None):
        """switches: {"FHE_DCT_FORCE_U64": "1", ...} -- experiment switches for THIS context (the library reads them
        from the environment once, in fhe_ctx_create; they are set around that call only).  Tests and A/B runs."""
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device: this framework has no CPU path")
        self.n, self.q, self.t, self.k = int(n), [int(x) for x in q], int(t), len(q)
        self.device = torch.device("cuda", device)
        arr = (C.c_uint64 * self.k)(*self.q)
        h = C.c_void_p()
        import os
        saved = {name: os.environ.get(name) for name in (switches or {})}
        try:
            for name, value in (switches or {}).items():
                os.environ[name] = str(value)
            _lib.call("fhe_ctx_create", self.n, arr, self.k, self.t, device, C.byref(h))
        finally:
            for name, value in saved.items():
                if value is None:
                    os.environ.pop(name, None)
                else:
                    os.environ[name] = value
        self.h = h

    @classmethod
    def preset(cls, name, device=0, switches=None):
        p = PRESETS[name]
        return cls(p["n"], p["q"], p["t"], device, switches)

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                _lib.load().fhe_ctx_destroy(h)
            except Exception:
                pass
            self.h = None

    def ct_shape(self, *lead, size=2):
        return tuple(lead) + (size, self.k, self.n)

    def empty(self, *lead, size=2):
        return torch.empty(self.ct_shape(*lead, size=size), dtype=torch.int64, device=self.device)

    def random_ct(self, *lead, size=2, seed=SEED, first_index=0):
        """Synthetic ciphertexts: splitmix64(seed ^ linear_index) mod q_i."""
        out = self.empty(*lead, size=size)
        n_polys = out.numel() // (self.k * self.n)
        _lib.call("fhe_fill_random", self.h, _ptr(out), n_polys, seed, first_index, _stream())
        return out

    def digest(self, t, index0=0):
        out = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.call("fhe_digest", self.h, _ptr(t), t.numel(), index0, _ptr(out), _stream())
        return int(out.cpu().numpy().view(np.uint64)[0])


    def digest_into(self, t, out_elem, index0=0):
        """asynchronous form: the digest of `t` is written to the one-element int64 device tensor `out_elem`"""
        _lib.call("fhe_digest", self.h, _ptr(t), t.numel(), index0, _ptr(out_elem), _stream())


class FractionalEncoder:
    """seal::FractionalEncoder(t, poly_modulus, 100, 100, 2) (homo/server_jpeg.cpp:100)."""

    def __init__(self, ctx, int_coeffs=100, frac_coeffs=100):
        self.ctx, self.int_coeffs, self.frac_coeffs = ctx, int_coeffs, frac_coeffs

    def encode(self, value):
        out = np.zeros(self.ctx.n, dtype=np.uint64)
        _lib.call("fhe_frac_encode", self.ctx.n, self.ctx.t, float(value), self.int_coeffs, self.frac_coeffs,
                  out.ctypes.data_as(C.c_void_p))
        return out

    def decode(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        return float(_lib.load().fhe_frac_decode(self.ctx.n, self.ctx.t, p.ctypes.data_as(C.c_void_p),
                                                 self.int_coeffs, self.frac_coeffs))


SPARSE_MAX_TERMS = 8       # FHE_SPARSE_MAX_TERMS in include/fhe_hip.h


class PreparedPlain:
    """A Plaintext lifted to the q-base and transformed once (the reference redoes this per call)."""

    def __init__(self, ctx, plain):
        self.ctx = ctx
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        nz = np.flatnonzero(p)
        ln = int(nz[-1]) + 1 if nz.size else 0     # significant coefficient count
        self.buf = torch.empty(2 * ctx.k * ctx.n, dtype=torch.int64, device=ctx.device)
        _lib.call("fhe_plain_prepare", ctx.h, p.ctypes.data_as(C.c_void_p), ln, _ptr(self.buf), _stream())
        # few non-zero coefficients (encode(3) = x+1, encode(0.5) = -x^(n-1), ...): the product is a sum
        # of signed rotations, done without any transform (fhe_multiply_plain_sparse)
        self.plain = p[:ln].copy()
        self.sparse = 0 < int(np.count_nonzero(self.plain)) <= SPARSE_MAX_TERMS and ctx.n <= 8192


class PreparedCt:
    """A multiply() operand in prepared form (Evaluator.prepare_operand); opaque device words."""

    def __init__(self, buf, size, lead):
        self.buf, self.size, self.lead = buf, size, lead

    def shared(self, div=1, first=0):
        return SharedPrepared(self, div, first)

    def gather(self, index):
        """The prepared operands `index` (a sequence of positions in the flattened batch, repeats allowed) as a new
        prepared batch -- a copy of words, no arithmetic.  Layout (include/fhe_hip.h, fhe_multiply_prepare):
        [count][size][k][n] over the coefficient base followed by [count][size][k+1][n] over the auxiliary base."""
        count = 1
        for d in self.lead:
            count *= d
        per = self.buf.numel() // count                       # size * (2k + 1) * n
        k = getattr(self, "_k", None)
        if k is None:
            raise ValueError("gather needs a PreparedCt made by Evaluator.prepare_operand")
        qw = per // (2 * k + 1) * k                           # size * k * n
        bw = per - qw                                         # size * (k + 1) * n
        idx = torch.as_tensor(index, dtype=torch.long, device=self.buf.device)
        m = int(idx.numel())
        buf = torch.empty(m * per, dtype=self.buf.dtype, device=self.buf.device)
        torch.index_select(self.buf[:count * qw].view(count, qw), 0, idx, out=buf[:m * qw].view(m, qw))
        torch.index_select(self.buf[count * qw:].view(count, bw), 0, idx, out=buf[m * qw:].view(m, bw))
        out = PreparedCt(buf, self.size, (m,))
        out._k = self._k
        return out


class SharedPrepared:
    """A prepared operand batch shared between the pairs of one multiply(): pair c takes entry
    (first + c // div) % count (fhe_multiply_prepared_shared); PreparedCt.shared(div, first) makes one.
    Right-hand side of multiply() only."""

    def __init__(self, prep, div, first=0):
        count = 1
        for d in prep.lead:
            count *= d
        self.prep, self.div, self.first, self.count, self.size = prep, int(div), int(first), count, prep.size


class DctPlan:
    def __init__(self, ctx, quant=YQT, int_coeffs=100, frac_coeffs=100):
        self.ctx = ctx
        h = C.c_void_p()
        qv = None
        if quant is not None:
            qv = (C.c_double * 64)(*[float(x) for x in quant])
        _lib.call("fhe_dct_plan_create", ctx.h, qv, int_coeffs, frac_coeffs, _stream(), C.byref(h))
        self.h = h

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                _lib.load().fhe_dct_plan_destroy(h)
            except Exception:
                pass
            self.h = None


def check_evaluation_keys(ctx, evk_ntt, dbc, need, who):
    """The library takes the keys as a bare pointer and reads need * fhe_evk_words(ctx, dbc) words behind it (include/fhe_hip.h): the host
    checks that the tensor it hands over holds them -- a key tensor made for another decomposition bit count (fewer digits), another context or
    fewer powers must be an error here, not a read behind the allocation."""
    if not 1 <= int(dbc) <= 60:
        raise ValueError("%s: decomposition bit count %r (1 .. 60)" % (who, dbc))
    if evk_ntt.dtype != torch.int64 or not evk_ntt.is_contiguous() or evk_ntt.device != ctx.device:
        raise ValueError("%s: evaluation keys must be a contiguous int64 tensor on the context's device" % who)
    words = int(_lib.load().fhe_evk_words(ctx.h, int(dbc)))
    if evk_ntt.numel() < need * words:
        raise ValueError("%s: %d key set(s) at dbc %d take %d words on this context, the tensor holds %d (keys generated with another "
                         "decomposition bit count, for another context, or for fewer powers)" % (who, need, dbc, need * words, evk_ntt.numel()))
    nd = int(_lib.load().fhe_evk_digits(ctx.h, int(dbc)))
    if evk_ntt.dim() >= 5 and tuple(evk_ntt.shape[-5:]) != (ctx.k, nd, 2, ctx.k, ctx.n):
        raise ValueError("%s: evaluation keys of shape %r, this context at dbc %d has [k = %d][digits = %d][2][k][n = %d]"
                         % (who, tuple(evk_ntt.shape), dbc, ctx.k, nd, ctx.n))


class Evaluator:
    """seal::Evaluator over batches.  In-place semantics of SEAL are expressed functionally:
    every method returns a new tensor unless `out=` is given (which may alias an input)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._scratch = None

    # -- helpers ---------------------------------------------------------------------------------
    def _npolys(self, t):
        kn = self.ctx.k * self.ctx.n
        assert t.dtype == torch.int64 and t.is_contiguous() and t.numel() % kn == 0
        return t.numel() // kn

    def _scratch_buf(self, nbytes):
        if self._scratch is None or self._scratch.numel() < nbytes:
            self._scratch = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=self.ctx.device)
        return self._scratch

    def _binary(self, name, a, b, out):
        sa, sb = a.shape[-3], b.shape[-3]
        kn = self.ctx.k * self.ctx.n
        if tuple(a.shape[-2:]) != (self.ctx.k, self.ctx.n) or tuple(b.shape[-2:]) != (self.ctx.k, self.ctx.n) or a.numel() // (sa * kn) != b.numel() // (sb * kn):
            raise ValueError("%s: operands of shapes %r and %r are not the same number of ciphertexts of this context (the library reads both with the first one's count)"
                             % (name, tuple(a.shape), tuple(b.shape)))
        if sa == sb:
            if out is not None and (out.numel() != a.numel() or out.dtype != a.dtype or not out.is_contiguous()):
                raise ValueError("%s: `out` must be a contiguous int64 tensor of %d words, got %r" % (name, a.numel(), tuple(out.shape)))
            out = torch.empty_like(a) if out is None else out
            _lib.call(name, self.ctx.h, _ptr(a), _ptr(b), _ptr(out), self._npolys(a), _stream())
            return out
        # unequal sizes: common prefix through the kernel, tail copied (add) or negated (sub of b's tail)
        lead = a.shape[:-3]
        s, m = max(sa, sb), min(sa, sb)
        res = self.ctx.empty(*lead, size=s)
        pa, pb = a[..., :m, :, :].contiguous(), b[..., :m, :, :].contiguous()
        pre = torch.empty_like(pa)
        _lib.call(name, self.ctx.h, _ptr(pa), _ptr(pb), _ptr(pre), self._npolys(pa), _stream())
        res[..., :m, :, :] = pre
        if sa > sb:
            res[..., m:, :, :] = a[..., m:, :, :]
        else:
            tail = b[..., m:, :, :].contiguous()
            if name == "fhe_sub":
                neg = torch.empty_like(tail)
                _lib.call("fhe_negate", self.ctx.h, _ptr(tail), _ptr(neg), self._npolys(tail), _stream())
                tail = neg
            res[..., m:, :, :] = tail
        return res

    # -- seal::Evaluator surface -------------------------------------------------------------------
    def add(self, a, b, out=None):
        return self._binary("fhe_add", a, b, out)

    def sub(self, a, b, out=None):
        return self._binary("fhe_sub", a, b, out)

    def negate(self, a, out=None):
        out = torch.empty_like(a) if out is None else out
        _lib.call("fhe_negate", self.ctx.h, _ptr(a), _ptr(out), self._npolys(a), _stream())
        return out

    def multiply_plain(self, a, plain, out=None):
        if not isinstance(plain, PreparedPlain):
            plain = PreparedPlain(self.ctx, plain)
        out = torch.empty_like(a) if out is None else out
        if plain.sparse:
            _lib.call("fhe_multiply_plain_sparse", self.ctx.h, _ptr(a), _ptr(out), self._npolys(a),
                      plain.plain.ctypes.data_as(C.c_void_p), len(plain.plain), _stream())
        else:
            _lib.call("fhe_multiply_plain", self.ctx.h, _ptr(a), _ptr(out), self._npolys(a), _ptr(plain.buf), _stream())
        return out

    def _plain_addsub(self, a, plain, sign):
        out = a.clone()
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        nz = np.flatnonzero(p)
        ln = int(nz[-1]) + 1 if nz.size else 0     # significant coefficient count
        size = a.shape[-3]
        stride = size * self.ctx.k * self.ctx.n
        count = out.numel() // stride
        _lib.call("fhe_add_plain", self.ctx.h, _ptr(out), stride, count, p.ctypes.data_as(C.c_void_p), ln, sign, _stream())
        return out

    def add_plain(self, a, plain):
        return self._plain_addsub(a, plain, 1)

    def sub_plain(self, a, plain):
        return self._plain_addsub(a, plain, -1)

    def prepare_operand(self, a):
        """Extend a ciphertext batch to the auxiliary base and transform it once, for several multiply()
        calls with the same operand (fhe_multiply_prepare); multiply() accepts the result on either side."""
        size = a.shape[-3]
        lead = tuple(a.shape[:-3])
        count = 1
        for d in lead:
            count *= d
        words = _lib.load().fhe_multiply_operand_words(self.ctx.h, size, count)
        buf = torch.empty(words, dtype=torch.int64, device=self.ctx.device)
        _lib.call("fhe_multiply_prepare", self.ctx.h, _ptr(a), size, count, _ptr(buf), _stream())
        out = PreparedCt(buf, size, lead)
        out._k = self.ctx.k
        return out

    def multiply(self, a, b):
        if isinstance(b, SharedPrepared):
            pa = a if isinstance(a, PreparedCt) else None
            sa = pa.size if pa else a.shape[-3]
            lead = pa.lead if pa else tuple(a.shape[:-3])
            count = 1
            for d in lead:
                count *= d
            out = self.ctx.empty(*lead, size=sa + b.size - 1)
            nbytes = _lib.load().fhe_multiply_scratch_bytes(self.ctx.h, sa, b.size, count)
            scr = self._scratch_buf(nbytes)
            null = C.c_void_p(None)
            _lib.call("fhe_multiply_prepared_shared", self.ctx.h, null if pa else _ptr(a), _ptr(pa.buf) if pa else null, sa,
                      _ptr(b.prep.buf), b.size, b.count, b.div, b.first, _ptr(out), count, _ptr(scr), nbytes, _stream())
            return out
        pa = a if isinstance(a, PreparedCt) else None
        pb = b if isinstance(b, PreparedCt) else None
        sa = pa.size if pa else a.shape[-3]
        sb = pb.size if pb else b.shape[-3]
        lead = pa.lead if pa else tuple(a.shape[:-3])
        assert (pb.lead if pb else tuple(b.shape[:-3])) == lead
        count = 1
        for d in lead:
            count *= d
        out = self.ctx.empty(*lead, size=sa + sb - 1)
        nbytes = _lib.load().fhe_multiply_scratch_bytes(self.ctx.h, sa, sb, count)
        scr = self._scratch_buf(nbytes)
        if pa is None and pb is None:
            _lib.call("fhe_multiply", self.ctx.h, _ptr(a), sa, _ptr(b), sb, _ptr(out), count, _ptr(scr), nbytes, _stream())
        else:
            null = C.c_void_p(None)
            _lib.call("fhe_multiply_prepared", self.ctx.h, null if pa else _ptr(a), _ptr(pa.buf) if pa else null, sa,
                      null if pb else _ptr(b), _ptr(pb.buf) if pb else null, sb, _ptr(out), count, _ptr(scr), nbytes, _stream())
        return out

    def square(self, a):
        sa = a.shape[-3]
        lead = a.shape[:-3]
        count = 1
        for d in lead:
            count *= d
        out = self.ctx.empty(*lead, size=2 * sa - 1)
        nbytes = _lib.load().fhe_multiply_scratch_bytes(self.ctx.h, sa, sa, count)
        scr = self._scratch_buf(nbytes)
        _lib.call("fhe_square", self.ctx.h, _ptr(a), sa, _ptr(out), count, _ptr(scr), nbytes, _stream())
        return out

    def relinearize(self, a, evk_ntt, dbc):
        """evaluator.relinearize(a, evk): ciphertexts of any size >= 2 down to 2, one key switch per polynomial above the second, the top
        one first (SEAL 2.3).  evk_ntt: KeyGenerator.generate_evaluation_keys(dbc) for size 3, generate_evaluation_keys(dbc, size - 2)
        ([size - 2][k][digits][2][k][n]: keys for s^2 .. s^(size-1)) above."""
        size = a.shape[-3]
        if size == 2:
            return a
        kn = self.ctx.k * self.ctx.n
        have = evk_ntt.shape[0] if evk_ntt.dim() == 6 else 1
        if have < size - 2:
            raise ValueError("relinearize: a ciphertext of %d polynomials needs the keys for s^2 .. s^%d (got %d key set(s))" % (size, size - 1, have))
        check_evaluation_keys(self.ctx, evk_ntt, dbc, size - 2, "relinearize")
        work = a.clone()                                   # the steps above the last one run in place
        count = work.numel() // (size * kn)
        out = torch.empty(tuple(a.shape[:-3]) + (2, self.ctx.k, self.ctx.n), dtype=a.dtype, device=a.device)
        nbytes = _lib.load().fhe_relinearize_n_scratch_bytes(self.ctx.h, size, dbc, count)
        scr = self._scratch_buf(nbytes)
        _lib.call("fhe_relinearize_n", self.ctx.h, _ptr(work), size, size * kn, _ptr(out), 2 * kn, count, _ptr(evk_ntt), dbc, _ptr(scr), nbytes, _stream())
        return out

    # -- primitives named by the north star ---------------------------------------------------------
    def cubic_coeffs(self, A, B, C, D):
        """a = 3B - A - 3C + D, b = 2A - 5B + 4C - D, c = C - A of Cubic (homo/fhe_resize.h:150-172) in one
        pass; valid for the base-2 FractionalEncoder (encode(3) = x+1, ...), which the caller checks."""
        size = A.shape[-3]
        count = A.numel() // (size * self.ctx.k * self.ctx.n)
        a, b, c = torch.empty_like(A), torch.empty_like(A), torch.empty_like(A)
        _lib.call("fhe_cubic_coeffs", self.ctx.h, _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(a), _ptr(b), _ptr(c), size, count, _stream())
        return a, b, c

    def cubic_combine(self, a, b, c, B):
        """0.5 (a + b + c) + B of Cubic (homo/fhe_resize.h:181-188); a, b, c of equal size >= size of B."""
        size_abc, size_b = a.shape[-3], B.shape[-3]
        count = a.numel() // (size_abc * self.ctx.k * self.ctx.n)
        out = torch.empty_like(a)
        _lib.call("fhe_cubic_combine", self.ctx.h, _ptr(a), _ptr(b), _ptr(c), size_abc, _ptr(B), size_b, _ptr(out), count, _stream())
        return out

    def ntt_forward(self, a, out=None):
        out = torch.empty_like(a) if out is None else out
        _lib.call("fhe_ntt_forward", self.ctx.h, _ptr(a), _ptr(out), self._npolys(a), _stream())
        return out

    def ntt_inverse(self, a, out=None):
        out = torch.empty_like(a) if out is None else out
        _lib.call("fhe_ntt_inverse", self.ctx.h, _ptr(a), _ptr(out), self._npolys(a), _stream())
        return out

    def dyadic_multiply(self, a, b, out=None):
        out = torch.empty_like(a) if out is None else out
        _lib.call("fhe_dyadic_multiply", self.ctx.h, _ptr(a), _ptr(b), _ptr(out), self._npolys(a), _stream())
        return out

    # -- fused circuits ---------------------------------------------------------------------------
    def dct8x8_quant(self, plan, blocks, out=None):
        """encrypted_dct + quantize_fhe on [n_blocks, 64, 2, k, n] (homo/fhe_image.h:196-305)."""
        assert blocks.shape[-4:] == (64, 2, self.ctx.k, self.ctx.n) and blocks.is_contiguous()
        if out is not None and (out.shape != blocks.shape or out.dtype != blocks.dtype or not out.is_contiguous() or out.device != blocks.device):
            raise ValueError("dct8x8_quant: `out` must be a contiguous tensor like `blocks` (%r), got %r" % (tuple(blocks.shape), tuple(out.shape)))
        out = torch.empty_like(blocks) if out is None else out
        n_blocks = blocks.numel() // (64 * 2 * self.ctx.k * self.ctx.n)
        nbytes = _lib.load().fhe_dct8x8_scratch_bytes(self.ctx.h, n_blocks)
        scr = self._scratch_buf(nbytes)
        _lib.call("fhe_dct8x8_quant", self.ctx.h, plan.h, _ptr(blocks), _ptr(out), n_blocks, _ptr(scr), nbytes, _stream())
        return out

    def rgb_to_ycc(self, r, g, b, int_coeffs=100, frac_coeffs=100):
        """rgb_to_ycc_fhe on [count, 2, k, n] tensors, in place (homo/fhe_image.h:310-325)."""
        if not (r.shape == g.shape == b.shape and tuple(r.shape[-3:]) == (2, self.ctx.k, self.ctx.n) and r.is_contiguous() and g.is_contiguous() and b.is_contiguous()):
            raise ValueError("rgb_to_ycc: three contiguous tensors of one shape [..., 2, k, n], got %r, %r, %r" % (tuple(r.shape), tuple(g.shape), tuple(b.shape)))
        count = r.numel() // (2 * self.ctx.k * self.ctx.n)
        _lib.call("fhe_rgb_to_ycc", self.ctx.h, _ptr(r), _ptr(g), _ptr(b), count, int_coeffs, frac_coeffs, _stream())
        return r, g, b

    def rgb_to_ycc_blocks(self, blocks, int_coeffs=100, frac_coeffs=100):
        """rgb_to_ycc_fhe on the stream layout [n_blocks, 3, 64, 2, k, n] (64 R, 64 G, 64 B per block), in place."""
        assert blocks.shape[-5:] == (3, 64, 2, self.ctx.k, self.ctx.n) and blocks.is_contiguous()
        n_blocks = blocks.numel() // (3 * 64 * 2 * self.ctx.k * self.ctx.n)
        _lib.call("fhe_rgb_to_ycc_blocks", self.ctx.h, _ptr(blocks), n_blocks, int_coeffs, frac_coeffs, _stream())
        return blocks
