"""ctypes binding of the CPU ORACLE (oracle/libfhe_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
See oracle/fhe_oracle.h for what it restates and its PARITY STATUS (JPEG path pinned by the
reference's published outputs; ciphertext bits and the ct x ct ops "parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfhe_oracle.so")

# Parameter presets.  S3 is what BASELINE.json configs 2/5 name
# ("n=4096, 3 coeff moduli"); SEAL23_* are the SEAL 2.3.1 coeff_modulus_128 defaults.
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


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "fhe_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None
_libs = {}
_OMP_LIB_PATH = os.path.join(_HERE, "libfhe_oracle_omp.so")
u64p = C.POINTER(C.c_uint64)


def lib(omp=False):
    """the oracle library: the portable single-threaded build, or (omp=True) the optional OpenMP / AVX2 build that only
    parallelises fo_dct_quant_blocks; same source file, same results"""
    global _lib
    if omp:
        if not os.path.exists(_OMP_LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libfhe_oracle_omp.so"])
    if _libs.get(bool(omp)) is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_OMP_LIB_PATH if omp else _LIB_PATH)
        L.fo_ctx_create.restype = C.c_void_p
        L.fo_ctx_create.argtypes = [C.c_uint32, u64p, C.c_uint32, C.c_uint64]
        L.fo_ctx_destroy.argtypes = [C.c_void_p]
        L.fo_ctx_aux.restype = C.c_uint64
        L.fo_ctx_aux.argtypes = [C.c_void_p, C.c_uint32]
        L.fo_splitmix64.restype = C.c_uint64
        L.fo_splitmix64.argtypes = [C.c_uint64]
        L.fo_mulmod_const_check.restype = C.c_uint64
        L.fo_mulmod_const_check.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.fo_fill_random_ct.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        for name in ("fo_ntt_fwd", "fo_ntt_inv"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        for name in ("fo_add", "fo_sub"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.fo_negate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        for name in ("fo_add_plain", "fo_sub_plain"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fo_multiply_plain.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.fo_plain_lift.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.fo_multiply.restype = C.c_uint32
        L.fo_multiply.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.fo_square.restype = C.c_uint32
        L.fo_square.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.fo_frac_encode.restype = C.c_uint32
        L.fo_frac_encode.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p]
        L.fo_frac_decode.restype = C.c_double
        L.fo_frac_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.fo_keygen.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.fo_encrypt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.fo_chacha20_block.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.fo_noise_cdt.argtypes = [C.c_void_p]
        L.fo_encrypt_draws.argtypes = [C.c_uint32, C.c_char_p, C.c_uint64, C.c_void_p]
        L.fo_encrypt_with_draws.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.fo_encrypt_keyed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64, C.c_void_p]
        L.fo_decrypt_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.fo_decrypt.restype = C.c_int
        L.fo_decrypt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.fo_decrypt_noise_bits.restype = C.c_int
        L.fo_decrypt_noise_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.fo_evk_digits.restype = C.c_uint32
        L.fo_evk_digits.argtypes = [C.c_void_p, C.c_uint32]
        L.fo_evk_gen.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.fo_relinearize3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fo_evk_gen_pow.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]
        L.fo_relinearize_poly.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.fo_encrypted_dct.argtypes = [C.c_void_p, C.c_void_p]
        L.fo_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_dct_quant_blocks.restype = C.c_int
        L.fo_dct_quant_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.fo_rgb_to_ycc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_cubic.restype = C.c_uint32
        L.fo_cubic.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p, C.c_void_p]
        L.fo_linear.restype = C.c_uint32
        L.fo_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.fo_digest.restype = C.c_uint64
        L.fo_digest.argtypes = [C.c_void_p, C.c_uint64]
        _libs[bool(omp)] = L
        if not omp:
            _lib = L
    return _libs[bool(omp)]


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _pb(a):                      # byte-sized arrays (sampler draws, cipher blocks)
    assert a.dtype in (np.int8, np.uint8) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """SEAL-Evaluator-shaped CPU oracle over numpy u64 arrays laid out [size][k][n]."""

    INT_COEFFS = 100   # FractionalEncoder(t, poly, 100, 100, 2): homo/server_jpeg.cpp:100
    FRAC_COEFFS = 100

    def __init__(self, n, q, t, omp=False):
        """omp=True: the OpenMP / AVX2 build (oracle/libfhe_oracle_omp.so) -- only bench.py's all-cores CPU baseline
        asks for it; every parity test runs on the portable single-threaded build."""
        self.n, self.q, self.t, self.k = int(n), [int(x) for x in q], int(t), len(q)
        self.L = lib(omp)
        arr = (C.c_uint64 * self.k)(*self.q)
        self.h = self.L.fo_ctx_create(self.n, arr, self.k, self.t)
        if not self.h:
            raise ValueError("invalid encryption parameters")
        self.aux = [int(self.L.fo_ctx_aux(self.h, i)) for i in range(self.k + 1)]

    @classmethod
    def preset(cls, name, omp=False):
        p = PRESETS[name]
        return cls(p["n"], p["q"], p["t"], omp)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.fo_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- helpers --------------------------------------------------------
    def ct_words(self, size=2):
        return size * self.k * self.n

    def random_ct(self, n_cts, size=2, seed=SEED, first_index=0):
        out = np.empty((n_cts, size, self.k, self.n), dtype=np.uint64)
        self.L.fo_fill_random_ct(self.h, _p(out), n_cts * size, seed, first_index)
        return out

    # -- ntt ------------------------------------------------------------
    def ntt_fwd(self, a, prime, base=0):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.L.fo_ntt_fwd(self.h, base, prime, _p(a))
        return a

    def ntt_inv(self, a, prime, base=0):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.L.fo_ntt_inv(self.h, base, prime, _p(a))
        return a

    # -- evaluator --------------------------------------------------------
    def _grow(self, a, size):
        if a.shape[0] >= size:
            return np.ascontiguousarray(a).copy()
        out = np.zeros((size,) + a.shape[1:], dtype=np.uint64)
        out[: a.shape[0]] = a
        return out

    def add(self, a, b):
        s = max(a.shape[0], b.shape[0])
        out = self._grow(a, s)
        self.L.fo_add(self.h, _p(out), a.shape[0], _p(np.ascontiguousarray(b)), b.shape[0])
        return out

    def sub(self, a, b):
        s = max(a.shape[0], b.shape[0])
        out = self._grow(a, s)
        self.L.fo_sub(self.h, _p(out), a.shape[0], _p(np.ascontiguousarray(b)), b.shape[0])
        return out

    def negate(self, a):
        out = np.ascontiguousarray(a).copy()
        self.L.fo_negate(self.h, _p(out), a.shape[0])
        return out

    def _plain(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        return p, len(p)

    def add_plain(self, a, plain):
        out = np.ascontiguousarray(a).copy()
        p, ln = self._plain(plain)
        self.L.fo_add_plain(self.h, _p(out), _p(p), ln)
        return out

    def sub_plain(self, a, plain):
        out = np.ascontiguousarray(a).copy()
        p, ln = self._plain(plain)
        self.L.fo_sub_plain(self.h, _p(out), _p(p), ln)
        return out

    def multiply_plain(self, a, plain):
        out = np.ascontiguousarray(a).copy()
        p, ln = self._plain(plain)
        self.L.fo_multiply_plain(self.h, _p(out), a.shape[0], _p(p), ln)
        return out

    def plain_lift(self, plain):
        p, ln = self._plain(plain)
        out = np.zeros((self.k, self.n), dtype=np.uint64)
        self.L.fo_plain_lift(self.h, _p(p), ln, _p(out))
        return out

    def multiply(self, a, b):
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        out = np.zeros((a.shape[0] + b.shape[0] - 1, self.k, self.n), dtype=np.uint64)
        self.L.fo_multiply(self.h, _p(a), a.shape[0], _p(b), b.shape[0], _p(out))
        return out

    def square(self, a):
        a = np.ascontiguousarray(a)
        out = np.zeros((2 * a.shape[0] - 1, self.k, self.n), dtype=np.uint64)
        self.L.fo_square(self.h, _p(a), a.shape[0], _p(out))
        return out

    # -- encoder ----------------------------------------------------------
    def encode(self, v):
        out = np.zeros(self.n, dtype=np.uint64)
        if self.L.fo_frac_encode(self.h, float(v), self.INT_COEFFS, self.FRAC_COEFFS, _p(out)) == 0xFFFFFFFF:
            raise ValueError("encode(%r) needs more fractional coefficients than n = %d holds" % (v, self.n))
        return out

    def decode(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        return float(self.L.fo_frac_decode(self.h, _p(p), self.INT_COEFFS, self.FRAC_COEFFS))

    # -- keys -------------------------------------------------------------
    def keygen(self, seed=1):
        sk = np.zeros((self.k, self.n), dtype=np.uint64)
        pk = np.zeros((2, self.k, self.n), dtype=np.uint64)
        self.L.fo_keygen(self.h, seed, _p(sk), _p(pk))
        return sk, pk

    def encrypt(self, pk, plain, seed=7):
        p, ln = self._plain(plain)
        ct = np.zeros((2, self.k, self.n), dtype=np.uint64)
        self.L.fo_encrypt(self.h, _p(pk), _p(p), ln, seed, _p(ct))
        return ct

    # the keyed sampler of include/fhe_hip.h ("server-side encryptions"): key = 32 bytes, index = number of the encryption under that key
    def encrypt_draws(self, key, index):
        d = np.zeros((3, self.n), dtype=np.int8)
        self.L.fo_encrypt_draws(self.n, bytes(key), int(index), _pb(d))
        return d

    def encrypt_with_draws(self, pk, plain, draws):
        p, ln = self._plain(plain)
        ct = np.zeros((2, self.k, self.n), dtype=np.uint64)
        self.L.fo_encrypt_with_draws(self.h, _p(np.ascontiguousarray(pk)), _p(p), ln, _pb(np.ascontiguousarray(draws, dtype=np.int8)), _p(ct))
        return ct

    def encrypt_keyed(self, pk, plain, key, index):
        p, ln = self._plain(plain)
        ct = np.zeros((2, self.k, self.n), dtype=np.uint64)
        self.L.fo_encrypt_keyed(self.h, _p(np.ascontiguousarray(pk)), _p(p), ln, bytes(key), int(index), _p(ct))
        return ct

    def decrypt(self, sk, ct):
        ct = np.ascontiguousarray(ct)
        plain = np.zeros(self.n, dtype=np.uint64)
        budget = self.L.fo_decrypt(self.h, _p(sk), _p(ct), ct.shape[0], _p(plain))
        return plain, int(budget)

    def decrypt_noise_bits(self, sk, ct):
        """(plain, bit length of the largest |t x - m q|, bit length of q): the raw figures behind the noise budget"""
        ct = np.ascontiguousarray(ct)
        plain = np.zeros(self.n, dtype=np.uint64)
        nb, mb = C.c_int(0), C.c_int(0)
        self.L.fo_decrypt_noise_bits(self.h, _p(np.ascontiguousarray(sk)), _p(ct), ct.shape[0], _p(plain), C.byref(nb), C.byref(mb))
        return plain, nb.value, mb.value

    def decrypt_phase(self, sk, ct):
        ct = np.ascontiguousarray(ct)
        ph = np.zeros((self.k, self.n), dtype=np.uint64)
        self.L.fo_decrypt_phase(self.h, _p(sk), _p(ct), ct.shape[0], _p(ph))
        return ph

    def evk_gen(self, sk, dbc=30, seed=11):
        nd = int(self.L.fo_evk_digits(self.h, dbc))
        evk = np.zeros((self.k, nd, 2, self.k, self.n), dtype=np.uint64)
        self.L.fo_evk_gen(self.h, _p(sk), dbc, seed, _p(evk))
        return evk

    def relinearize(self, ct, evk, dbc=30):
        assert ct.shape[0] == 3
        out = np.ascontiguousarray(ct).copy()
        self.L.fo_relinearize3(self.h, _p(out), _p(evk), dbc)
        return out[:2].copy()

    def evk_gen_powers(self, sk, dbc=30, count=2, seed=11):
        """[count][k][nd][2][k][n]: keys for s^2 .. s^(count+1) (SEAL 2.3 generate_evaluation_keys(dbc, count, keys)); entry 0 is evk_gen's"""
        nd = int(self.L.fo_evk_digits(self.h, dbc))
        evk = np.zeros((count, self.k, nd, 2, self.k, self.n), dtype=np.uint64)
        for j in range(count):
            self.L.fo_evk_gen_pow(self.h, _p(sk), j + 2, dbc, seed, _p(evk[j]))
        return evk

    def relinearize_n(self, ct, evks, dbc=30):
        """evaluator.relinearize of a ciphertext of any size >= 2 down to 2 (SEAL 2.3: one key-switch step per polynomial above the
        second, the top one first, with the keys for s^(size-1)); evks: evk_gen_powers(..)"""
        out = np.ascontiguousarray(ct).copy()
        assert out.shape[0] - 2 <= evks.shape[0], "not enough evaluation keys for a ciphertext of this size"
        for p in range(out.shape[0] - 1, 1, -1):
            self.L.fo_relinearize_poly(self.h, _p(out), p, _p(evks[p - 2]), dbc)
        return out[:2].copy()

    # -- circuits ---------------------------------------------------------
    def encrypted_dct(self, block):
        out = np.ascontiguousarray(block).copy()
        assert out.shape == (64, 2, self.k, self.n)
        self.L.fo_encrypted_dct(self.h, _p(out))
        return out

    def quantize(self, block, quant=YQT):
        out = np.ascontiguousarray(block).copy()
        qv = (C.c_double * 64)(*[float(x) for x in quant])
        self.L.fo_quantize(self.h, _p(out), qv)
        return out

    def dct_quant(self, block, quant=YQT):
        return self.quantize(self.encrypted_dct(block), quant)

    def dct_quant_blocks(self, blocks, quant=YQT):
        """dct_quant on [n_blocks, 64, 2, k, n], OpenMP over blocks; returns (result, threads used)"""
        out = np.ascontiguousarray(blocks).copy()
        qv = (C.c_double * 64)(*[float(x) for x in quant])
        threads = self.L.fo_dct_quant_blocks(self.h, _p(out), out.shape[0], qv)
        return out, int(threads)

    def rgb_to_ycc(self, r, g, b):
        r, g, b = (np.ascontiguousarray(x).copy() for x in (r, g, b))
        self.L.fo_rgb_to_ycc(self.h, _p(r), _p(g), _p(b))
        return r, g, b

    def cubic(self, A, B, Cc, D, t):
        s = A.shape[0]
        out = np.zeros((s + 2, self.k, self.n), dtype=np.uint64)
        A, B, Cc, D, t = (np.ascontiguousarray(x) for x in (A, B, Cc, D, t))
        self.L.fo_cubic(self.h, _p(A), _p(B), _p(Cc), _p(D), s, _p(t), _p(out))
        return out

    def linear(self, A, B, t):
        s = A.shape[0]
        out = np.zeros((s + 1, self.k, self.n), dtype=np.uint64)
        A, B, t = (np.ascontiguousarray(x) for x in (A, B, t))
        self.L.fo_linear(self.h, _p(A), _p(B), s, _p(t), _p(out))
        return out


def chacha20_block(key, counter, nonce):
    """64 bytes of the ChaCha20 stream (64-bit block counter, 64-bit nonce) -- the oracle's restatement"""
    out = np.zeros(64, dtype=np.uint8)
    lib().fo_chacha20_block(bytes(key), int(counter), int(nonce), _pb(out))
    return out.tobytes()


def noise_cdt():
    out = np.zeros(19, dtype=np.uint64)
    lib().fo_noise_cdt(_p(out))
    return [int(v) for v in out]


def digest(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return int(lib().fo_digest(_p(a), a.size))


# ---------------------------------------------------------------------------------------------
# decode circuits (homo/fhe_decode.h), composed from the C oracle's single operations.
# Written as explicit operation lists, independently of the product's circuits.py.
# ---------------------------------------------------------------------------------------------
import math as _math


def _taylor_terms(sign):
    # (number of squarings, number of extra multiplies by shifted_x, coefficient)
    # homo/fhe_decode.h:66-98 (sin) / :146-178 (cos); sin coefficients are -1 * cos coefficients
    return [(1, 0, sign * 0.5), (2, 0, sign * -1.0 / 24.0), (2, 2, sign * 1.0 / 720.0),
            (3, 0, sign * -1.0 / 40320.0), (3, 2, sign * 1.0 / 3628800.0)]


def _oracle_taylor(orc, x, zero, sign, constant):
    shifted = orc.add_plain(x, orc.encode(-3 * _math.pi / 2.0))      # :57 / :137
    res = orc.add_plain(zero, orc.encode(constant))                   # :113 / :193
    for squarings, mults, coeff in _taylor_terms(sign):
        p = shifted
        for _ in range(squarings):
            p = orc.square(p)
        for _ in range(mults):
            p = orc.multiply(p, shifted)
        res = orc.add(res, orc.multiply_plain(p, orc.encode(coeff)))  # :114-118
    return res


def oracle_homomorphic_sin(orc, x, zero):
    return _oracle_taylor(orc, x, zero, +1.0, -1.0)


def oracle_homomorphic_cos(orc, x, zero):
    return _oracle_taylor(orc, x, zero, -1.0, 1.0)


def oracle_approximated_step(orc, amplitude, index, count, order, degree, delta, width, height, zeros, positions=None):
    """homo/fhe_decode.h:202-242 (homomorphic overload), including the offset mutation at :229.
    positions=(p0, p1): only those output positions are evaluated (the checker for a shard of the position loop); the
    positions before p0 still advance `offset` exactly as the reference's loop does."""
    p0, p1 = positions if positions is not None else (0, width * height)
    b = orc.multiply_plain(count, orc.encode(0.5))                    # :214-215
    offset = orc.add(index, b)                                        # :216-217
    offset = orc.add_plain(offset, orc.encode(-0.5))                  # :218
    offset = orc.negate(offset)                                       # :219
    b = orc.add_plain(b, orc.encode(delta - 0.5))                     # :220
    run = []
    for i in range(p1):
        if i < p0:
            for j in range(1, degree + 1):
                offset = orc.add_plain(offset, orc.encode(float(i)))  # :229, the only effect of position i on later positions
            continue
        c = orc.multiply_plain(b, orc.encode(1.0 / float(order)))     # :222-223
        for j in range(1, degree + 1):
            arg_factor = float(np.float32(j)) * _math.pi / float(order)   # :225
            sin_arg = orc.multiply_plain(b, orc.encode(arg_factor))   # :226-227
            cos_arg = offset.copy()                                   # :228
            offset = orc.add_plain(offset, orc.encode(float(i)))      # :229
            cos_arg = orc.multiply_plain(cos_arg, orc.encode(arg_factor))   # :230
            s = oracle_homomorphic_sin(orc, sin_arg, zeros(i, j, "sin"))
            co = oracle_homomorphic_cos(orc, cos_arg, zeros(i, j, "cos"))
            term = orc.multiply(s, co)                                # :234-235
            term = orc.multiply_plain(term, orc.encode(2.0 / (_math.pi * float(np.float32(j)))))   # :236
            c = orc.add(c, term)                                      # :237
        run.append(orc.multiply(c, amplitude))                       # :239-240
    return run


# ---------------------------------------------------------------------------------------------
# The RELINEARISED mode (include/fhe_circuits.h fhe_circuits_create_relin): the reference's own
# Evaluator call sequences with evaluator.relinearize(x, evk) after every multiply / square.  The reference never
# relinearises (homo/fhe_resize.h:174-179, homo/fhe_decode.h:67-98,235,239) although it carries the decomposition bit
# count it would need (homo/client_resize.cpp:26,47,72; DBC = 30, homo/fhe_image.h:28): these are the checker's
# definitions of that mode, plain compositions of fo_multiply / fo_square + fo_relinearize3, written op by op.
# ---------------------------------------------------------------------------------------------
class RelinOracle:
    """An Oracle whose multiply / square relinearise their result: every ciphertext keeps two polynomials.
    evk: the oracle's own evaluation keys (Oracle.evk_gen), dbc their decomposition bit count."""

    def __init__(self, orc, evk, dbc):
        self.orc, self.evk, self.dbc = orc, evk, int(dbc)

    def __getattr__(self, name):                 # every other operation is the plain oracle's
        return getattr(self.orc, name)

    def _rl(self, p):
        assert p.shape[0] <= 3, "relinearised mode: operands have two polynomials"
        return self.orc.relinearize(p, self.evk, self.dbc) if p.shape[0] == 3 else p

    def multiply(self, a, b):
        return self._rl(self.orc.multiply(a, b))

    def square(self, a):
        return self._rl(self.orc.square(a))


class TailRelinOracle:
    """The second relinearised mode ("per Cubic"): the reference's call sequences UNCHANGED (products grow to 3 / 4 polynomials as
    in homo/fhe_resize.h:174-179,196-199) and ONE evaluator.relinearize(result, evk) at the end of every Cubic / Linear, taking the
    size-4 (Cubic) or size-3 (Linear) result to 2 with the keys for s^2 and s^3 -- two key switches per Cubic where RelinOracle
    spends five.  An oracle-shaped object whose operations are the plain oracle's; `tail(x)` is that one call."""

    def __init__(self, orc, evks, dbc):
        self.orc, self.evks, self.dbc = orc, evks, int(dbc)

    def __getattr__(self, name):
        return getattr(self.orc, name)

    def tail(self, x):
        return self.orc.relinearize_n(x, self.evks, self.dbc) if x.shape[0] > 2 else x


class SampleRelinOracle:
    """The third relinearised mode ("per sample", FHE_RELIN_PER_SAMPLE): SampleBicubic / SampleLinear exactly as the reference evaluates them
    (homo/fhe_resize.h:237-248,293-303: sizes 2 -> 4 -> 6 and 2 -> 3 -> 4, no relinearisation inside) and ONE evaluator.relinearize of every
    output (6 -> 2: keys for s^2 .. s^5; 4 -> 2).  `sample_tail(x)` is that call; a stand-alone Cubic / Linear is relinearised the same way by
    its caller (`sample_tail(oracle_cubic_calls(orc.orc, ...))`)."""

    def __init__(self, orc, evks, dbc):
        self.orc, self.evks, self.dbc = orc, evks, int(dbc)

    def __getattr__(self, name):
        return getattr(self.orc, name)

    def sample_tail(self, x):
        return self.orc.relinearize_n(x, self.evks, self.dbc) if x.shape[0] > 2 else x


def _tail(orc, x):
    return orc.tail(x) if hasattr(orc, "tail") else x


def _sample_tail(orc, x):
    return orc.sample_tail(x) if hasattr(orc, "sample_tail") else x


def oracle_cubic_calls(orc, A, B, Cc, D, t):
    """Cubic (homo/fhe_resize.h:143-189) one Evaluator call per line -- fo_cubic restates the same sequence in C; this form
    takes any oracle-shaped object, so oracle_cubic_calls(RelinOracle(...), ...) is the relinearised Cubic."""
    E = orc.encode
    a = orc.multiply_plain(B, E(3))                     # :150
    a = orc.sub(a, A)                                   # :151
    a = orc.sub(a, orc.multiply_plain(Cc, E(3)))        # :152-153
    a = orc.add(a, D)                                   # :154-155
    b = orc.multiply_plain(A, E(2))                     # :158
    b = orc.sub(b, orc.multiply_plain(B, E(5)))         # :159-160
    b = orc.add(b, orc.multiply_plain(Cc, E(4)))        # :161-162
    b = orc.sub(b, D)                                   # :163-164
    c = orc.sub(Cc, A)                                  # :167-169
    t2 = orc.square(t)                                  # :174
    t3 = orc.multiply(t, t)                             # :175 (t3 IS t * t)
    a = orc.multiply(a, t3)                             # :177
    b = orc.multiply(b, t2)                             # :178
    c = orc.multiply(c, t)                              # :179
    a = orc.add(a, b)                                   # :181
    a = orc.add(a, c)                                   # :182
    a = orc.multiply_plain(a, E(0.5))                   # :183
    return _tail(orc, orc.add(a, B))                    # :184 (d = B); TailRelinOracle: + one evaluator.relinearize(result, evk)


def oracle_linear_calls(orc, A, B, t):
    """Linear (homo/fhe_resize.h:191-204) one Evaluator call per line."""
    omt = orc.add_plain(orc.negate(t), orc.encode(1.0))     # :196
    x = orc.multiply(omt, A)                                # :197
    y = orc.multiply(B, t)                                  # :198
    return _tail(orc, orc.add(x, y))                        # :199


def oracle_sample_bicubic_calls(orc, p, xfract, yfract):
    """SampleBicubic (homo/fhe_resize.h:293-303): p = the sixteen clamped taps, row-major 4 x 4"""
    cols = [oracle_cubic_calls(orc, p[4 * r], p[4 * r + 1], p[4 * r + 2], p[4 * r + 3], xfract) for r in range(4)]
    return _sample_tail(orc, oracle_cubic_calls(orc, cols[0], cols[1], cols[2], cols[3], yfract))      # SampleRelinOracle: + one relinearize of the pixel


def oracle_sample_linear_calls(orc, p, xfract, yfract):
    """SampleLinear (homo/fhe_resize.h:237-248): p = p00, p10, p01, p11"""
    c0 = oracle_linear_calls(orc, p[0], p[1], xfract)
    c1 = oracle_linear_calls(orc, p[2], p[3], xfract)
    return _sample_tail(orc, oracle_linear_calls(orc, c0, c1, yfract))
