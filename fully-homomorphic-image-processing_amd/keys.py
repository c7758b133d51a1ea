"""Client-side objects over the C ABI: KeyGenerator, Encryptor, Decryptor (seal:: names).

The reference's clients (homo/client_jpeg.cpp:98-165,266-280) generate keys, encrypt every pixel and
decrypt the results on the CPU through SEAL.  Here the ring arithmetic of those steps (NTT, dyadic
products, additions) runs on the GPU through the same C ABI as the server path; sampling and the
final exact rounding t*x/q (big integers) are host work.  Textbook BFV as in SURVEY.md App. A.7:
  sk s <- ternary;  pk = (-(a s + e), a);  Enc(m) = (Delta m' + pk0 u + e1, pk1 u + e2)
  Dec(c) = round(t/q [sum_j c_j s^j]_q) mod t;   evk[i][d] = (-(a s + e) + 2^(dbc d) s^2 E_i, a)
"""
import ctypes as C
import os
from functools import reduce

import numpy as np
import torch

from . import _lib
from .evaluator import _ptr, _stream, to_device, to_host


class _OsRandom:
    """numpy-Generator-shaped sampling backed by the operating system's CSPRNG (os.urandom):
    rejection sampling for the integers (no modulo bias), Box-Muller for the clipped normal."""

    @staticmethod
    def _words(count):
        return np.frombuffer(os.urandom(8 * count), dtype=np.uint64)

    def integers(self, low, high, size, dtype=np.int64):
        bound = int(high) - int(low)
        limit = ((1 << 64) // bound) * bound
        out = np.empty(size, dtype=np.uint64)
        filled = 0
        while filled < size:
            w = self._words(size - filled + 16)
            if limit < (1 << 64):
                w = w[w < np.uint64(limit)]
            take = min(len(w), size - filled)
            out[filled:filled + take] = w[:take] % np.uint64(bound)
            filled += take
        return (out + np.uint64(low)).astype(dtype)

    def normal(self, mean, sigma, size):
        u1 = ((self._words(size) >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740993.0
        u2 = ((self._words(size) >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740993.0
        return mean + sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


class _Sampler:
    """seed=None (the default): every draw comes from the OS CSPRNG.  An explicit seed selects a
    reproducible numpy PCG64 stream -- tests only; never pass a seed for keys that protect data."""

    def __init__(self, ctx, seed=None):
        self.ctx = ctx
        self.rng = _OsRandom() if seed is None else np.random.default_rng(seed)

    def ternary(self):
        v = self.rng.integers(0, 3, size=self.ctx.n)
        return np.stack([np.where(v == 2, q - 1, v).astype(np.uint64) for q in self.ctx.q])

    def noise(self):
        n = self.ctx.n
        e = np.rint(self.rng.normal(0.0, 3.19, size=n))
        bad = np.abs(e) > 19
        while bad.any():                                   # clipped at 6 sigma
            e[bad] = np.rint(self.rng.normal(0.0, 3.19, size=int(bad.sum())))
            bad = np.abs(e) > 19
        e = e.astype(np.int64)
        return np.stack([np.where(e < 0, q + e, e).astype(np.uint64) for q in self.ctx.q])

    def uniform(self):
        return np.stack([self.rng.integers(0, q, size=self.ctx.n, dtype=np.uint64) for q in self.ctx.q])


def _check_key(ctx, key, polys, what):
    """a key of another context (other degree / number of primes) would be read with THIS context's strides: refuse it here"""
    if tuple(key.shape) != ((polys, ctx.k, ctx.n) if polys > 1 else (ctx.k, ctx.n)) or key.dtype != torch.int64 or key.device != ctx.device:
        raise ValueError("%s: expected an int64 tensor of shape %r on %s, got %r (%s, %s) -- a key of another context?"
                         % (what, (polys, ctx.k, ctx.n) if polys > 1 else (ctx.k, ctx.n), ctx.device, tuple(key.shape), key.dtype, key.device))


def _ntt(ctx, a):
    out = torch.empty_like(a)
    polys = a.numel() // (ctx.k * ctx.n)
    _lib.call("fhe_ntt_forward", ctx.h, _ptr(a), _ptr(out), polys, _stream())
    return out


def _ring_mul(ctx, a, b_ntt):
    """a: [polys, k, n] coefficient form (device), b_ntt: [k, n] NTT form -> a*b in R_q, coefficient form."""
    fa = _ntt(ctx, a)
    for j in range(a.shape[0]):
        _lib.call("fhe_dyadic_multiply", ctx.h, _ptr(fa[j]), _ptr(b_ntt), _ptr(fa[j]), 1, _stream())
    _lib.call("fhe_ntt_inverse", ctx.h, _ptr(fa), _ptr(fa), a.shape[0], _stream())
    return fa


class KeyGenerator:
    def __init__(self, ctx, seed=None):
        self.ctx = ctx
        smp = _Sampler(ctx, seed)
        self._sk = to_device(smp.ternary()[None], ctx.device)               # [1, k, n]
        self._sk_ntt = _ntt(ctx, self._sk)[0]
        a = to_device(smp.uniform()[None], ctx.device)
        e = to_device(smp.noise()[None], ctx.device)
        as_e = _ring_mul(ctx, a, self._sk_ntt)
        _lib.call("fhe_add", ctx.h, _ptr(as_e), _ptr(e), _ptr(as_e), 1, _stream())
        _lib.call("fhe_negate", ctx.h, _ptr(as_e), _ptr(as_e), 1, _stream())
        self._pk = torch.cat([as_e, a], dim=0).contiguous()                  # [2, k, n]
        self._smp = smp

    def secret_key(self):
        return self._sk[0]

    def public_key(self):
        return self._pk

    def generate_evaluation_keys(self, dbc, count=1):
        """Evaluation keys in NTT form (library slot order) for Evaluator.relinearize: count == 1 (SEAL 2.3
        generate_evaluation_keys(dbc, keys)): [k][digits][2][k][n], the keys for s^2; count > 1 (generate_evaluation_keys(dbc, count,
        keys)): [count][k][digits][2][k][n], the keys for s^2 .. s^(count+1) -- what relinearising a ciphertext of count + 2
        polynomials takes (Circuits(relin=(keys, dbc, "cubic")): count = 2)."""
        ctx = self.ctx
        nd = int(_lib.load().fhe_evk_digits(ctx.h, dbc))
        out = []
        power = self._sk                                                        # s^j, coefficient form [1, k, n]
        for _ in range(count):
            power = _ring_mul(ctx, power, self._sk_ntt)
            sp = to_host(power)[0]
            evk = np.zeros((ctx.k, nd, 2, ctx.k, ctx.n), dtype=np.uint64)
            for i in range(ctx.k):
                for d in range(nd):
                    a = self._smp.uniform()
                    e = to_device(self._smp.noise()[None], ctx.device)
                    k0 = _ring_mul(ctx, to_device(a[None], ctx.device), self._sk_ntt)
                    _lib.call("fhe_add", ctx.h, _ptr(k0), _ptr(e), _ptr(k0), 1, _stream())
                    _lib.call("fhe_negate", ctx.h, _ptr(k0), _ptr(k0), 1, _stream())
                    h0 = to_host(k0)[0].copy()
                    qi = ctx.q[i]
                    wd = pow(2, dbc * d, qi)
                    h0[i] = np.array([(int(x) + int(s) * wd) % qi for x, s in zip(h0[i], sp[i])], dtype=np.uint64)
                    evk[i, d, 0], evk[i, d, 1] = h0, a
            out.append(_ntt(ctx, to_device(evk, ctx.device)))
        return out[0] if count == 1 else torch.stack(out).contiguous()


class Encryptor:
    def __init__(self, ctx, public_key, seed=None):
        self.ctx = ctx
        _check_key(ctx, public_key, 2, "Encryptor: public key")
        self._pk_ntt = _ntt(ctx, public_key.contiguous())
        self._smp = _Sampler(ctx, seed)

    def encrypt(self, plain):
        ctx = self.ctx
        u = _ntt(ctx, to_device(self._smp.ternary()[None], ctx.device))[0]
        ct = torch.empty((2, ctx.k, ctx.n), dtype=torch.int64, device=ctx.device)
        for j in range(2):
            _lib.call("fhe_dyadic_multiply", ctx.h, _ptr(self._pk_ntt[j]), _ptr(u), _ptr(ct[j]), 1, _stream())
        _lib.call("fhe_ntt_inverse", ctx.h, _ptr(ct), _ptr(ct), 2, _stream())
        e = to_device(np.stack([self._smp.noise(), self._smp.noise()]), ctx.device)
        _lib.call("fhe_add", ctx.h, _ptr(ct), _ptr(e), _ptr(ct), 2, _stream())
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        nz = np.flatnonzero(p)
        ln = int(nz[-1]) + 1 if nz.size else 0     # significant coefficient count
        if ln:
            _lib.call("fhe_add_plain", ctx.h, _ptr(ct), 2 * ctx.k * ctx.n, 1, p.ctypes.data_as(C.c_void_p), ln, 1, _stream())
        return ct


class DeviceEncryptor:
    """Batches of fresh encryptions formed on the device (include/fhe_hip.h fhe_encrypt_batch; csrc/encrypt.hip) -- what a
    SERVER needs: the reference's loops encrypt two fractions per output pixel (homo/fhe_resize.h:230,234,262,266) and an
    encode(0) per homomorphic_sin / cos, accumulator and index (homo/fhe_decode.h:54,134; homo/server_decode.cpp:121,126).
    u, e1, e2 come from the ChaCha20 stream of (key, index): the key is 32 bytes from the operating system's generator unless
    `key` is given (a key must never meet the same index twice), `index` counts the encryptions made under it and only ever
    counts UPWARDS -- unless the encryptor was made with reproducible=True (tests, seeded benchmark runs; needs a key): then
    `seek(i)` sets the number of the next one, so a shard of a job that starts at encryption i of the reference's sequence
    produces the same ciphertexts as the whole job (the role _IndexedEncryptions plays for the host sampler).  A caller who
    passes a real key without the flag gets a stream that cannot be rewound: seek() refuses, and server_resize /
    server_decode (which position any encryptor that exposes `seek`) never see one."""

    def __init__(self, ctx, public_key, key=None, int_coeffs=None, frac_coeffs=None, reproducible=False):
        self.ctx = ctx
        _check_key(ctx, public_key, 2, "DeviceEncryptor: public key")
        self._pk_ntt = _ntt(ctx, public_key.contiguous())
        if reproducible and key is None:
            raise ValueError("reproducible=True needs an explicit key (the OS generator's key is never reused)")
        self.reproducible = bool(reproducible)
        self.key = os.urandom(32) if key is None else bytes(key)
        if len(self.key) != 32:
            raise ValueError("the sampler key has 32 bytes")
        self.next = 0
        self.int_coeffs = 100 if int_coeffs is None else int(int_coeffs)          # seal::FractionalEncoder(t, poly_modulus, 100, 100, 2), homo/server_resize.cpp
        self.frac_coeffs = 100 if frac_coeffs is None else int(frac_coeffs)
        self._scratch = None

    def seek(self, i):
        if not self.reproducible:
            raise RuntimeError("seek could repeat a (key, index) pair -- identical u, e1, e2 under two plaintexts; only an encryptor made with "
                               "reproducible=True (tests, seeded runs) may position its stream")
        self.next = int(i)

    def _run(self, plain, count):
        ctx = self.ctx
        out = torch.empty((count, 2, ctx.k, ctx.n), dtype=torch.int64, device=ctx.device)
        if count:
            need = int(_lib.load().fhe_encrypt_scratch_bytes(ctx.h, count))
            if self._scratch is None or self._scratch.numel() * 8 < need:
                self._scratch = torch.empty((need + 7) // 8, dtype=torch.int64, device=ctx.device)
            _lib.call("fhe_encrypt_batch", ctx.h, _ptr(self._pk_ntt), _ptr(plain) if plain is not None else None, count, self.key, self.next,
                      _ptr(out), _ptr(self._scratch), self._scratch.numel() * 8, _stream())
            self.next += count
        return out

    def encrypt_values(self, values):
        """[len(values), 2, k, n]: Enc(encode(v)) for every v (FractionalEncoder::encode on the device, bit for bit the host encoder)"""
        vals = np.ascontiguousarray(values, dtype=np.float64)
        plain = torch.empty((len(vals), self.ctx.n), dtype=torch.int64, device=self.ctx.device)
        if len(vals):
            _lib.call("fhe_frac_encode_batch", self.ctx.h, vals.ctypes.data_as(C.c_void_p), len(vals), self.int_coeffs, self.frac_coeffs, _ptr(plain), _stream())
        return self._run(plain, len(vals))

    def encrypt_plains(self, plains):
        """plains: [count, n] device tensor of coefficients below t"""
        return self._run(plains.contiguous(), int(plains.shape[0]))

    def encrypt_zeros(self, count):
        return self._run(None, int(count))

    def draws(self, first, count):
        """[count, 3, n] int8 on the host: (u, e1, e2) of encryptions first .. first + count - 1 (for checks)"""
        d = torch.empty((count, 3, self.ctx.n), dtype=torch.int8, device=self.ctx.device)
        _lib.call("fhe_encrypt_draws", self.ctx.h, self.key, int(first), int(count), _ptr(d), _stream())
        return d.cpu().numpy()


class Decryptor:
    def __init__(self, ctx, secret_key):
        self.ctx = ctx
        _check_key(ctx, secret_key, 1, "Decryptor: secret key")
        self._sk_ntt = _ntt(ctx, secret_key[None].contiguous())[0]
        self.Q = reduce(lambda a, b: a * b, ctx.q, 1)
        self._crt = [(self.Q // q) * pow(self.Q // q, -1, q) for q in ctx.q]

    def _phase(self, ct):
        ctx = self.ctx
        f = _ntt(ctx, ct.contiguous())
        acc = f[-1].clone()
        for j in range(ct.shape[0] - 2, -1, -1):
            _lib.call("fhe_dyadic_multiply", ctx.h, _ptr(acc), _ptr(self._sk_ntt), _ptr(acc), 1, _stream())
            _lib.call("fhe_add", ctx.h, _ptr(acc), _ptr(f[j]), _ptr(acc), 1, _stream())
        out = acc[None].contiguous()
        _lib.call("fhe_ntt_inverse", ctx.h, _ptr(out), _ptr(out), 1, _stream())
        return to_host(out)[0]

    def decrypt_batch(self, cts, with_budget=False):
        """cts: [count, size, k, n] device tensor -> plaintext coefficients [count, n] (numpy uint64) (and the invariant noise budgets):
        fhe_decrypt_batch -- phase by Horner's rule per NTT slot and the exact rounding floor((t x + floor(q/2)) / q) mod t on the
        device (csrc/encrypt.hip k_dec_round), no big-integer loop on the host"""
        ctx = self.ctx
        cts = cts.contiguous()
        count, size = int(cts.shape[0]), int(cts.shape[1])
        plain = torch.empty((count, ctx.n), dtype=torch.int64, device=ctx.device)
        bits = torch.zeros(max(count, 1), dtype=torch.int32, device=ctx.device)
        if count:
            L = _lib.load()
            need = int(L.fhe_decrypt_scratch_bytes(ctx.h, size, count))
            scratch = torch.empty((need + 7) // 8, dtype=torch.int64, device=ctx.device)
            _lib.call("fhe_decrypt_batch", ctx.h, _ptr(self._sk_ntt), _ptr(cts), size, count, _ptr(plain), _ptr(bits), _ptr(scratch), scratch.numel() * 8, _stream())
        out = plain.cpu().numpy().view(np.uint64)
        if with_budget:
            qbits = int(_lib.load().fhe_ctx_modulus_bits(ctx.h))
            return out, [max(0, qbits - int(b) - 1) for b in bits.cpu().numpy()[:count]]
        return out

    def decrypt(self, ct, with_budget=False):
        """one ciphertext [size, k, n] -> plaintext coefficients [n] (and the invariant noise budget)"""
        if with_budget:
            plain, budget = self.decrypt_batch(ct[None], with_budget=True)
            return plain[0], budget[0]
        return self.decrypt_batch(ct[None])[0]

    def decrypt_host(self, ct, with_budget=False):
        """the same by CRT composition and big-integer rounding on the host (rounds 1-4; kept as an independent check of the device rounding)"""
        ph = self._phase(ct)
        Q, t = self.Q, self.ctx.t
        plain = np.zeros(self.ctx.n, dtype=np.uint64)
        worst = 0
        for c in range(self.ctx.n):
            x = sum(int(ph[i, c]) * f for i, f in enumerate(self._crt)) % Q
            m = (t * x + Q // 2) // Q
            worst = max(worst, abs(t * x - m * Q))
            plain[c] = m % t
        if with_budget:
            return plain, max(0, Q.bit_length() - worst.bit_length() - 1)
        return plain

    def invariant_noise_budget(self, ct):
        return self.decrypt(ct, with_budget=True)[1]
