// modarith.h -- 64-bit modular arithmetic device primitives for gfx950.
//
// CDNA4 has no 64x64 multiplier: every __umul64hi / 64-bit multiply below lowers to
// v_mad_u64_u32 / v_mul_hi_u32 chains.  Twiddles and cached plaintext constants therefore carry a
// precomputed Shoup companion w' = floor(w * 2^64 / q) so that a modular multiplication by a
// known constant costs one high product and two low products; products of two unknown operands
// (BEHZ tensor step) use a two-multiply Barrett reduction with mu = floor(2^(2b) / q).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

struct Modulus {
    u64 q;      // prime < 2^61
    u64 mu;     // floor(2^(2b) / q), b = bit length of q  (b <= 61, so mu < 2^62)
    u32 s1;     // b - 1
    u32 s2;     // b + 1
};

__device__ __forceinline__ u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }
__device__ __forceinline__ u64 addmod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
__device__ __forceinline__ u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
__device__ __forceinline__ u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// x * w mod q for any x < 2^64, result in [0, 2q)  (Shoup / Harvey lazy product)
__device__ __forceinline__ u64 mul_shoup_lazy(u64 x, u64 w, u64 wp, u64 q) {
    return x * w - __umul64hi(x, wp) * q;
}
// result in [0, q)
__device__ __forceinline__ u64 mul_shoup(u64 x, u64 w, u64 wp, u64 q) {
    return csub(mul_shoup_lazy(x, w, wp, q), q);
}

// a * b mod q for a, b in [0, q); Barrett with two multiplications, result in [0, q).
// x = floor(z / 2^(b-1)) < 2^(b+1); qhat = floor(x mu / 2^(b+1)) in [Q-2, Q]  =>  z - qhat q in [0, 3q).
__device__ __forceinline__ u64 mul_barrett(u64 a, u64 b, const Modulus &m) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u64 x = (hi << (64 - m.s1)) | (lo >> m.s1);
    u64 plo = x * m.mu, phi = __umul64hi(x, m.mu);
    u64 qhat = (phi << (64 - m.s2)) | (plo >> m.s2);
    u64 r = lo - qhat * m.q;
    r = csub(r, 2 * m.q);
    return csub(r, m.q);
}

// floor(w * 2^64 / q) for w < q, by restoring division (setup kernels only)
__device__ inline u64 shoup_companion(u64 w, u64 q) {
    u64 rem = w, quo = 0;
    for (int i = 0; i < 64; i++) {
        quo <<= 1;
        u64 top = rem >> 63;
        rem <<= 1;
        if (top || rem >= q) { rem -= q; quo |= 1; }
    }
    return quo;
}

__device__ __forceinline__ u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
