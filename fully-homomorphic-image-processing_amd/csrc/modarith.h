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

// Multiplier rates measured on gfx950 (tools/ubench.hip): v_mad_u64_u32 52 lanes/clk/CU, v_mul_hi_u32 33,
// v_mul_lo_u32 29.  hipcc narrows every 32 x 32 product of which only the low word is demanded to the slow
// v_mul_lo_u32 (four of the ten multiplies of the product above).  mul_shoup_lazy4 keeps them on v_mad_u64_u32:
//  * the subtraction of qhat * q is an addition of qhat * (2^64 - q), so the four cross terms whose low words make
//    up the high word of the result form ONE multiply-add chain;
//  * that chain's high word is kept demanded by and-ing it with a zero the compiler cannot see through
//    (fhe_opaque_zero: a __device__ word that nobody writes; one scalar load per kernel);
//  * the high word of x * wp leaves out the low x low partial product and the carries of the cross terms: qhat in
//    [exact - 2, exact], so the result is x * w mod q + {0, 1, 2} q, i.e. in [0, 4q) for ANY 64-bit x.
// 2 v_mul_hi_u32 + 7 v_mad_u64_u32: 4.5 against 3.5 products/clk/CU, 4.2 against 3.3 as a butterfly without the
// conditional subtraction the wider range makes unnecessary (tools/ubench4.hip, profiles/r02_ubench4_shoup_products.txt;
// the exact product written the same way measured no faster inside the transforms).
inline __device__ u32 fhe_opaque_zero;
// nq = 2^64 - q; `zero` = fhe_opaque_zero, read once per kernel
__device__ __forceinline__ u64 mul_shoup_lazy4(u64 x, u64 w, u64 wp, u64 nq, u32 zero) {
    const u32 xl = (u32)x, xh = (u32)(x >> 32), wl = (u32)w, wh = (u32)(w >> 32), pl = (u32)wp, ph = (u32)(wp >> 32);
    u32 cy;
    const u32 s = __builtin_addc(__umulhi(xh, pl), __umulhi(xl, ph), 0u, &cy);
    const u64 A = (u64)xh * ph + (((u64)cy << 32) | s);
    const u32 al = (u32)A, ah = (u32)(A >> 32), nl = (u32)nq, nh = (u32)(nq >> 32);
    const u64 P = (u64)al * nl + (u64)xl * wl;
    const u64 C = (u64)ah * nl + ((u64)al * nh + ((u64)xh * wl + (u64)xl * wh));
    const u32 hi = (u32)(P >> 32) + (u32)C + ((u32)(C >> 32) & zero);      // one v_add3_u32; no 64-bit add of a shifted word
    return ((u64)hi << 32) | (u32)P;
}
// acc + x * w mod q (same product; the accumulator enters the multiply-add chain as its first addend, and the high
// word is finished with one three-operand add instead of a 64-bit add of a shifted word): a Harvey butterfly gets
// X + T from the product itself and X - T + 4q as 2X + 4q - (X + T).  All arithmetic is mod 2^64, so the values are
// the ones mul_shoup_lazy4 followed by a 64-bit addition gives.
__device__ __forceinline__ u64 mul_shoup_lazy4_acc(u64 x, u64 w, u64 wp, u64 nq, u32 zero, u64 acc) {
    const u32 xl = (u32)x, xh = (u32)(x >> 32), wl = (u32)w, wh = (u32)(w >> 32), pl = (u32)wp, ph = (u32)(wp >> 32);
    u32 cy;
    const u32 s = __builtin_addc(__umulhi(xh, pl), __umulhi(xl, ph), 0u, &cy);
    const u64 A = (u64)xh * ph + (((u64)cy << 32) | s);
    const u32 al = (u32)A, ah = (u32)(A >> 32), nl = (u32)nq, nh = (u32)(nq >> 32);
    const u64 P = (u64)al * nl + ((u64)xl * wl + acc);
    const u64 C = (u64)ah * nl + ((u64)al * nh + ((u64)xh * wl + (u64)xl * wh));
    const u32 hi = (u32)(P >> 32) + (u32)C + ((u32)(C >> 32) & zero);
    return ((u64)hi << 32) | (u32)P;
}
// any 64-bit v -> v mod q + {0, 1, 2} q, below 4q  (product with 1; one_p = floor(2^64 / q))
__device__ __forceinline__ u64 reduce_lazy4(u64 v, u64 one_p, u64 nq, u32 zero) { return mul_shoup_lazy4(v, 1, one_p, nq, zero); }

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

// the same without the two conditional subtractions: a * b mod q + {0, 1, 2} q, in [0, 3q)
__device__ __forceinline__ u64 mul_barrett_lazy3(u64 a, u64 b, const Modulus &m) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u64 x = (hi << (64 - m.s1)) | (lo >> m.s1);
    u64 plo = x * m.mu, phi = __umul64hi(x, m.mu);
    u64 qhat = (phi << (64 - m.s2)) | (plo >> m.s2);
    return lo - qhat * m.q;
}

// a * b mod q + {0..4} q for a, b in [0, q), q <= 58 bits, on the multiply-add chain of mul_shoup_lazy4: with
// mu' = mu << (63 - b) the Barrett quotient floor(x mu / 2^(b+1)) IS the high word of x * mu', which is taken with the
// same approximation (exact - 2 ... exact), and z - qhat q becomes lo(z) + qhat (2^64 - q) with lo(z) as the chain's
// first addend.  qhat in [Q - 4, Q]: result in [0, 5q).  8 v_mad_u64_u32 + 2 v_mul_hi_u32 against 9 + 6 v_mul_lo_u32.
struct BarrettLazy { u64 mup, nq; u32 s1, zero; };      // mup = mu << (63 - b), nq = 2^64 - q, s1 = b - 1
__device__ __forceinline__ BarrettLazy barrett_lazy(const Modulus &m) {
    BarrettLazy o; o.mup = m.mu << (62 - m.s1); o.nq = 0 - m.q; o.s1 = m.s1; o.zero = fhe_opaque_zero; return o;
}
__device__ __forceinline__ u64 mul_barrett_lazy5(u64 a, u64 b, const BarrettLazy &k) {
    const u64 lo = a * b, hi = __umul64hi(a, b);
    const u64 x = (hi << (64 - k.s1)) | (lo >> k.s1);
    const u32 xl = (u32)x, xh = (u32)(x >> 32), pl = (u32)k.mup, ph = (u32)(k.mup >> 32);
    u32 cy;
    const u32 s = __builtin_addc(__umulhi(xh, pl), __umulhi(xl, ph), 0u, &cy);
    const u64 A = (u64)xh * ph + (((u64)cy << 32) | s);
    const u32 al = (u32)A, ah = (u32)(A >> 32), nl = (u32)k.nq, nh = (u32)(k.nq >> 32);
    const u64 P = (u64)al * nl + lo;
    const u64 C = (u64)ah * nl + (u64)al * nh;
    const u32 h = (u32)(P >> 32) + (u32)C + ((u32)(C >> 32) & k.zero);
    return ((u64)h << 32) | (u32)P;
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
