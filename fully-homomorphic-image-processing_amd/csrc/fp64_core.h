// fp64_core.h -- exact FP64-FMA modular arithmetic and the LL&M half-line circuit, shared by the
// fused DCT kernels (dct_fused.hip).
#pragma once
#include "internal.h"

#pragma clang fp contract(off)

namespace fp64 {

__device__ __forceinline__ double mm(double y, double w, double p, double pinv) {
    const double h = y * w;
    const double l = __builtin_fma(y, w, -h);
    const double q = __builtin_rint(h * pinv);
    return __builtin_fma(-q, p, h) + l;
}
__device__ __forceinline__ double red(double x, double p, double pinv) {
    return __builtin_fma(-__builtin_rint(x * pinv), p, x);
}
// K independent products written stage by stage: with only two waves per SIMD the FP64 pipe needs
// instruction-level parallelism, and hipcc otherwise emits each five-op chain back to back.
template <int K>
__device__ __forceinline__ void mmv(double (&y)[K], const double (&w)[K], double p, double pinv) {
    double h[K], l[K], q[K];
#pragma unroll
    for (int i = 0; i < K; i++) h[i] = y[i] * w[i];
#pragma unroll
    for (int i = 0; i < K; i++) q[i] = h[i] * pinv;
#pragma unroll
    for (int i = 0; i < K; i++) l[i] = __builtin_fma(y[i], w[i], -h[i]);
#pragma unroll
    for (int i = 0; i < K; i++) q[i] = __builtin_rint(q[i]);
#pragma unroll
    for (int i = 0; i < K; i++) h[i] = __builtin_fma(-q[i], p, h[i]);
#pragma unroll
    for (int i = 0; i < K; i++) y[i] = h[i] + l[i];
}

// exact integer <-> double moves: for 0 <= v < 2^52, bits(2^52 + v) = 0x4330000000000000 | v
__device__ __forceinline__ double u52_to_f64(u64 v) { return __longlong_as_double((long long)(v | 0x4330000000000000ULL)) - 4503599627370496.0; }
__device__ __forceinline__ u64 f64_to_u52(double v) { return (u64)__double_as_longlong(v + 4503599627370496.0) & 0x000FFFFFFFFFFFFFULL; }

// Even / odd half of one LL&M line (homo/fhe_image.h:215-242) on a single NTT slot.
// In:  even: x[m] = d_m + d_(7-m) (tmp0..tmp3);  odd: x[m] = d_m - d_(7-m) (tmp7,tmp6,tmp5,tmp4).
// Out: x[m] = line output 2m + HALF.  c[] = constants 0..2 (even) or 3..11 (odd) at this slot.
template <int HALF>
__device__ __forceinline__ void line_half(double &x0, double &x1, double &x2, double &x3, const double (&c)[9], double p, double pinv) {
    if constexpr (HALF == 0) {
        const double tmp10 = x0 + x3, tmp13 = x0 - x3, tmp11 = x1 + x2, tmp12 = x1 - x2;
        double y[3] = {tmp12 + tmp13, tmp13, tmp12};
        const double w[3] = {c[0], c[1], c[2]};
        mmv<3>(y, w, p, pinv);
        x0 = tmp10 + tmp11;        // out 0
        x2 = tmp10 - tmp11;        // out 4
        x1 = y[0] + y[1];          // out 2
        x3 = y[0] + y[2];          // out 6
    } else {
        const double tmp7 = x0, tmp6 = x1, tmp5 = x2, tmp4 = x3;
        const double z1 = tmp4 + tmp7, z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
        double ya[5] = {z3 + z4, tmp4, tmp5, tmp6, tmp7}, yb[4] = {z1, z2, z3, z4};
        const double wa[5] = {c[0], c[1], c[2], c[3], c[4]}, wb[4] = {c[5], c[6], c[7], c[8]};
        mmv<5>(ya, wa, p, pinv);
        mmv<4>(yb, wb, p, pinv);
        const double y[9] = {ya[0], ya[1], ya[2], ya[3], ya[4], yb[0], yb[1], yb[2], yb[3]};
        const double z3b = y[7] + y[0], z4b = y[8] + y[0];
        x0 = y[4] + y[5] + z4b;    // out 1 = tmp7' + z1' + z4
        x1 = y[3] + y[6] + z3b;    // out 3 = tmp6' + z2' + z3
        x2 = y[2] + y[6] + z4b;    // out 5 = tmp5' + z2' + z4
        x3 = y[1] + y[5] + z3b;    // out 7 = tmp4' + z1' + z3
    }
}
template <int HALF> struct HalfC { static constexpr int NC = HALF ? 9 : 3, FIRST = HALF ? 3 : 0; };

struct Work { u32 blk, line, poly, prime, half; };
// blockIdx -> work item.  The two halves of an item sit 8 apart so they land on the same XCD, and the
// prime is the SLOWEST index: workgroups resident at the same time share one prime, so its twiddles
// and circuit constants (0.4 - 2.5 MB per prime) stay in the XCD L2s instead of thrashing them.
__device__ __forceinline__ Work decode(u32 idx, u32 k) {
    const u32 w = ((idx >> 4) << 3) | (idx & 7);
    const u32 per_prime = (gridDim.x >> 1) / k;
    Work o;
    o.half = (idx >> 3) & 1;
    o.prime = w / per_prime;
    u32 t = w - o.prime * per_prime;
    o.poly = t & 1;
    t >>= 1;
    o.line = t & 7;
    o.blk = t >> 3;
    return o;
}


}  // namespace fp64
