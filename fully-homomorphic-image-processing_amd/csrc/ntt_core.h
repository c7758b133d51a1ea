// ntt_core.h -- register-resident negacyclic NTT for one 2^L-point residue polynomial per
// workgroup (gfx950, wave64).
//
// Shape: N/16 threads, 16 coefficients per thread held in VGPRs.  The L butterfly stages are
// grouped into ceil(L/4) passes of up to four stages; inside a pass every butterfly partner lives
// in the same thread (radix-16 in registers), between passes the polynomial is transposed through
// LDS (one 8-byte ds_write / ds_read per coefficient, padded so that neither side bank-conflicts).
// For L = 12 that is 3 register passes and 2 LDS transposes per direction.
//
// Forward: Cooley-Tukey, natural order in -> bit-reversed order out, psi powers merged into the
// twiddles (Harvey lazy butterflies, values kept in [0, 4q)).  Inverse: Gentleman-Sande, n^-1
// merged into the last stage.  Twiddle tables hold (w, floor(w 2^64 / q)) pairs.
//
// Index algebra.  During pass P the 4 register-index bits stand for coefficient-index bits
// [LO, LO+4) with LO = max(L - 4P - 4, 0):
//     j(tid, r) = ((tid >> LO) << (LO + 4)) | (r << LO) | (tid & (2^LO - 1)).
// Stage sigma pairs indices that differ in bit b = L - 1 - sigma and uses twiddle
// 2^sigma + (j >> (b + 1)).  Pass 0 therefore reads/writes global memory with consecutive lanes on
// consecutive coefficients (coalesced), and the last pass holds 16 consecutive slots per thread.
// NTT-form buffers are stored in "slot order": slot r of thread tid lives at r * (N/16) + tid, so
// those accesses are coalesced too.  (The order is internal; see include/fhe_hip.h.)
#pragma once
#include "modarith.h"

template <int L> struct NttShape {
    static constexpr int N = 1 << L;
    static constexpr int TP = N / 16;         // threads per polynomial
    static constexpr int NP = (L + 3) / 4;    // register passes
    static constexpr int LDS_WORDS = N + N / 16;
};

__host__ __device__ constexpr int pass_lo(int L, int p) { return (L - 4 * p - 4) < 0 ? 0 : (L - 4 * p - 4); }
__host__ __device__ constexpr int pass_stages(int L, int p) { return (L - 4 * p) > 4 ? 4 : (L - 4 * p); }
__host__ __device__ constexpr int imin(int a, int b) { return a < b ? a : b; }

template <int LO> __device__ __forceinline__ int elem_index(int tid, int r) {
    return ((tid >> LO) << (LO + 4)) | (r << LO) | (tid & ((1 << LO) - 1));
}
template <int LO> __device__ __forceinline__ int lds_pad(int j) { return j + ((j >> (LO + 4)) << LO); }

struct RnsBase {              // device pointers, passed to kernels by value
    const ulonglong2 *tw;     // [count][n]  psi^bitrev(i) with Shoup companion
    const ulonglong2 *itw;    // [count][n]  psi^-bitrev(i); entry 0 = n^-1, entry 1 pre-multiplied by n^-1
    const Modulus *mod;       // [count]
    u32 count;
};

template <int L, int P>
__device__ __forceinline__ void ntt_fwd_pass(u64 (&x)[16], const ulonglong2 *__restrict__ tw, u64 q, int tid) {
    constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    const u64 twoq = 2 * q;
    const int th = (P == 0) ? 0 : (tid >> LO);
#pragma unroll
    for (int u = 0; u < S; u++) {
        const int sigma = 4 * P + u, b = L - 1 - sigma, rb = b - LO;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const ulonglong2 w = tw[(1 << sigma) + ((th << (3 - rb)) | (r0 >> (rb + 1)))];
            const u64 X = csub(x[r0], twoq);
            const u64 T = mul_shoup_lazy(x[r1], w.x, w.y, q);
            x[r0] = X + T;
            x[r1] = X - T + twoq;
        }
    }
}

template <int L, int P>
__device__ __forceinline__ void ntt_inv_pass(u64 (&x)[16], const ulonglong2 *__restrict__ itw, u64 q, int tid) {
    constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    const u64 twoq = 2 * q;
    const int th = (P == 0) ? 0 : (tid >> LO);
#pragma unroll
    for (int u = S - 1; u >= 0; u--) {
        const int sigma = 4 * P + u, b = L - 1 - sigma, rb = b - LO;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const ulonglong2 w = itw[(1 << sigma) + ((th << (3 - rb)) | (r0 >> (rb + 1)))];
            const u64 X = x[r0], Y = x[r1];            // both in [0, 2q)
            const u64 T = csub(X + Y, twoq);
            const u64 D = X - Y + twoq;                // [0, 4q)
            if (sigma == 0) {
                const ulonglong2 ni = itw[0];
                x[r0] = mul_shoup_lazy(T, ni.x, ni.y, q);
            } else {
                x[r0] = T;
            }
            x[r1] = mul_shoup_lazy(D, w.x, w.y, q);    // [0, 2q)
        }
    }
}

// ---- the same passes on mul_shoup_lazy4 (modarith.h: cheaper multiplies, result in [0, 4q)) -------------------------
// Forward: Harvey butterflies with doubled ranges, values in [0, 8q) in and out (8q < 2^64 for every q < 2^61); with
// LAZY (primes <= 58 bits, (2 + 4L) q < 2^64) no conditional subtraction at all, values grow by 4q per stage.
// Inverse: Gentleman-Sande, values in [0, 4q) in and out.
struct NttMod { u64 q, nq, q4; u32 zero; };      // nq = 2^64 - q, q4 = 4q, zero = fhe_opaque_zero
__device__ __forceinline__ NttMod ntt_mod(u64 q) { NttMod o; o.q = q; o.nq = 0 - q; o.q4 = 4 * q; o.zero = fhe_opaque_zero; return o; }
// LAZY is valid when (2 + 4 log2 n) q < 2^64 for n <= 2^14: primes of at most 58 bits
__device__ __forceinline__ bool lazy_ok(const Modulus &m) { return m.s1 + 1 <= 58; }
// floor(2^64 / q) from the Barrett constant mu = floor(2^(2b) / q) (q is odd, so floor((2^64 - 1) / q) is the same)
__device__ __forceinline__ u64 one_companion(const Modulus &m) {
    const u32 b2 = 2 * (m.s1 + 1);
    return b2 >= 64 ? m.mu >> (b2 - 64) : ~0ULL / m.q;
}

// Canonical residue of the LAZY forward transform's outputs (any v < 64q, 2^33 <= q < 2^58).  The quotient is below 64,
// so a single-precision estimate from the high word is enough: c = (2^32 / q)(1 - 2^-17) rounded to float stays below
// 2^32 / q by more than the three float roundings involved (3 x 2^-24), and v.hi 2^32 <= v, so the estimate never
// exceeds v / q; it falls short of it by less than 2^32 / q + 64 x 2^-16.9 < 1.  qhat = trunc(estimate) in {Q - 1, Q}:
// v - qhat q in [0, 2q), one conditional subtraction.  Three full-rate conversions / multiplies and a 32 x 64 product
// instead of a 64-bit Shoup product with 1 and two conditional subtractions (114 -> ~56 issue cycles per coefficient).
__device__ __forceinline__ float canon_scale(u64 q) { return (float)((4294967296.0 / (double)q) * (1.0 - 0x1p-17)); }
__device__ __forceinline__ u64 canon_below_64q(u64 v, u64 q, float c) {
    const u32 qhat = (u32)((float)(u32)(v >> 32) * c);
    return csub(v - (u64)qhat * q, q);
}

template <int L, int P, bool LAZY>
__device__ __forceinline__ void ntt_fwd_pass4(u64 (&x)[16], const ulonglong2 *__restrict__ tw, const NttMod &m, int tid) {
    constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    const int th = (P == 0) ? 0 : (tid >> LO);
#pragma unroll
    for (int u = 0; u < S; u++) {
        const int sigma = 4 * P + u, b = L - 1 - sigma, rb = b - LO;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const ulonglong2 w = tw[(1 << sigma) + ((th << (3 - rb)) | (r0 >> (rb + 1)))];
            const u64 X = LAZY ? x[r0] : csub(x[r0], m.q4);
            const u64 S = mul_shoup_lazy4_acc(x[r1], w.x, w.y, m.nq, m.zero, X);     // X + T
            x[r0] = S;
            x[r1] = (X << 1) + m.q4 - S;                                              // X - T + 4q
        }
    }
}
template <int L, int P>
__device__ __forceinline__ void ntt_inv_pass4(u64 (&x)[16], const ulonglong2 *__restrict__ itw, const NttMod &m, int tid) {
    constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    const int th = (P == 0) ? 0 : (tid >> LO);
#pragma unroll
    for (int u = S - 1; u >= 0; u--) {
        const int sigma = 4 * P + u, b = L - 1 - sigma, rb = b - LO;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const ulonglong2 w = itw[(1 << sigma) + ((th << (3 - rb)) | (r0 >> (rb + 1)))];
            const u64 X = x[r0], Y = x[r1];            // both in [0, 4q)
            const u64 T = csub(X + Y, m.q4);
            const u64 D = X - Y + m.q4;                // [0, 8q)
            if (sigma == 0) {
                const ulonglong2 ni = itw[0];
                x[r0] = mul_shoup_lazy4(T, ni.x, ni.y, m.nq, m.zero);
            } else {
                x[r0] = T;
            }
            x[r1] = mul_shoup_lazy4(D, w.x, w.y, m.nq, m.zero);    // [0, 4q)
        }
    }
}

// Inverse pass for primes of at most 58 bits (32q < 2^64) with STATIC range tracking: which register is the sum side and
// which the product side of a butterfly is known at compile time, so the bound of every register (in units of q) is a
// compile-time value after unrolling.  A sum X + Y is left unreduced while its bound stays <= 16q (its difference
// X - Y + bound(Y) q then stays below 32q, and the product takes any 64-bit operand); a product resets its register to
// [0, 4q).  Entry: every register below EB q (4 for the first pass executed, 8 after a transpose, where the bounds of the
// previous pass sit in the lane index and must be uniform); exit: below 8q (one subtraction for the registers that
// reached 16q), and below 4q after the last pass, whose final stage sends both sides through a product.  For L = 13 that
// is 36 conditional subtractions per 16 coefficients instead of 104.
template <int L, int P>
__device__ __forceinline__ void ntt_inv_pass4t(u64 (&x)[16], const ulonglong2 *__restrict__ itw, const NttMod &m, int tid) {
    constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    const int th = (P == 0) ? 0 : (tid >> LO);
    const u64 q8 = m.q4 << 1, q16 = m.q4 << 2;
    int bd[16];
#pragma unroll
    for (int r = 0; r < 16; r++) bd[r] = (P == NttShape<L>::NP - 1) ? 4 : 8;
#pragma unroll
    for (int u = S - 1; u >= 0; u--) {
        const int sigma = 4 * P + u, b = L - 1 - sigma, rb = b - LO;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const ulonglong2 w = itw[(1 << sigma) + ((th << (3 - rb)) | (r0 >> (rb + 1)))];
            const u64 X = x[r0], Y = x[r1];
            const u64 off = bd[r1] == 4 ? m.q4 : bd[r1] == 8 ? q8 : q16;
            u64 T = X + Y;                             // < (bd[r0] + bd[r1]) q <= 32q
            const u64 D = X - Y + off;                 // in (0, 32q)
            if (sigma == 0) {
                const ulonglong2 ni = itw[0];
                x[r0] = mul_shoup_lazy4(T, ni.x, ni.y, m.nq, m.zero);
                bd[r0] = 4;
            } else {
                int bs = bd[r0] + bd[r1];
                if (bs > 16) { T = csub(T, q16); bs = 16; }
                x[r0] = T;
                bd[r0] = bs;
            }
            x[r1] = mul_shoup_lazy4(D, w.x, w.y, m.nq, m.zero);    // [0, 4q)
            bd[r1] = 4;
        }
    }
    if (P > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (bd[r] > 8) x[r] = csub(x[r], q8);
    }
}

template <int LO_FROM, int LO_TO>
__device__ __forceinline__ void ntt_transpose(u64 (&x)[16], u64 *lds, int tid) {
    constexpr int PL = imin(LO_FROM, LO_TO);
    __syncthreads();   // earlier readers of this buffer are done
#pragma unroll
    for (int r = 0; r < 16; r++) lds[lds_pad<PL>(elem_index<LO_FROM>(tid, r))] = x[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = lds[lds_pad<PL>(elem_index<LO_TO>(tid, r))];
}

// all forward passes: in = pass-0 register mapping (natural order), out = last-pass mapping
template <int L, int P = 0>
__device__ __forceinline__ void ntt_fwd_regs(u64 (&x)[16], const ulonglong2 *__restrict__ tw, u64 q, u64 *lds, int tid) {
    ntt_fwd_pass<L, P>(x, tw, q, tid);
    if constexpr (P + 1 < NttShape<L>::NP) {
        ntt_transpose<pass_lo(L, P), pass_lo(L, P + 1)>(x, lds, tid);
        ntt_fwd_regs<L, P + 1>(x, tw, q, lds, tid);
    }
}
// all inverse passes: in = last-pass mapping with values in [0, 2q), out = pass-0 mapping, [0, 2q)
template <int L, int P = NttShape<L>::NP - 1>
__device__ __forceinline__ void ntt_inv_regs(u64 (&x)[16], const ulonglong2 *__restrict__ itw, u64 q, u64 *lds, int tid) {
    ntt_inv_pass<L, P>(x, itw, q, tid);
    if constexpr (P > 0) {
        ntt_transpose<pass_lo(L, P), pass_lo(L, P - 1)>(x, lds, tid);
        ntt_inv_regs<L, P - 1>(x, itw, q, lds, tid);
    }
}

// the same drivers on the lazy4 passes: forward out below 8q (LAZY: below (2 + 4L) q), inverse [0, 4q) -> [0, 4q)
template <int L, bool LAZY, int P = 0>
__device__ __forceinline__ void ntt_fwd_regs4(u64 (&x)[16], const ulonglong2 *__restrict__ tw, const NttMod &m, u64 *lds, int tid) {
    ntt_fwd_pass4<L, P, LAZY>(x, tw, m, tid);
    if constexpr (P + 1 < NttShape<L>::NP) {
        ntt_transpose<pass_lo(L, P), pass_lo(L, P + 1)>(x, lds, tid);
        ntt_fwd_regs4<L, LAZY, P + 1>(x, tw, m, lds, tid);
    }
}
// LAZY: every prime of the base has at most 58 bits (the range-tracking passes above)
template <int L, bool LAZY = false, int P = NttShape<L>::NP - 1>
__device__ __forceinline__ void ntt_inv_regs4(u64 (&x)[16], const ulonglong2 *__restrict__ itw, const NttMod &m, u64 *lds, int tid) {
    if constexpr (LAZY) ntt_inv_pass4t<L, P>(x, itw, m, tid);
    else ntt_inv_pass4<L, P>(x, itw, m, tid);
    if constexpr (P > 0) {
        ntt_transpose<pass_lo(L, P), pass_lo(L, P - 1)>(x, lds, tid);
        ntt_inv_regs4<L, LAZY, P - 1>(x, itw, m, lds, tid);
    }
}

// ---- M polynomials of one prime per workgroup: every twiddle pair is fetched once for the M butterflies that use it ----
template <int L, int P, bool LAZY, int M>
__device__ __forceinline__ void ntt_fwd_pass4m(u64 (&x)[M][16], const ulonglong2 *__restrict__ tw, const NttMod &m, int tid) {
    constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    const int th = (P == 0) ? 0 : (tid >> LO);
#pragma unroll
    for (int u = 0; u < S; u++) {
        const int sigma = 4 * P + u, b = L - 1 - sigma, rb = b - LO;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const ulonglong2 w = tw[(1 << sigma) + ((th << (3 - rb)) | (r0 >> (rb + 1)))];
#pragma unroll
            for (int j = 0; j < M; j++) {
                const u64 X = LAZY ? x[j][r0] : csub(x[j][r0], m.q4);
                const u64 S = mul_shoup_lazy4_acc(x[j][r1], w.x, w.y, m.nq, m.zero, X);     // X + T
                x[j][r0] = S;
                x[j][r1] = (X << 1) + m.q4 - S;                                              // X - T + 4q
            }
        }
    }
}
template <int L, bool LAZY, int M, int P = 0>
__device__ __forceinline__ void ntt_fwd_regs4m(u64 (&x)[M][16], const ulonglong2 *__restrict__ tw, const NttMod &m, u64 *lds, int tid) {
    ntt_fwd_pass4m<L, P, LAZY, M>(x, tw, m, tid);
    if constexpr (P + 1 < NttShape<L>::NP) {
#pragma unroll
        for (int j = 0; j < M; j++) ntt_transpose<pass_lo(L, P), pass_lo(L, P + 1)>(x[j], lds, tid);
        ntt_fwd_regs4m<L, LAZY, M, P + 1>(x, tw, m, lds, tid);
    }
}

// M polynomials of one prime.  The range-tracking pass is NOT used here: with two polynomials in flight it spills
// (k_ntt_inv2<13>: 80 B of scratch per lane against 20, -14 % measured; with the polynomial index innermost 272 B), so
// the pair kernels keep one conditional subtraction per butterfly whatever LAZY says.
template <int L, int P, int M, bool LAZY>
__device__ __forceinline__ void ntt_inv_pass4m(u64 (&x)[M][16], const ulonglong2 *__restrict__ itw, const NttMod &m, int tid) {
#pragma unroll
    for (int j = 0; j < M; j++) ntt_inv_pass4<L, P>(x[j], itw, m, tid);      // the twiddle loads of the M copies are merged by the compiler
}
template <int L, int M, bool LAZY = false, int P = NttShape<L>::NP - 1>
__device__ __forceinline__ void ntt_inv_regs4m(u64 (&x)[M][16], const ulonglong2 *__restrict__ itw, const NttMod &m, u64 *lds, int tid) {
    ntt_inv_pass4m<L, P, M, LAZY>(x, itw, m, tid);
    if constexpr (P > 0) {
#pragma unroll
        for (int j = 0; j < M; j++) ntt_transpose<pass_lo(L, P), pass_lo(L, P - 1)>(x[j], lds, tid);
        ntt_inv_regs4m<L, M, LAZY, P - 1>(x, itw, m, lds, tid);
    }
}

// global <-> register helpers
template <int L> __device__ __forceinline__ void load_coeff(u64 (&x)[16], const u64 *__restrict__ p, int tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = p[elem_index<L - 4>(tid, r)];
}
template <int L> __device__ __forceinline__ void store_coeff(const u64 (&x)[16], u64 *__restrict__ p, int tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) p[elem_index<L - 4>(tid, r)] = x[r];
}
template <int L> __device__ __forceinline__ void load_slots(u64 (&x)[16], const u64 *__restrict__ p, int tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = p[r * NttShape<L>::TP + tid];
}
template <int L> __device__ __forceinline__ void store_slots(const u64 (&x)[16], u64 *__restrict__ p, int tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) p[r * NttShape<L>::TP + tid] = x[r];
}
