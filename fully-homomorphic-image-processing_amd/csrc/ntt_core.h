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
#include <utility>
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

struct PmMod;
struct RnsBase {              // device pointers, passed to kernels by value
    const ulonglong2 *tw;     // [count][n]  psi^bitrev(i) with Shoup companion
    const ulonglong2 *itw;    // [count][n]  psi^-bitrev(i); entry 0 = n^-1, entry 1 pre-multiplied by n^-1
    const Modulus *mod;       // [count]
    u32 count;
    // pseudo-Mersenne tables (below; null when the base does not qualify): the same twiddles beside w 2^31 mod q
    const ulonglong2 *tw_pm, *itw_pm;
    const PmMod *pm;          // [count]
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

// A transpose between passes with register bits at [LO, LO + 4) regroups elements inside aligned blocks of 2^(max LO + 4)
// coefficients, i.e. among 2^(max LO) consecutive threads.  For max LO <= 6 that is (part of) ONE wavefront: every element a
// lane reads was written by a lane of its own wave, to a region of the buffer no other wave touches in this transpose -- and
// the wave's own earlier reads of that region (the previous transpose) are ordered before by the in-order LDS queue.  So the
// exchange needs no workgroup barrier at all, only that the compiler keeps the writes before the reads (wave-level fences):
// n = 8192 keeps 2 of its 6 s_barriers per transform (the first, 512-thread-wide transpose), n = 4096 2 of 4.  (Round 5; before,
// every transpose was bracketed by two __syncthreads.  NTT_WAVE_LOCAL_TRANSPOSE 0 restores that for A/B measurements.)
//
// CONTRACT of every driver below (ntt_fwd_regs* / ntt_inv_regs*): on entry no OTHER wave may still be reading `lds`.  That holds
// at kernel start and after a forward transform (its last transposes read wave-locally; a forward transform of L >= 11 also opens
// with a workgroup-wide transpose, i.e. with a barrier).  It does NOT hold after an inverse transform of L >= 11: that one ends
// with a workgroup-wide transpose whose READ phase fetches elements from every wave's region, and the next inverse transform
// opens with a wave-local transpose that writes its region without a barrier -- a lagging wave would read clobbered data.  A
// kernel that runs a second transform on the same buffer after an inverse one calls ntt_lds_release() in between (k_rgb2ycc).
#ifndef NTT_WAVE_LOCAL_TRANSPOSE
#define NTT_WAVE_LOCAL_TRANSPOSE 1
#endif
// wave64: the wave-local transposes and every 64-lane shuffle fold of csrc/ assume it.  ROCm 7 no longer defines
// __AMDGCN_WAVEFRONT_SIZE; gfx9 has no wave32 mode, so "compiled for gfx950" is the assertion (and the macro is checked where it exists)
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__)
#error "csrc/ is written for gfx950 (wave64) only"
#endif
#if defined(__AMDGCN_WAVEFRONT_SIZE)
static_assert(__AMDGCN_WAVEFRONT_SIZE == 64, "wave64 only");
#endif
#endif
__device__ __forceinline__ void ntt_lds_release() { __syncthreads(); }   // every wave is done reading the exchange buffer
template <int LO_FROM, int LO_TO>
__device__ __forceinline__ void ntt_transpose(u64 (&x)[16], u64 *lds, int tid) {
    constexpr int PL = imin(LO_FROM, LO_TO);
    constexpr bool wave_local = NTT_WAVE_LOCAL_TRANSPOSE && (LO_FROM <= 6) && (LO_TO <= 6);
    if constexpr (wave_local) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();   // earlier readers of this buffer are done
    }
#pragma unroll
    for (int r = 0; r < 16; r++) lds[lds_pad<PL>(elem_index<LO_FROM>(tid, r))] = x[r];
    if constexpr (wave_local) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
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

// ---- pseudo-Mersenne passes ------------------------------------------------------------------------------------------
// Every prime SEAL 2.3 hands out (and every auxiliary prime the library picks) is q = 2^b - delta with a small delta
// (delta < 2^25 for b = 54 .. 58).  A twiddle w kept beside w2 = w 2^31 mod q gives x w mod q for ANY x < 2^62 as
//     x = xl + 2^31 xh;  S = xl w + xh w2 < 2^(b+32):  A = xl wl + xh w2l (< 2^64),  B = xl wh + xh w2h + (A >> 32);
//     S = zl + 2^b zh,  zh = B >> (b - 32) < 2^32;   result = zl + zh delta  in [0, 2^b + 2^32 delta)
// -- five v_mad_u64_u32 in two chains and no v_mul_hi_u32, against seven and two for the Shoup product with the
// approximate high word (tools/ubench5.hip: 3.81 against 3.04 forward butterflies per clock per CU).  A value is brought
// back below 1.0001 x 2^b by ONE multiply-add: fold(x) = (x mod 2^b) + (x >> b) delta, where the Shoup passes spend a
// compare, two selects and a 64-bit subtraction per conditional subtraction.  Ranges are tracked statically in sixteenths
// of q: LIM = the largest product operand (2^62 / q, rounded down to a power of two), RQ = the product's result bound.
struct PmMod {
    u64 q;
    u32 delta, sh, mb, pad;    // q = 2^b - delta, sh = b - 32, mb = 2^sh - 1
};
__device__ __forceinline__ u64 pm_pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ u64 mul_pm(u64 x, const ulonglong2 w, const PmMod &m) {
    const u32 xl = (u32)x & 0x7fffffffu, xh = __builtin_amdgcn_alignbit((u32)(x >> 32), (u32)x, 31);
    const u32 wl = (u32)w.x, wh = (u32)(w.x >> 32), vl = (u32)w.y, vh = (u32)(w.y >> 32);
    const u64 A = (u64)xh * vl + (u64)xl * wl;
    u64 B = (u64)xl * wh + (A >> 32);
    asm("" : "+v"(B));                      // keeps the addend inside the multiply-add (the compiler would re-associate it out)
    B = (u64)xh * vh + B;
    const u32 zh = __builtin_amdgcn_alignbit((u32)(B >> 32), (u32)B, m.sh);
    u64 zl = pm_pack((u32)A, (u32)B & m.mb);
    asm("" : "+v"(zl));
    return (u64)zh * m.delta + zl;
}
// any 64-bit x -> x mod q + (0 or 1) q, below 2^b + 2^(64-b) delta < (17/16) q
__device__ __forceinline__ u64 fold_pm(u64 x, const PmMod &m) {
    const u32 top = (u32)(x >> 32) >> m.sh;
    u64 lo = pm_pack((u32)x, (u32)(x >> 32) & m.mb);
    asm("" : "+v"(lo));
    return (u64)top * m.delta + lo;
}
// a b mod q for a below 2^(b+1) (a folded value) and a canonical b, neither with a prepared companion: the 128-bit
// product (four multiply-adds), then two folds of its part above 2^b (three multiply-adds).  Result below
// 2^b + 2^(85-b) delta, which the host checks against the class's RQ.
__device__ __forceinline__ u64 mulvv_pm(u64 a, u64 b, const PmMod &m) {
    const u32 al = (u32)a, ah = (u32)(a >> 32), bl = (u32)b, bh = (u32)(b >> 32);
    const u64 P0 = (u64)al * bl;
    u64 mid = (u64)al * bh + (P0 >> 32);
    asm("" : "+v"(mid));
    mid = (u64)ah * bl + mid;
    u64 top = (u64)ah * bh + (mid >> 32);
    asm("" : "+v"(top));
    // z = P0.lo + 2^32 mid.lo + 2^64 top;  zh = z >> b (below 2^(b+1)),  zl = z mod 2^b
    const u32 zh_lo = __builtin_amdgcn_alignbit((u32)top, (u32)mid, m.sh), zh_hi = __builtin_amdgcn_alignbit((u32)(top >> 32), (u32)top, m.sh);
    u64 zl = pm_pack((u32)P0, (u32)mid & m.mb);
    asm("" : "+v"(zl));
    const u64 F = (u64)zh_lo * m.delta + zl;
    u64 G = (u64)zh_hi * m.delta + (F >> 32);
    asm("" : "+v"(G));
    const u32 zh2 = __builtin_amdgcn_alignbit((u32)(G >> 32), (u32)G, m.sh);
    u64 lo = pm_pack((u32)F, (u32)G & m.mb);
    asm("" : "+v"(lo));
    return (u64)zh2 * m.delta + lo;
}
// The two classes of bases the library instantiates (host side: fhe_build_base picks one or none):
//   A  every prime <= 55 bits, product below 6 q:     LIM = 128 q, RQ = 6 q        (SEAL's 54- and 55-bit primes: delta < 2^25.1)
//   B  every prime <= 58 bits, product below 1.5 q:   LIM = 16 q,  RQ = 1.5 q      (the 58-bit auxiliary base: delta < 2^25)
// CS: forward offset 2^CS q >= RQ;  XB: exit bound of an inverse pass (the value with the fewest folds: 11 and 17 per
// sixteen coefficients of an 8192-point transform, beside its 104 butterflies).
struct PmA { static constexpr int LIM = 2048, RQ = 96, CS = 3, XB = 192; };
struct PmB { static constexpr int LIM = 256, RQ = 24, CS = 1, XB = 48; };
constexpr int PM_FOLDED = 17;               // bound of fold_pm's result, sixteenths of q
__host__ __device__ constexpr int pm_ceil_log2_q(int bd16) {      // smallest s with 2^s q >= bd16 / 16 q
    int s = 0;
    while ((16 << s) < bd16) s++;
    return s;
}
// Forward (Cooley-Tukey, Harvey form without conditional subtractions): X' = X + T, Y' = X - T + 2^CS q with 2^CS q >= the
// product bound RQ; every register grows by at most 2^CS q per stage, uniformly, so ONE bound describes the pass.  When the
// product operand would pass LIM all sixteen registers are folded first (58-bit primes: once per transform; 55-bit: never).
__host__ __device__ constexpr int pm_fwd_bound(int e0, int lim, int cs, int stage) {   // bound entering `stage`, after its fold if it has one
    int bd = e0;
    for (int s = 0; s < stage; s++) {
        bd += 16 << cs;
        if (bd > lim) bd = PM_FOLDED;       // stage s + 1 folds first
    }
    return bd;
}
// Twiddles of one pass, fetched AHEAD of their use: the (w, w2) pairs of stage U sit in tw[U][0 .. 8 >> rb).  Left to the
// compiler, every 16-byte twiddle load of passes 1.. is issued a few dozen instructions before its first use and each wave
// stands still for most of an L2 round trip, fifteen times per pass; the drivers below issue the loads of a pass's first
// stages BEFORE the LDS transpose that precedes it and the rest one or two stages ahead, behind compiler fences.
struct PmPassTw { ulonglong2 w[4][8]; };
template <int L, int P, int U>
__device__ __forceinline__ void pm_tw_load(PmPassTw &t, const ulonglong2 *__restrict__ tw, int tid) {
    constexpr int LO = pass_lo(L, P), sigma = 4 * P + U, b = L - 1 - sigma, rb = b - LO;
    const int th = (P == 0) ? 0 : (tid >> LO);
    // the pseudo-Mersenne tables keep the 2^sigma twiddles of a stage as [i][th] where the Shoup tables have [th][i]
    // (pm_tw_index, host side): consecutive lanes read consecutive pairs, one or two cache lines per load where the
    // [th][i] order touches up to 64
#pragma unroll
    for (int i = 0; i < (8 >> rb); i++) t.w[U][i] = tw[(1 << sigma) + (i << (sigma - 3 + rb)) + th];
}
// position of twiddle number `idx` (the Shoup tables' index, 2^sigma + th (8 >> rb) + i) in a pseudo-Mersenne table
__host__ __device__ constexpr unsigned pm_tw_index(int L, unsigned idx) {
    if (idx < 2) return idx;
    int sigma = 0;
    while ((2u << sigma) <= idx) sigma++;
    const int P = sigma / 4, b = L - 1 - sigma, rb = b - pass_lo(L, P);
    const unsigned off = idx - (1u << sigma), cnt = 8u >> rb, th = off / cnt, i = off % cnt;
    return (1u << sigma) + (i << (sigma - 3 + rb)) + th;
}
#define PM_FENCE() asm volatile("" ::: "memory")
template <int L, int P, int U, int M, int E0, int LIM, int CS>
__device__ __forceinline__ void ntt_fwd_stage_pm(u64 (&x)[M][16], const PmPassTw &t, const PmMod &m, u64 off) {
    constexpr int LO = pass_lo(L, P), sigma = 4 * P + U, b = L - 1 - sigma, rb = b - LO;
    constexpr bool folds = pm_fwd_bound(E0, LIM, CS, sigma) == PM_FOLDED && sigma > 0 && pm_fwd_bound(E0, LIM, CS, sigma - 1) + (16 << CS) > LIM;
    if constexpr (folds) {
#pragma unroll
        for (int j = 0; j < M; j++) {
#pragma unroll
            for (int r = 0; r < 16; r++) x[j][r] = fold_pm(x[j][r], m);
        }
    }
#pragma unroll
    for (int r0 = 0; r0 < 16; r0++) {
        if (r0 & (1 << rb)) continue;
        const int r1 = r0 | (1 << rb);
        const ulonglong2 w = t.w[U][r0 >> (rb + 1)];
#pragma unroll
        for (int j = 0; j < M; j++) {
            const u64 X = x[j][r0], T = mul_pm(x[j][r1], w, m);
            x[j][r0] = X + T;
            x[j][r1] = X - T + off;
        }
    }
}
// stages [0, PRE) of the pass have their twiddles in t already; the others are fetched two stages ahead
template <int L, int P, int M, int E0, int LIM, int CS, int PRE>
__device__ __forceinline__ void ntt_fwd_pass_pm(u64 (&x)[M][16], PmPassTw &t, const ulonglong2 *__restrict__ tw, const PmMod &m, int tid) {
    constexpr int S = pass_stages(L, P);
    static_assert(E0 <= LIM, "the first stage takes its operands as they come");
    const u64 off = m.q << CS;
    if constexpr (PRE < 1) pm_tw_load<L, P, 0>(t, tw, tid);
    if constexpr (S > 1 && PRE < 2) pm_tw_load<L, P, 1>(t, tw, tid);
    ntt_fwd_stage_pm<L, P, 0, M, E0, LIM, CS>(x, t, m, off);
    if constexpr (S > 2 && PRE < 3) { pm_tw_load<L, P, 2>(t, tw, tid); PM_FENCE(); }
    if constexpr (S > 1) ntt_fwd_stage_pm<L, P, 1, M, E0, LIM, CS>(x, t, m, off);
    if constexpr (S > 3 && PRE < 4) { pm_tw_load<L, P, 3>(t, tw, tid); PM_FENCE(); }
    if constexpr (S > 2) ntt_fwd_stage_pm<L, P, 2, M, E0, LIM, CS>(x, t, m, off);
    if constexpr (S > 3) ntt_fwd_stage_pm<L, P, 3, M, E0, LIM, CS>(x, t, m, off);
}
// how many stages of pass P have their twiddles fetched before the transpose in front of it: all but the last (7 pairs
// at most), or the only one
__host__ __device__ constexpr int pm_fwd_pre(int L, int P) { return P == 0 ? 0 : (pass_stages(L, P) == 1 ? 1 : imin(pass_stages(L, P) - 1, 3)); }
// all forward passes; values enter below E0 / 16 q and leave below pm_fwd_bound(E0, LIM, CS, L) / 16 q
template <int L, int M, int E0, int LIM, int CS, int P = 0>
__device__ __forceinline__ void ntt_fwd_regs_pm(u64 (&x)[M][16], const ulonglong2 *__restrict__ tw, const PmMod &m, u64 *lds, int tid, PmPassTw *pre = nullptr) {
    PmPassTw t0;
    PmPassTw &t = (P == 0) ? t0 : *pre;
    ntt_fwd_pass_pm<L, P, M, E0, LIM, CS, pm_fwd_pre(L, P)>(x, t, tw, m, tid);
    if constexpr (P + 1 < NttShape<L>::NP) {
        PmPassTw nx;
        constexpr int PRE = pm_fwd_pre(L, P + 1);
        pm_tw_load<L, P + 1, 0>(nx, tw, tid);
        if constexpr (PRE > 1) pm_tw_load<L, P + 1, 1>(nx, tw, tid);
        if constexpr (PRE > 2) pm_tw_load<L, P + 1, 2>(nx, tw, tid);
        PM_FENCE();
#pragma unroll
        for (int j = 0; j < M; j++) ntt_transpose<pass_lo(L, P), pass_lo(L, P + 1)>(x[j], lds, tid);
        ntt_fwd_regs_pm<L, M, E0, LIM, CS, P + 1>(x, tw, m, lds, tid, &nx);
    }
}
// canonical residue of a forward output
__device__ __forceinline__ u64 canon_pm(u64 v, const PmMod &m) { return csub(fold_pm(v, m), m.q); }

// Inverse (Gentleman-Sande) with the bound of every register tracked at compile time (which register is the sum side and
// which the product side of a butterfly is static after unrolling): T = X + Y stays unreduced, D = X - Y + 2^s q with
// 2^s q >= bound(Y) goes through the product, which resets it to RQ.  An operand is folded only when D would pass LIM.
// Entry: every register below EB / 16 q.  Exit (P > 0): registers above XB / 16 q are folded, so the next pass -- where
// the bounds sit in the lane index after the transpose -- enters with EB = XB.  After the last pass (P == 0) both sides
// of every butterfly came out of a product: below RQ / 16 q.
struct PmInvPlan {
    bool fold_y[4][16], fold_x[4][16];       // [stage of the pass][r0]: fold the operand before the butterfly
    int shift[4][16];                        // offset 2^shift q of the difference
    bool fold_exit[16];
};
template <int L, int P>
__host__ __device__ constexpr PmInvPlan pm_inv_plan(int EB, int XB, int LIM, int RQ) {
    constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    PmInvPlan pl{};
    int bd[16] = {};
    for (int r = 0; r < 16; r++) bd[r] = EB;
    for (int u = S - 1; u >= 0; u--) {
        const int sigma = 4 * P + u, b = L - 1 - sigma, rb = b - LO;
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            if (bd[r0] + (16 << pm_ceil_log2_q(bd[r1])) > LIM) { pl.fold_y[u][r0] = true; bd[r1] = PM_FOLDED; }     // the cheapest way to shrink the offset
            if (bd[r0] + (16 << pm_ceil_log2_q(bd[r1])) > LIM) { pl.fold_x[u][r0] = true; bd[r0] = PM_FOLDED; }
            pl.shift[u][r0] = pm_ceil_log2_q(bd[r1]);
            bd[r0] = sigma == 0 ? RQ : bd[r0] + bd[r1];
            bd[r1] = RQ;
        }
    }
    for (int r = 0; r < 16; r++) pl.fold_exit[r] = P > 0 && bd[r] > XB;
    return pl;
}
template <int L, int P, int U, int R0, int M, int EB, int XB, int LIM, int RQ>
__device__ __forceinline__ void ntt_inv_bfly_pm(u64 (&x)[M][16], const PmPassTw &t, const ulonglong2 ninv, const PmMod &m) {
    constexpr int LO = pass_lo(L, P), sigma = 4 * P + U, b = L - 1 - sigma, rb = b - LO;
    if constexpr ((R0 & (1 << rb)) == 0) {
        constexpr PmInvPlan pl = pm_inv_plan<L, P>(EB, XB, LIM, RQ);
        constexpr int r1 = R0 | (1 << rb);
        const ulonglong2 w = t.w[U][R0 >> (rb + 1)];
        const u64 off = m.q << pl.shift[U][R0];
#pragma unroll
        for (int j = 0; j < M; j++) {
            u64 X = x[j][R0], Y = x[j][r1];
            if constexpr (pl.fold_y[U][R0]) Y = fold_pm(Y, m);
            if constexpr (pl.fold_x[U][R0]) X = fold_pm(X, m);
            const u64 T = X + Y;
            const u64 D = X - Y + off;
            if constexpr (sigma == 0) x[j][R0] = mul_pm(T, ninv, m);
            else x[j][R0] = T;
            x[j][r1] = mul_pm(D, w, m);
        }
    }
}
template <int L, int P, int U, int M, int EB, int XB, int LIM, int RQ, int... R>
__device__ __forceinline__ void ntt_inv_stage_pm(u64 (&x)[M][16], const PmPassTw &t, const ulonglong2 ninv, const PmMod &m, std::integer_sequence<int, R...>) {
    (ntt_inv_bfly_pm<L, P, U, R, M, EB, XB, LIM, RQ>(x, t, ninv, m), ...);
}
template <int P, int M, typename Plan, int... R>
__device__ __forceinline__ void ntt_inv_exit_pm(u64 (&x)[M][16], const PmMod &m, std::integer_sequence<int, R...>) {
    ([&] {
        if constexpr (Plan::plan.fold_exit[R]) {
#pragma unroll
            for (int j = 0; j < M; j++) x[j][R] = fold_pm(x[j][R], m);
        }
    }(), ...);
}
template <int L, int P, int EB, int XB, int LIM, int RQ> struct PmInvPlanOf { static constexpr PmInvPlan plan = pm_inv_plan<L, P>(EB, XB, LIM, RQ); };
// the stages run S-1 .. 0; the top PRE of them have their twiddles in t already, the others are fetched two stages ahead
template <int L, int P, int M, int EB, int XB, int LIM, int RQ, int PRE>
__device__ __forceinline__ void ntt_inv_pass_pm(u64 (&x)[M][16], PmPassTw &t, const ulonglong2 *__restrict__ itw, const PmMod &m, int tid) {
    constexpr int S = pass_stages(L, P);
    using Seq = std::make_integer_sequence<int, 16>;
    ulonglong2 ninv = make_ulonglong2(0, 0);
    if constexpr (P == 0) ninv = itw[0];
    if constexpr (PRE < 1) pm_tw_load<L, P, S - 1>(t, itw, tid);
    if constexpr (S > 1 && PRE < 2) pm_tw_load<L, P, S - 2>(t, itw, tid);
    if constexpr (S > 2 && PRE < 3) { pm_tw_load<L, P, S - 3>(t, itw, tid); PM_FENCE(); }
    ntt_inv_stage_pm<L, P, S - 1, M, EB, XB, LIM, RQ>(x, t, ninv, m, Seq{});
    if constexpr (S > 3 && PRE < 4) { pm_tw_load<L, P, S - 4>(t, itw, tid); PM_FENCE(); }
    if constexpr (S > 1) ntt_inv_stage_pm<L, P, S - 2, M, EB, XB, LIM, RQ>(x, t, ninv, m, Seq{});
    if constexpr (S > 2) ntt_inv_stage_pm<L, P, S - 3, M, EB, XB, LIM, RQ>(x, t, ninv, m, Seq{});
    if constexpr (S > 3) ntt_inv_stage_pm<L, P, S - 4, M, EB, XB, LIM, RQ>(x, t, ninv, m, Seq{});
    ntt_inv_exit_pm<P, M, PmInvPlanOf<L, P, EB, XB, LIM, RQ>>(x, m, Seq{});
}
// stages of pass P fetched before the transpose in front of it: the first two it runs (8 + 4 pairs); pass 0's addresses
// are uniform (scalar loads), nothing to gain
__host__ __device__ constexpr int pm_inv_pre(int L, int P) { return (P == 0 || P == (L + 3) / 4 - 1) ? 0 : imin(pass_stages(L, P), 2); }
// all inverse passes: in = last-pass mapping, below E0 / 16 q; out = pass-0 mapping, below RQ / 16 q
template <int L, int M, int E0, int XB, int LIM, int RQ, int P = NttShape<L>::NP - 1>
__device__ __forceinline__ void ntt_inv_regs_pm(u64 (&x)[M][16], const ulonglong2 *__restrict__ itw, const PmMod &m, u64 *lds, int tid, PmPassTw *pre = nullptr) {
    PmPassTw t0;
    PmPassTw &t = pre ? *pre : t0;
    ntt_inv_pass_pm<L, P, M, (P == NttShape<L>::NP - 1 ? E0 : XB), XB, LIM, RQ, pm_inv_pre(L, P)>(x, t, itw, m, tid);
    if constexpr (P > 0) {
        PmPassTw nx;
        constexpr int PRE = pm_inv_pre(L, P - 1), S1 = pass_stages(L, P - 1);
        if constexpr (PRE > 0) pm_tw_load<L, P - 1, S1 - 1>(nx, itw, tid);
        if constexpr (PRE > 1) pm_tw_load<L, P - 1, S1 - 2>(nx, itw, tid);
        if constexpr (PRE > 0) PM_FENCE();
#pragma unroll
        for (int j = 0; j < M; j++) ntt_transpose<pass_lo(L, P), pass_lo(L, P - 1)>(x[j], lds, tid);
        ntt_inv_regs_pm<L, M, E0, XB, LIM, RQ, P - 1>(x, itw, m, lds, tid, PRE > 0 ? &nx : nullptr);
    }
}
// canonical residue of an inverse output (below RQ / 16 q)
template <int RQ> __device__ __forceinline__ u64 canon_rq_pm(u64 v, const PmMod &m) {
    if constexpr (RQ > 32) v = fold_pm(v, m);
    return csub(v, m.q);
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
