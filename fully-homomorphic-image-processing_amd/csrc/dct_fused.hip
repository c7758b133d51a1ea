// dct_fused.hip -- the headline kernels: encrypted_dct + quantize_fhe (homo/fhe_image.h:196-305)
// as two fused launches per wave of blocks, with exact FP64-FMA modular arithmetic.
//
// Why FP64.  gfx950 has no 64-bit integer multiplier; a Shoup modular product costs ~10 quarter/half
// rate 32-bit multiplies (measured 4.6 lane-products/clk/CU, profiles/r01_ubench_*).  For primes
// below 2^48 the same residue can be computed EXACTLY with five double-precision operations
//     h = y*w;  l = fma(y,w,-h);  q = rint(h/p);  r = fma(-q,p,h) + l          (|r| <= p/2 + eps)
// (h+l is the exact 106-bit product, q p is exact, so r is the exact centred remainder), measured
// at 10.1 lane-products/clk/CU.  Values stay integers of magnitude < 2^53 throughout, additions
// need no reduction at all, and the final residues are bit-identical to the u64 path.  This is
// integer modular arithmetic carried by the FP64 FMA pipe, not floating-point approximation.
//
// Dataflow per (block, polynomial j, prime i):
//   kernel A (rows):    for each row: x_m = d_m +- d_(7-m) on coefficients, forward NTT of the
//                       four sums (even half) or differences (odd half), the even/odd part of the
//                       LL&M row pass per NTT slot, store the four row outputs (NTT form; packed to
//                       40 bytes per thread and polynomial for primes <= 37 bits, FP64 otherwise)
//   kernel B (columns): same for columns on the NTT-form intermediates, per-output scale
//                       encode(0.125)*encode(1/quant) folded into one product, inverse NTT, store.
// Each workgroup carries 4 polynomials x 2^LE coefficients per thread in registers (n = 4096: LE = 3,
// 512 threads, 64 data VGPRs, two workgroups = four waves per SIMD) and shares every twiddle across
// the four; the even and odd workgroup of a line read the same eight inputs and are placed on the
// same XCD (blockIdx = 16g+u and 16g+8+u) so the second
// read is an L2 hit.  All 832 Evaluator calls of the reference are exact ring operations, so this
// evaluation order yields bit-identical ciphertexts.
#include "internal.h"

#include <cstdlib>

#include "fp64_core.h"

#pragma clang fp contract(off)

namespace {
using namespace fp64;

// Register-pass geometry with 2^LE coefficients per thread (LE = 3: 512 threads per 4096-point
// polynomial, 64 data VGPRs for four polynomials -> 4 waves/SIMD; LE = 4: 256 threads, 128 VGPRs).
template <int L, int LE> struct Shape {
    static constexpr int N = 1 << L, E = 1 << LE, TP = N >> LE, NP = (L + LE - 1) / LE, LDS_WORDS = N + (N >> LE);
};
__host__ __device__ constexpr int g_lo(int L, int LE, int p) { return (L - LE * p - LE) < 0 ? 0 : (L - LE * p - LE); }
__host__ __device__ constexpr int g_stages(int L, int LE, int p) { return (L - LE * p) > LE ? LE : (L - LE * p); }
template <int LO, int LE> __device__ __forceinline__ int g_index(int tid, int r) {
    return ((tid >> LO) << (LO + LE)) | (r << LO) | (tid & ((1 << LO) - 1));
}
template <int LO, int LE> __device__ __forceinline__ int g_pad(int j) { return j + ((j >> (LO + LE)) << LO); }

// twiddles of one register pass, fetched ahead of use (before the preceding LDS barrier) so that
// their L2 latency is hidden behind the previous pass: 2^(LE-1-rb) values per stage, < 2^LE in all.
template <int L, int LE, int P> struct PassTw {
    static constexpr int LO = g_lo(L, LE, P), S = g_stages(L, LE, P), NTW = (1 << LE) - 1;
    static constexpr int rb(int u) { return (L - 1 - (LE * P + u)) - LO; }
    static constexpr int count(int u) { return 1 << (LE - 1 - rb(u)); }
    static constexpr int offset(int u) { int o = 0; for (int v = 0; v < u; v++) o += count(v); return o; }
};
template <int L, int LE, int P>
__device__ __forceinline__ void load_tw(double (&w)[(1 << LE) - 1], const double *__restrict__ tw, int tid) {
    using T = PassTw<L, LE, P>;
    const int th = (P == 0) ? 0 : (tid >> T::LO);
#pragma unroll
    for (int u = 0; u < T::S; u++) {
#pragma unroll
        for (int i = 0; i < T::count(u); i++) w[T::offset(u) + i] = tw[(1 << (LE * P + u)) + ((th << (LE - 1 - T::rb(u))) | i)];
    }
}

template <int L, int LE, int P, int M>
__device__ __forceinline__ void fwd_pass(double (&x)[M][1 << LE], const double (&w)[(1 << LE) - 1], double p, double pinv) {
    using T = PassTw<L, LE, P>;
    constexpr int E = 1 << LE, KB = (LE == 3) ? 2 : 1;     // butterflies batched per product group
#pragma unroll
    for (int u = 0; u < T::S; u++) {
        const int rb = T::rb(u);
        // enumerate the E/2 butterflies of this stage: index b -> r0 = b with a zero inserted at bit rb
#pragma unroll
        for (int b0 = 0; b0 < E / 2; b0 += KB) {
            double t[KB * M], wm[KB * M];
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                const int b = b0 + kb, r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
                const double wv = w[T::offset(u) + (r0 >> (rb + 1))];
#pragma unroll
                for (int m = 0; m < M; m++) { t[kb * M + m] = x[m][r1]; wm[kb * M + m] = wv; }
            }
            mmv<KB * M>(t, wm, p, pinv);
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                const int b = b0 + kb, r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const double X = x[m][r0];
                    x[m][r0] = X + t[kb * M + m];
                    x[m][r1] = X - t[kb * M + m];
                }
            }
        }
    }
}

template <int L, int LE, int P, int M>
__device__ __forceinline__ void inv_pass(double (&x)[M][1 << LE], const double (&w)[(1 << LE) - 1], double ninv, double p, double pinv) {
    using T = PassTw<L, LE, P>;
    constexpr int E = 1 << LE, KB = (LE == 3) ? 2 : 1;
#pragma unroll
    for (int u = T::S - 1; u >= 0; u--) {
        const int sigma = LE * P + u, rb = T::rb(u);
        if (sigma == 0) {          // last stage of the whole transform: both outputs carry n^-1
#pragma unroll
            for (int b = 0; b < E / 2; b++) {
                const int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
                const double wv = w[T::offset(u) + (r0 >> (rb + 1))];
                double t[2 * M], wm[2 * M];
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const double X = x[m][r0], Y = x[m][r1];
                    t[m] = X + Y; wm[m] = ninv;
                    t[M + m] = X - Y; wm[M + m] = wv;
                }
                mmv<2 * M>(t, wm, p, pinv);
#pragma unroll
                for (int m = 0; m < M; m++) { x[m][r0] = t[m]; x[m][r1] = t[M + m]; }
            }
        } else {
#pragma unroll
            for (int b0 = 0; b0 < E / 2; b0 += KB) {
                double t[KB * M], wm[KB * M];
#pragma unroll
                for (int kb = 0; kb < KB; kb++) {
                    const int b = b0 + kb, r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
                    const double wv = w[T::offset(u) + (r0 >> (rb + 1))];
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        const double X = x[m][r0], Y = x[m][r1];
                        x[m][r0] = X + Y;
                        t[kb * M + m] = X - Y; wm[kb * M + m] = wv;
                    }
                }
                mmv<KB * M>(t, wm, p, pinv);
#pragma unroll
                for (int kb = 0; kb < KB; kb++) {
                    const int b = b0 + kb, r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
#pragma unroll
                    for (int m = 0; m < M; m++) x[m][r1] = t[kb * M + m];
                }
            }
        }
    }
}

// M polynomials through two alternating LDS buffers: one barrier per polynomial.
// When both layouts keep the wave bits of the thread id (tid >> 6) on the same index bits -- every
// exchange except the one next to pass 0 -- a wave reads only what it wrote itself: no workgroup
// barrier is needed (LDS operations of one wave complete in order), and the eight waves of a
// workgroup are free to drift apart instead of meeting twelve times per transform.
template <int L, int LE, int LO_FROM, int LO_TO, int M>
__device__ __forceinline__ void transpose(double (&x)[M][1 << LE], double *lds, int tid, int &phase) {
    constexpr int PL = imin(LO_FROM, LO_TO), E = 1 << LE;
    constexpr bool WAVE_LOCAL = (Shape<L, LE>::TP <= 64) || (LO_FROM <= 6 && LO_TO <= 6);
#pragma unroll
    for (int m = 0; m < M; m++) {
        double *buf = lds + (phase & 1) * Shape<L, LE>::LDS_WORDS;
        phase++;
#pragma unroll
        for (int r = 0; r < E; r++) buf[g_pad<PL, LE>(g_index<LO_FROM, LE>(tid, r))] = x[m][r];
        if constexpr (WAVE_LOCAL) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < E; r++) x[m][r] = buf[g_pad<PL, LE>(g_index<LO_TO, LE>(tid, r))];
    }
}

// forward transform; `w` holds the pass-0 twiddles on entry.  `pre()` is invoked once, right
// before the last transpose, so that the caller can start fetching what it needs after the NTT.
template <int L, int LE, int M, typename PRE, int P = 0>
__device__ __forceinline__ void ntt_fwd(double (&x)[M][1 << LE], double (&w)[(1 << LE) - 1], const double *__restrict__ tw, double p, double pinv,
                                        double *lds, int tid, int &phase, PRE pre) {
    if constexpr (P + 1 < Shape<L, LE>::NP) {
        double wn[(1 << LE) - 1];
        if (LE >= 4) load_tw<L, LE, P + 1>(wn, tw, tid);   // in flight during this pass and the transpose
        fwd_pass<L, LE, P, M>(x, w, p, pinv);
        if (LE < 4) load_tw<L, LE, P + 1>(wn, tw, tid);    // four waves/SIMD: the transpose alone hides it
        if constexpr (P + 2 == Shape<L, LE>::NP) pre();
        transpose<L, LE, g_lo(L, LE, P), g_lo(L, LE, P + 1), M>(x, lds, tid, phase);
        ntt_fwd<L, LE, M, PRE, P + 1>(x, wn, tw, p, pinv, lds, tid, phase, pre);
    } else {
        fwd_pass<L, LE, P, M>(x, w, p, pinv);
    }
}
template <int L, int LE, int M, bool BIG, int P = Shape<L, LE>::NP - 1>
__device__ __forceinline__ void ntt_inv(double (&x)[M][1 << LE], double (&w)[(1 << LE) - 1], const double *__restrict__ itw, double p, double pinv,
                                        double *lds, int tid, int &phase) {
    if constexpr (P > 0) {
        double wn[(1 << LE) - 1];
        if (LE >= 4) load_tw<L, LE, P - 1>(wn, itw, tid);
        inv_pass<L, LE, P, M>(x, w, 0.0, p, pinv);
        if (LE < 4) load_tw<L, LE, P - 1>(wn, itw, tid);
        if constexpr (BIG) {
#pragma unroll
            for (int m = 0; m < M; m++)
#pragma unroll
                for (int r = 0; r < (1 << LE); r++) x[m][r] = red(x[m][r], p, pinv);
        }
        transpose<L, LE, g_lo(L, LE, P), g_lo(L, LE, P - 1), M>(x, lds, tid, phase);
        ntt_inv<L, LE, M, BIG, P - 1>(x, wn, itw, p, pinv, lds, tid, phase);
    } else {
        inv_pass<L, LE, P, M>(x, w, itw[0], p, pinv);
    }
}

// Packed intermediate (primes <= 37 bits): a row output is an integer |v| < 2^39, so it travels as
// its low 32 bits plus one signed high byte -- 40 bytes per thread and polynomial instead of 64.
// t = v + (2^52 + 2^51) has bits(t) = (0x43380000 + floor(v / 2^32)) : (v mod 2^32), so packing is one
// FP64 add and byte picks, unpacking (cols_body) rebuilds t from (low word, 0x43380000 + sign-extended byte),
// and the column kernel's first butterfly works on the biased values directly:
//   t_a - t_b = a - b,   t_a + (t_b - 2 (2^52 + 2^51)) = a + b      (all exact).
constexpr double PACK_BIAS = 6755399441055744.0;
template <int E, int TP>
__device__ __forceinline__ void store_packed(u64 *__restrict__ o, const double (&x)[E]) {
    static_assert(E == 8, "packed layout is defined for 8 values per thread");
    u32 lo[8], hi[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double t = x[r] + PACK_BIAS;
        lo[r] = (u32)__double2loint(t);
        hi[r] = (u32)__double2hiint(t);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) o[q * TP] = (u64)lo[2 * q] | ((u64)lo[2 * q + 1] << 32);
    u32 h[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const u32 p01 = __builtin_amdgcn_perm(hi[4 * g + 1], hi[4 * g + 0], 0x0c0c0400u);
        const u32 p23 = __builtin_amdgcn_perm(hi[4 * g + 3], hi[4 * g + 2], 0x0c0c0400u);
        h[g] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
    }
    o[4 * TP] = (u64)h[0] | ((u64)h[1] << 32);
}
// Circuit constants through LDS (gfx950 `global_load_lds_dwordx4`): the per-slot constants of a wave -- NT tables x 64
// doubles -- are brought straight from L2 into that wave's own LDS region without passing through VGPRs, one slot
// ahead (two regions per wave, used alternately): the request for slot r+1 is issued before slot r is evaluated, so
// the L2 round trip that each of the eight per-slot rounds used to expose runs behind the products of the current
// slot, and no prefetch registers are needed (the 128-VGPR budget is what defeated every register prefetch,
// DESIGN.md 5c).  Requests of one wave complete in order, so `s_waitcnt vmcnt(<requests of slot r+1>)` means slot r
// has landed; the waits are written by hand (hipcc does not order a ds_read behind an LDS-DMA request).
// One instruction moves 1 KiB: lanes 0..31 fetch the wave's 64 values (two per lane) of table i, lanes 32..63 those of
// table i+1; LDS destination = M0 base + 16 * lane, i.e. table i at doubles [0, 64), table i+1 at [64, 128).
// (A divergent `if (lane < 32)` for an odd last table would split the basic block, and hipcc then sinks the slots'
// arithmetic below all eight fetches: 351 spilled VGPRs.)
typedef __attribute__((address_space(1))) const void lc_gptr;
typedef __attribute__((address_space(3))) void lc_lptr;
// `ubase` is wave-uniform (table FIRST of this prime, slot r), `voff` the lane's byte offset: scalar base + 32-bit
// vector offset keeps the eight slots' addresses out of the VGPRs.
template <int NT>
__device__ __forceinline__ void stage_consts(const char *ubase, u32 voff, size_t cstride_bytes, double *stg, int lane) {
#pragma unroll
    for (int i0 = 0; i0 < NT; i0 += 2) {
        const int i = (i0 + 1 < NT) ? i0 : NT - 2;      // odd count: the last instruction fetches tables NT-2 (again) and NT-1 --
        const char *b = ubase + (size_t)i * cstride_bytes;      // no divergent branch, no spare table, no extra LDS
        __builtin_amdgcn_global_load_lds((lc_gptr *)(b + voff), (lc_lptr *)(stg + i * 64), 16, 0, 0);
    }
}

template <int L, int LE, bool BIG, int HALF, bool PACK, bool LDSC>
__device__ __forceinline__ void rows_body(const u64 *__restrict__ in, double *__restrict__ mid, const double *__restrict__ consts,
                                          const double *__restrict__ tw, const Work &wk, double p, double pinv, u32 k, double *lds) {
    using SH = Shape<L, LE>;
    constexpr int N = SH::N, TP = SH::TP, E = SH::E, NC = HalfC<HALF>::NC, FIRST = HalfC<HALF>::FIRST;
    const int tid = threadIdx.x;
    const size_t poly_words = (size_t)k * N, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + 8 * wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * N;
    double w0[E - 1];
    load_tw<L, LE, 0>(w0, tw, tid);
    double x[4][E];
    // d_m +- d_(7-m) in integers (inputs are < 2^47), then one exact move to double.  Two line
    // pairs are kept in flight; the compiler fences stop it from hoisting every load to the top.
    constexpr u64 OFF = 1ULL << 48;
    u64 ra[2][E], rb[2][E];
    auto issue = [&](int m) {
        const u64 *a = in + base + (size_t)m * ct_words + tid, *b = in + base + (size_t)(7 - m) * ct_words + tid;
#pragma unroll
        for (int r = 0; r < E; r++) {       // pass-0 mapping: coefficient r*TP + tid
            ra[m & 1][r] = a[r * TP];
            rb[m & 1][r] = b[r * TP];
        }
    };
    auto combine = [&](int m) {
#pragma unroll
        for (int r = 0; r < E; r++) {
            const u64 A = ra[m & 1][r], B = rb[m & 1][r];
            x[m][r] = HALF ? u52_to_f64(A + OFF - B) - (double)OFF : u52_to_f64(A + B);
        }
    };
    if constexpr (LE >= 4) {
        issue(0);
        issue(1);
        asm volatile("" ::: "memory");
        combine(0);
        issue(2);
        asm volatile("" ::: "memory");
        combine(1);
        issue(3);
        asm volatile("" ::: "memory");
        combine(2);
        combine(3);
    } else {   // four waves per SIMD hide the round trips; keep the staging registers small
#pragma unroll
        for (int m = 0; m < 4; m++) {
            issue(m);
            combine(m);
            asm volatile("" ::: "memory");
        }
    }
    const double *cp = consts + (size_t)FIRST * k * N + (size_t)wk.prime * N + tid;
    const size_t cstride = (size_t)k * N;
    double cn[9];
    auto fetch = [&](int r) {
#pragma unroll
        for (int i = 0; i < NC; i++) cn[i] = cp[(size_t)i * cstride + r * TP];
    };
    int phase = 0;
    ntt_fwd<L, LE, 4>(x, w0, tw, p, pinv, lds, tid, phase, [&] { if (LE >= 4) fetch(0); });
    // LDSC: two staging buffers of NC x 64 doubles per wave (9 x 512 B x 2 x 8 waves = the two exchange buffers exactly)
    const int lane = tid & 63;
    double *stg = lds + (tid >> 6) * (2 * NC * 64);
    const char *sbase = (const char *)(consts + (size_t)FIRST * cstride + (size_t)wk.prime * N);
    const u32 svoff = (u32)(((size_t)(lane >> 5) * cstride + (tid & ~63) + 2 * (lane & 31)) * sizeof(double));
    constexpr int NDMA = (NC + 1) / 2;      // instructions per slot
    if constexpr (LDSC) {
        __syncthreads();                    // every wave is done with the exchange buffers
        stage_consts<NC>(sbase, svoff, cstride * sizeof(double), stg, lane);
    }
#pragma unroll
    for (int r = 0; r < E; r++) {
        double c[9];
        if constexpr (LDSC) {
            // slot r+1 on its way into the other buffer (last read in slot r-1) while slot r is evaluated
            if (r + 1 < E) stage_consts<NC>(sbase + (size_t)(r + 1) * TP * sizeof(double), svoff, cstride * sizeof(double), stg + ((r + 1) & 1) * NC * 64, lane);
            if (r + 1 < E) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const double *sr = stg + (r & 1) * NC * 64 + lane;
#pragma unroll
            for (int i = 0; i < NC; i++) c[i] = sr[i * 64];
        } else {
        if (LE < 4) fetch(r);               // four waves per SIMD: no software prefetch, fewer registers
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = cn[i];
        if (LE >= 4 && r + 1 < E) fetch(r + 1);
        }
        line_half<HALF>(x[0][r], x[1][r], x[2][r], x[3][r], c, p, pinv);
        if (BIG) {
#pragma unroll
            for (int m = 0; m < 4; m++) x[m][r] = red(x[m][r], p, pinv);
        } else if (PACK && HALF == 0) {   // outputs 0 and 4 carry no product: bring them below 2^39 too
            x[0][r] = red(x[0][r], p, pinv);
            x[2][r] = red(x[2][r], p, pinv);
        }
        // pin the slot's arithmetic between its own constant reads and the next slot's fetch: without a consumer here
        // hipcc orders all eight fetch / wait / read groups first and the arithmetic after them (constants spilled)
        if constexpr (LDSC) asm volatile("" ::"v"(x[0][r]), "v"(x[1][r]), "v"(x[2][r]), "v"(x[3][r]) : "memory");
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        double *o = mid + base + (size_t)(2 * m + HALF) * ct_words + tid;
        if constexpr (PACK) {
            store_packed<E, TP>((u64 *)o, x[m]);
        } else {
#pragma unroll
            for (int r = 0; r < E; r++) o[r * TP] = x[m][r];
        }
    }
}

// waves per SIMD requested from the register allocator = what LDS lets be resident (at most two workgroups)
__host__ __device__ constexpr int occ_waves(int tp, int lds_words) {
    return ((2 * lds_words * 16 <= 160 * 1024) ? 2 : 1) * tp / 256 < 1 ? 1 : ((2 * lds_words * 16 <= 160 * 1024) ? 2 : 1) * tp / 256;
}
template <int L, int LE> struct Occ { static constexpr int W = occ_waves(Shape<L, LE>::TP, Shape<L, LE>::LDS_WORDS); };

template <int L, int LE, bool BIG, bool PACK, bool LDSC>
__global__ __launch_bounds__((Shape<L, LE>::TP), (Occ<L, LE>::W)) void k_dct_rows(const u64 *__restrict__ in, double *__restrict__ mid,
                                                                  const double *__restrict__ consts, const double *__restrict__ tw_all,
                                                                  const Modulus *__restrict__ mods, u32 k) {
    __shared__ double lds[2 * Shape<L, LE>::LDS_WORDS];
    const Work wk = decode(blockIdx.x, k);
    const double p = (double)mods[wk.prime].q, pinv = 1.0 / p;
    const double *tw = tw_all + (size_t)wk.prime * Shape<L, LE>::N;
    if (wk.half) rows_body<L, LE, BIG, 1, PACK, LDSC>(in, mid, consts, tw, wk, p, pinv, k, lds);
    else rows_body<L, LE, BIG, 0, PACK, LDSC>(in, mid, consts, tw, wk, p, pinv, k, lds);
}

template <int L, int LE, bool BIG, int HALF, bool PACK>
__device__ __forceinline__ void cols_body(const double *__restrict__ mid, u64 *__restrict__ out, const double *__restrict__ consts,
                                          const double *__restrict__ itw, const Work &wk, double p, double pinv, u32 k, double *lds) {
    using SH = Shape<L, LE>;
    constexpr int N = SH::N, TP = SH::TP, E = SH::E, NC = HalfC<HALF>::NC, FIRST = HalfC<HALF>::FIRST;
    constexpr int LASTP = SH::NP - 1;
    const int tid = threadIdx.x;
    const size_t poly_words = (size_t)k * N, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * N;
    const size_t row_stride = 8 * ct_words;
    const size_t cstride = (size_t)k * N;
    const double *cp = consts + (size_t)FIRST * cstride + (size_t)wk.prime * N + tid;
    // per-output scale: row 2m+HALF, column wk.line -> constant 12 + 8*row + col
    const double *sp = consts + (size_t)(12 + 8 * HALF + wk.line) * cstride + (size_t)wk.prime * N + tid;
    double cn[9], sn[4];
    auto fetch = [&](int r) {
#pragma unroll
        for (int i = 0; i < NC; i++) cn[i] = cp[(size_t)i * cstride + r * TP];
#pragma unroll
        for (int m = 0; m < 4; m++) sn[m] = sp[(size_t)(16 * m) * cstride + r * TP];
    };
    double wl[E - 1];
    if constexpr (!PACK) {           // in flight behind the bulk loads
        fetch(0);
        load_tw<L, LE, LASTP>(wl, itw, tid);
    }
    double x[4][E];
    if constexpr (PACK) {
        // rebuild the biased doubles t = 2^52 + 2^51 + v from (0x43380000 + sign-extended byte : low
        // word) and let the first butterfly remove the bias.  All 80 packed words are requested
        // before the first is used (x is not live yet).
        u64 wa[4][5], wb[4][5];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const u64 *a = (const u64 *)(mid + base + (size_t)m * row_stride) + tid, *b = (const u64 *)(mid + base + (size_t)(7 - m) * row_stride) + tid;
#pragma unroll
            for (int q = 0; q < 5; q++) { wa[m][q] = a[q * TP]; wb[m][q] = b[q * TP]; }
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
            for (int r = 0; r < E; r++) {
                const u32 la = (r & 1) ? (u32)(wa[m][r >> 1] >> 32) : (u32)wa[m][r >> 1];
                const u32 lb = (r & 1) ? (u32)(wb[m][r >> 1] >> 32) : (u32)wb[m][r >> 1];
                const u32 hwa = (r & 4) ? (u32)(wa[m][4] >> 32) : (u32)wa[m][4], hwb = (r & 4) ? (u32)(wb[m][4] >> 32) : (u32)wb[m][4];
                const double ta = __hiloint2double((int)(0x43380000u + (u32)(int)(signed char)(hwa >> (8 * (r & 3)))), (int)la);
                const double tb = __hiloint2double((int)(0x43380000u + (u32)(int)(signed char)(hwb >> (8 * (r & 3)))), (int)lb);
                x[m][r] = HALF ? ta - tb : ta + (tb - 2.0 * PACK_BIAS);
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const double *a = mid + base + (size_t)m * row_stride + tid, *b = mid + base + (size_t)(7 - m) * row_stride + tid;
#pragma unroll
            for (int r = 0; r < E; r++) {
                const double A = a[r * TP], B = b[r * TP];
                x[m][r] = HALF ? A - B : A + B;
            }
        }
    }
    if constexpr (PACK) load_tw<L, LE, LASTP>(wl, itw, tid);   // the unpacking needs the registers first
#pragma unroll
    for (int r = 0; r < E; r++) {
        double c[9], sc[4];
        if (PACK) fetch(r);
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = cn[i];
#pragma unroll
        for (int m = 0; m < 4; m++) sc[m] = sn[m];
        if (!PACK && r + 1 < E) fetch(r + 1);
        line_half<HALF>(x[0][r], x[1][r], x[2][r], x[3][r], c, p, pinv);
        double y[4] = {x[0][r], x[1][r], x[2][r], x[3][r]};
        mmv<4>(y, sc, p, pinv);
#pragma unroll
        for (int m = 0; m < 4; m++) x[m][r] = y[m];
    }
    int phase = 0;
    ntt_inv<L, LE, 4, BIG>(x, wl, itw, p, pinv, lds, tid, phase);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        u64 *o = out + base + (size_t)(2 * m + HALF) * row_stride + tid;
#pragma unroll
        for (int r = 0; r < E; r++) {
            double v = x[m][r];
            v = v < 0.0 ? v + p : v;
            o[r * TP] = f64_to_u52(v);
        }
    }
}

template <int L, int LE, bool BIG, bool PACK>
__global__ __launch_bounds__((Shape<L, LE>::TP), (Occ<L, LE>::W)) void k_dct_cols(const double *__restrict__ mid, u64 *__restrict__ out,
                                                                  const double *__restrict__ consts, const double *__restrict__ itw_all,
                                                                  const Modulus *__restrict__ mods, u32 k) {
    __shared__ double lds[2 * Shape<L, LE>::LDS_WORDS];
    const Work wk = decode(blockIdx.x, k);   // line = column index
    const double p = (double)mods[wk.prime].q, pinv = 1.0 / p;
    const double *itw = itw_all + (size_t)wk.prime * Shape<L, LE>::N;
    if (wk.half) cols_body<L, LE, BIG, 1, PACK>(mid, out, consts, itw, wk, p, pinv, k, lds);
    else cols_body<L, LE, BIG, 0, PACK>(mid, out, consts, itw, wk, p, pinv, k, lds);
}

// EXPERIMENT (round 4, FHE_DCT_ONE_LAUNCH=D; off by default): rows and columns in ONE launch.  A unit is (prime, block,
// polynomial): 16 row workgroups write its 1.25 MB packed intermediate, 16 column workgroups read it.  All 32 sit on one
// XCD (blockIdx = 8 * sequence + xcd) and the XCD's sequence alternates [16 row workgroups of unit j + D][16 column
// workgroups of unit j], so a unit's intermediate is consumed D units after it was produced -- while it is still in that
// XCD's L2 / the Infinity Cache -- instead of a whole 256-block wave (3 GiB) later.  Workgroups are dispatched in blockIdx
// order, so the rows a column workgroup waits for were dispatched before it and never wait themselves: no deadlock; the
// wait is a bounded spin on a per-unit arrival counter (release: __threadfence + atomicAdd by every row workgroup;
// acquire: atomic load + __threadfence), and a spin that runs out raises *err instead of hanging the device.
template <int L, int LE>
__global__ __launch_bounds__((Shape<L, LE>::TP), (Occ<L, LE>::W)) void k_dct_one_launch(const u64 *__restrict__ in, double *__restrict__ mid, u64 *__restrict__ out,
                                                                        const double *__restrict__ consts, const double *__restrict__ tw_all,
                                                                        const double *__restrict__ itw_all, const Modulus *__restrict__ mods, u32 k,
                                                                        u32 n_blocks, u32 dist, u32 *__restrict__ arrived, u32 *__restrict__ err) {
    __shared__ double lds[2 * Shape<L, LE>::LDS_WORDS];
    const u32 xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, chunk = seq >> 5, r = seq & 31;
    const bool is_row = r < 16;
    const long j = is_row ? (long)chunk : (long)chunk - (long)(dist & 0xffffu);
    const u32 n_units = n_blocks * 2 * k;
    if (j < 0) return;
    const u64 g = (u64)j * 8 + xcd;
    if (g >= n_units) return;
    Work wk;
    wk.prime = (u32)(g / (n_blocks * 2));
    const u32 rem = (u32)(g - (u64)wk.prime * n_blocks * 2);
    wk.blk = rem >> 1;
    wk.poly = rem & 1;
    wk.line = (r & 15) >> 1;
    wk.half = r & 1;
    const double p = (double)mods[wk.prime].q, pinv = 1.0 / p;
    if (is_row) {
        const double *tw = tw_all + (size_t)wk.prime * Shape<L, LE>::N;
        if (wk.half) rows_body<L, LE, false, 1, true, true>(in, mid, consts, tw, wk, p, pinv, k, lds);
        else rows_body<L, LE, false, 0, true, true>(in, mid, consts, tw, wk, p, pinv, k, lds);
        if (dist & 0x10000u) return;                       // timing-only switch of the experiment: no hand-over at all (results invalid)
        __syncthreads();                                   // every thread's stores are issued ...
        if (threadIdx.x == 0) {
            __threadfence();                               // ... and visible device-wide before the unit counts this workgroup
            atomicAdd(arrived + g, 1u);
        }
    } else {
        if (!(dist & 0x10000u) && threadIdx.x == 0) {
            u32 spins = 0;
            while (__hip_atomic_load(arrived + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 24)) { atomicExch(err, 1u); break; }
            }
            __threadfence();
        }
        __syncthreads();
        const double *itw = itw_all + (size_t)wk.prime * Shape<L, LE>::N;
        if (wk.half) cols_body<L, LE, false, 1, true>(mid, out, consts, itw, wk, p, pinv, k, lds);
        else cols_body<L, LE, false, 0, true>(mid, out, consts, itw, wk, p, pinv, k, lds);
    }
}

// Shoup-pair table in the u64 kernels' slot order (16 slots per thread) -> centred doubles in the
// fused kernels' slot order: bit-reversed index j = (t << LE) + r lives at r * (n >> LE) + t.
__global__ void k_consts_to_f64(const ulonglong2 *__restrict__ in, double *__restrict__ out, const Modulus *__restrict__ mods, u32 k, u32 n, u32 le, u32 total) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u32 pos = i % n, rowbase = i - pos;          // destination position within its [n] row
    const u32 tp = n >> le, r = pos / tp, t = pos % tp;
    const u32 j = (t << le) + r;
    const u32 src = (j & 15) * (n >> 4) + (j >> 4);
    const u64 q = mods[(i / n) % k].q;
    const u64 w = in[rowbase + src].x;
    out[i] = w > q / 2 ? -(double)(q - w) : (double)w;
}

// rgb_to_ycc_fhe (homo/fhe_image.h:310-325) for one residue polynomial of one pixel: three joint
// forward transforms, the nine constant products per slot, three joint inverse transforms, in
// place.  No intermediate leaves the workgroup: global traffic is the compulsory 3 in + 3 out.
template <int L, int LE>
__global__ __launch_bounds__((Shape<L, LE>::TP), (Occ<L, LE>::W)) void k_rgb2ycc_f64(u64 *__restrict__ R, u64 *__restrict__ G, u64 *__restrict__ Bc,
                                                                     const double *__restrict__ consts, const double *__restrict__ tw_all,
                                                                     const double *__restrict__ itw_all, const Modulus *__restrict__ mods,
                                                                     const u64 *__restrict__ yoff, u32 yoff_len, u32 k, u32 group, u64 gstride) {
    using SH = Shape<L, LE>;
    constexpr int N = SH::N, TP = SH::TP, E = SH::E, LASTP = SH::NP - 1;
    __shared__ double lds[2 * SH::LDS_WORDS];
    const int tid = threadIdx.x;
    // prime-major like the DCT kernels: resident workgroups share one prime's tables
    const u32 per_prime = gridDim.x / k;
    const u32 prime = blockIdx.x / per_prime;
    const u32 pp = blockIdx.x - prime * per_prime;           // pixel * 2 + poly
    const u32 poly = pp & 1, pix = pp >> 1;
    const u64 q = mods[prime].q;
    const double p = (double)q, pinv = 1.0 / p;
    const double *tw = tw_all + (size_t)prime * N, *itw = itw_all + (size_t)prime * N;
    // pixel `pix` of a plane: contiguous (group == 0), or `group` pixels every gstride words (the block layout of the
    // ciphertext streams, [block][R G B][64]: fhe_rgb_to_ycc_blocks)
    const size_t pix_off = group ? (size_t)(pix / group) * gstride + (size_t)(pix % group) * 2 * k * N : (size_t)pix * 2 * k * N;
    const size_t off = pix_off + ((size_t)poly * k + prime) * N + tid;
    double w0[E - 1];
    load_tw<L, LE, 0>(w0, tw, tid);
    double x[3][E];
    {
        const u64 *src[3] = {R + off, G + off, Bc + off};
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int r = 0; r < E; r++) x[m][r] = u52_to_f64(src[m][r * TP]);
    }
    int phase = 0;
    ntt_fwd<L, LE, 3>(x, w0, tw, p, pinv, lds, tid, phase, [] {});
    double wl[E - 1];
    load_tw<L, LE, LASTP>(wl, itw, tid);
    const double *cp = consts + (size_t)prime * N + tid;
    const size_t cstride = (size_t)k * N;
#pragma unroll
    for (int r = 0; r < E; r++) {
        double c[9], y[9];
#pragma unroll
        for (int i = 0; i < 9; i++) { c[i] = cp[(size_t)i * cstride + r * TP]; y[i] = x[i % 3][r]; }
        mmv<9>(y, c, p, pinv);
        x[0][r] = y[0] + y[1] + y[2];          // Y  =  0.299 R + 0.587 G + 0.114 B      (- 128 below)
        x[1][r] = y[3] - y[4] + y[5];          // Cb = (-0.168736) R - 0.331264 G + 0.5 B
        x[2][r] = y[6] - y[7] - y[8];          // Cr =  0.5 R - 0.418688 G - 0.081312 B
    }
    ntt_inv<L, LE, 3, false>(x, wl, itw, p, pinv, lds, tid, phase);
    u64 *dst[3] = {R + off, G + off, Bc + off};
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int r = 0; r < E; r++) {
            double v = x[m][r];
            v = v < 0.0 ? v + p : v;
            u64 o = f64_to_u52(v);
            if (m == 0 && poly == 0) {
                const u32 j = (u32)(r * TP + tid);
                if (j < yoff_len) o = submod(o, yoff[(size_t)prime * yoff_len + j], q);
            }
            dst[m][r * TP] = o;
        }
}

// Standalone transforms and multiply_plain on the FP64 machinery, reading and writing the library's
// NTT-form order (the u64 kernels' 16-slots-per-thread order, include/fhe_hip.h layout note):
// bit-reversed index j = (t << LE) + r lives at (j & 15) * (n / 16) + (j >> 4).
// MODE 0: forward NTT; 1: inverse NTT; 2: multiply_plain (NTT, product with the prepared plaintext,
// inverse NTT).  M polynomials of one prime per workgroup share every twiddle.
template <int L, int LE, int M, int MODE>
__global__ __launch_bounds__((Shape<L, LE>::TP), (Occ<L, LE>::W)) void k_poly_f64(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                                  const ulonglong2 *__restrict__ plain, const double *__restrict__ tw_all,
                                                                  const double *__restrict__ itw_all, const Modulus *__restrict__ mods, u32 k) {
    using SH = Shape<L, LE>;
    constexpr int N = SH::N, TP = SH::TP, E = SH::E, LASTP = SH::NP - 1;
    static_assert(LE == 3, "slot-order mapping below is written for 8 values per thread");
    __shared__ double lds[2 * SH::LDS_WORDS];
    const int tid = threadIdx.x;
    const u32 per_prime = gridDim.x / k;
    const u32 prime = blockIdx.x / per_prime;
    const u32 g = blockIdx.x - prime * per_prime;
    const double p = (double)mods[prime].q, pinv = 1.0 / p;
    const double *tw = tw_all + (size_t)prime * N, *itw = itw_all + (size_t)prime * N;
    const int slot0 = ((tid & 1) << 3) * (N >> 4) + (tid >> 1);        // + r * (N >> 4)
    double x[M][E];
    int phase = 0;
    if constexpr (MODE != 1) {
        double w0[E - 1];
        load_tw<L, LE, 0>(w0, tw, tid);
#pragma unroll
        for (int m = 0; m < M; m++) {
            const u64 *src = in + ((size_t)(g * M + m) * k + prime) * N + tid;
#pragma unroll
            for (int r = 0; r < E; r++) x[m][r] = u52_to_f64(src[r * TP]);
        }
        ntt_fwd<L, LE, M>(x, w0, tw, p, pinv, lds, tid, phase, [] {});
    } else {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const u64 *src = in + ((size_t)(g * M + m) * k + prime) * N + slot0;
#pragma unroll
            for (int r = 0; r < E; r++) x[m][r] = u52_to_f64(src[r * (N >> 4)]);
        }
    }
    if constexpr (MODE == 0) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            u64 *dst = out + ((size_t)(g * M + m) * k + prime) * N + slot0;
#pragma unroll
            for (int r = 0; r < E; r++) {
                double v = red(x[m][r], p, pinv);
                v = v < 0.0 ? v + p : v;
                dst[r * (N >> 4)] = f64_to_u52(v);
            }
        }
        return;
    }
    double wl[E - 1];
    load_tw<L, LE, LASTP>(wl, itw, tid);
    if constexpr (MODE == 2) {
        const ulonglong2 *pp = plain + (size_t)prime * N + slot0;
#pragma unroll
        for (int r = 0; r < E; r++) {
            const double w = u52_to_f64(pp[r * (N >> 4)].x);
            double y[M], wm[M];
#pragma unroll
            for (int m = 0; m < M; m++) { y[m] = x[m][r]; wm[m] = w; }
            mmv<M>(y, wm, p, pinv);
#pragma unroll
            for (int m = 0; m < M; m++) x[m][r] = y[m];
        }
    }
    ntt_inv<L, LE, M, false>(x, wl, itw, p, pinv, lds, tid, phase);
#pragma unroll
    for (int m = 0; m < M; m++) {
        u64 *dst = out + ((size_t)(g * M + m) * k + prime) * N + tid;
#pragma unroll
        for (int r = 0; r < E; r++) {
            double v = x[m][r];
            v = v < 0.0 ? v + p : v;
            dst[r * TP] = f64_to_u52(v);
        }
    }
}

}  // namespace


// coefficients per thread = 2^LE of the fused pair for this context: 8 (four waves per SIMD) at n = 4096 with primes
// <= 40 bits and at n = 8192 (1024-thread workgroups, one per CU); 16 otherwise
static u32 dct_shape_le(const fhe_ctx *c) {
    if (c->opt.dct_le != 3) return 4;      // coefficients per thread = 2^LE; 3 keeps four waves per SIMD resident
    if (c->logn == 12 && c->max_prime_bits <= 40) return 3;
    if (c->logn == 13 || c->logn == 11) return 3;
    return 4;
}

bool fhe_dct_f64_supported(const fhe_ctx *c) {
    return c && c->qb.d_tw_f64 && c->max_prime_bits <= 47 && (c->logn >= 10 && c->logn <= 13);
}

int fhe_dct_f64_make_consts(const fhe_ctx *c, fhe_dct_plan *plan, hipStream_t st) {
    const u32 total = DCT_NCONST * c->k * c->n;
    HIP_TRY(hipMalloc(&plan->d_consts_f64, sizeof(double) * total));
    const u32 le = dct_shape_le(c);
    k_consts_to_f64<<<(total + 255) / 256, 256, 0, st>>>(plan->d_consts, plan->d_consts_f64, c->qb.d_mod, c->k, c->n, le, total);
    KERNEL_CHECK();
    return FHE_OK;
}


// row kernel: circuit constants through LDS (stage_consts); FheOptions::dct_ldsc = false reads them into registers as before.
// (The column kernel has four more constants per slot and no register to spare: the same staging spills 36-56 VGPRs
// there and measured 68 k blocks/s against 80 k, so it keeps its loads.)

template <int L, int LE>
static void launch_pair(const fhe_ctx *c, const fhe_dct_plan *plan, const u64 *in, u64 *out, double *mid, unsigned grid, bool big, hipStream_t st, int which) {
    constexpr int TP = Shape<L, LE>::TP;
    if constexpr (LE == 4) {
        if (big) {
            if (which & 1) k_dct_rows<L, LE, true, false, false><<<grid, TP, 0, st>>>(in, mid, plan->d_consts_f64, c->qb.d_tw_f64, c->qb.d_mod, c->k);
            if (which & 2) k_dct_cols<L, LE, true, false><<<grid, TP, 0, st>>>(mid, out, plan->d_consts_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k);
            return;
        }
    }
    if constexpr (LE == 3) {
        if (big) {
            if (which & 1) k_dct_rows<L, LE, true, false, false><<<grid, TP, 0, st>>>(in, mid, plan->d_consts_f64, c->qb.d_tw_f64, c->qb.d_mod, c->k);
            if (which & 2) k_dct_cols<L, LE, true, false><<<grid, TP, 0, st>>>(mid, out, plan->d_consts_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k);
            return;
        }
        if (c->max_prime_bits <= 37 && c->opt.dct_pack) {      // packed intermediate, 40 instead of 64 bytes
            if (which & 1) {
                if (c->opt.dct_ldsc) k_dct_rows<L, LE, false, true, true><<<grid, TP, 0, st>>>(in, mid, plan->d_consts_f64, c->qb.d_tw_f64, c->qb.d_mod, c->k);
                else k_dct_rows<L, LE, false, true, false><<<grid, TP, 0, st>>>(in, mid, plan->d_consts_f64, c->qb.d_tw_f64, c->qb.d_mod, c->k);
            }
            if (which & 2) k_dct_cols<L, LE, false, true><<<grid, TP, 0, st>>>(mid, out, plan->d_consts_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k);
            return;
        }
    }
    {
        if (which & 1) k_dct_rows<L, LE, false, false, false><<<grid, TP, 0, st>>>(in, mid, plan->d_consts_f64, c->qb.d_tw_f64, c->qb.d_mod, c->k);
        if (which & 2) k_dct_cols<L, LE, false, false><<<grid, TP, 0, st>>>(mid, out, plan->d_consts_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k);
    }
}

int fhe_dct_f64_launch(const fhe_ctx *c, const fhe_dct_plan *plan, const u64 *in, u64 *out, u64 n_blocks, double *mid, hipStream_t st, int which) {
    const u64 items = n_blocks * 8 * 2 * c->k;   // (block, line, poly, prime), multiple of 8
    const u64 grid = items * 2;
    if (grid > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many blocks for one launch");
    const bool big = c->max_prime_bits > 40;
    if (c->opt.dct_one_launch && which == 3 && c->logn == 12 && dct_shape_le(c) == 3 && c->max_prime_bits <= 37 && c->opt.dct_pack && c->d_arrived) {
        const u64 n_units = n_blocks * 2 * c->k;
        if (n_units > c->arrived_cap) return fail(FHE_ERR_PARAM, "one-launch experiment: wave too large for the arrival counters");
        HIP_TRY(hipMemsetAsync(c->d_arrived, 0, (n_units + 1) * sizeof(u32), st));
        const u32 dist = c->opt.dct_one_launch;
        const u64 chunks = (n_units + 7) / 8 + (dist & 0xffffu);
        k_dct_one_launch<12, 3><<<(unsigned)(chunks * 32 * 8), Shape<12, 3>::TP, 0, st>>>(in, mid, out, plan->d_consts_f64, c->qb.d_tw_f64, c->qb.d_itw_f64, c->qb.d_mod,
                                                                                       c->k, (u32)n_blocks, dist, c->d_arrived + 1, c->d_arrived);
        KERNEL_CHECK();
        // EXPERIMENT path (FHE_DCT_ONE_LAUNCH; profiles/EXPERIMENTS.md section 1; never a default, never in a parity path other than its
        // own test): a column workgroup that gives up waiting for its rows sets d_arrived[0] and goes on with an incomplete
        // intermediate, so the flag is read back before the call reports success.  That makes this path synchronous -- acceptable
        // for a measurement switch, and the only way the caller cannot receive a silently wrong result.  (The no-deadlock argument
        // also rests on workgroups being dispatched in blockIdx order, which HIP does not promise: one more reason it stays off.)
        u32 timed_out = 0;
        HIP_TRY(hipMemcpyAsync(&timed_out, c->d_arrived, sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (timed_out) return fail(FHE_ERR_HIP, "one-launch experiment: a column workgroup timed out waiting for its row transforms; the output is incomplete");
        return FHE_OK;
    }
    switch (c->logn) {   // LE = 3 is only built for the headline size
        case 10: launch_pair<10, 4>(c, plan, in, out, mid, (unsigned)grid, big, st, which); break;
        case 11: launch_pair<11, 3>(c, plan, in, out, mid, (unsigned)grid, big, st, which); break;
        case 12: if (dct_shape_le(c) == 3) launch_pair<12, 3>(c, plan, in, out, mid, (unsigned)grid, false, st, which); else launch_pair<12, 4>(c, plan, in, out, mid, (unsigned)grid, big, st, which); break;
        case 13: if (dct_shape_le(c) == 3) launch_pair<13, 3>(c, plan, in, out, mid, (unsigned)grid, big, st, which); else launch_pair<13, 4>(c, plan, in, out, mid, (unsigned)grid, big, st, which); break;
        default: return fail(FHE_ERR_PARAM, "fused FP64 path supports n in {1024, 2048, 4096, 8192}");
    }
    KERNEL_CHECK();
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// rgb_to_ycc_fhe on the FP64 kernels (n = 4096, primes <= 40 bits: the P4096 preset)
// ------------------------------------------------------------------------------------------------
bool fhe_rgb_f64_supported(const fhe_ctx *c) {
    return fhe_dct_f64_supported(c) && c->logn == 12 && c->max_prime_bits <= 40;
}

int fhe_rgb_f64_make_consts(const fhe_ctx *c, const ulonglong2 *d_c, double **out, hipStream_t st) {
    const u32 total = 9 * c->k * c->n;
    HIP_TRY(hipMalloc(out, sizeof(double) * total));
    k_consts_to_f64<<<(total + 255) / 256, 256, 0, st>>>(d_c, *out, c->qb.d_mod, c->k, c->n, 3, total);
    KERNEL_CHECK();
    return FHE_OK;
}

int fhe_rgb_f64_launch(const fhe_ctx *c, u64 *r, u64 *g, u64 *b, u64 count, const double *consts, const u64 *yoff, u32 yoff_len, hipStream_t st, u32 group, u64 gstride) {
    const u64 grid = count * 2 * c->k;
    if (grid > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many pixels for one launch");
    k_rgb2ycc_f64<12, 3><<<(unsigned)grid, Shape<12, 3>::TP, 0, st>>>(r, g, b, consts, c->qb.d_tw_f64, c->qb.d_itw_f64, c->qb.d_mod, yoff, yoff_len, c->k, group, gstride);
    KERNEL_CHECK();
    return FHE_OK;
}

// mode 0 forward NTT, 1 inverse NTT, 2 multiply_plain; n_polys RNS polynomials of k residues each
int fhe_poly_f64_launch(int mode, const fhe_ctx *c, const u64 *in, u64 *out, u64 n_polys, const ulonglong2 *plain, hipStream_t st) {
    const u64 grid2 = (n_polys / 2) * c->k, grid1 = n_polys * c->k;
    if (grid1 > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    constexpr int TP = Shape<12, 3>::TP;
#define LAUNCH_POLY(M, GRID)                                                                                                          \
    switch (mode) {                                                                                                                   \
        case 0: k_poly_f64<12, 3, M, 0><<<(unsigned)(GRID), TP, 0, st>>>(in, out, plain, c->qb.d_tw_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k); break; \
        case 1: k_poly_f64<12, 3, M, 1><<<(unsigned)(GRID), TP, 0, st>>>(in, out, plain, c->qb.d_tw_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k); break; \
        default: k_poly_f64<12, 3, M, 2><<<(unsigned)(GRID), TP, 0, st>>>(in, out, plain, c->qb.d_tw_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k); break; \
    }
    if (n_polys % 2 == 0) { LAUNCH_POLY(2, grid2) } else { LAUNCH_POLY(1, grid1) }
#undef LAUNCH_POLY
    KERNEL_CHECK();
    return FHE_OK;
}
