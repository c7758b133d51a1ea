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
//                       LL&M row pass per NTT slot, store the four row outputs (NTT form, FP64)
//   kernel B (columns): same for columns on the NTT-form intermediates, per-output scale
//                       encode(0.125)*encode(1/quant) folded into one product, inverse NTT, store.
// Each workgroup (256 threads at n=4096) carries 4 polynomials x 16 coefficients per thread in
// registers and shares every twiddle across the four; the even and odd workgroup of a line read the
// same eight inputs and are placed on the same XCD (blockIdx = 16g+u and 16g+8+u) so the second
// read is an L2 hit.  All 832 Evaluator calls of the reference are exact ring operations, so this
// evaluation order yields bit-identical ciphertexts.
#include "internal.h"

#pragma clang fp contract(off)

namespace {

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

// twiddles of one register pass, fetched ahead of use (before the preceding LDS barrier) so that
// their L2 latency is hidden behind the previous pass: 2^(3-rb) values per stage, at most 15.
template <int L, int P> struct PassTw {
    static constexpr int LO = pass_lo(L, P), S = pass_stages(L, P);
    static constexpr int rb(int u) { return (L - 1 - (4 * P + u)) - LO; }
    static constexpr int count(int u) { return 1 << (3 - rb(u)); }
    static constexpr int offset(int u) { int o = 0; for (int v = 0; v < u; v++) o += count(v); return o; }
};
template <int L, int P>
__device__ __forceinline__ void load_tw(double (&w)[15], const double *__restrict__ tw, int tid) {
    using T = PassTw<L, P>;
    const int th = (P == 0) ? 0 : (tid >> T::LO);
#pragma unroll
    for (int u = 0; u < T::S; u++) {
#pragma unroll
        for (int i = 0; i < T::count(u); i++) w[T::offset(u) + i] = tw[(1 << (4 * P + u)) + ((th << (3 - T::rb(u))) | i)];
    }
}

template <int L, int P, int M>
__device__ __forceinline__ void fwd_pass(double (&x)[M][16], const double (&w)[15], double p, double pinv) {
    using T = PassTw<L, P>;
#pragma unroll
    for (int u = 0; u < T::S; u++) {
        const int rb = T::rb(u);
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const double wv = w[T::offset(u) + (r0 >> (rb + 1))];
            double t[M], wm[M];
#pragma unroll
            for (int m = 0; m < M; m++) { t[m] = x[m][r1]; wm[m] = wv; }
            mmv<M>(t, wm, p, pinv);
#pragma unroll
            for (int m = 0; m < M; m++) {
                const double X = x[m][r0];
                x[m][r0] = X + t[m];
                x[m][r1] = X - t[m];
            }
        }
    }
}

template <int L, int P, int M>
__device__ __forceinline__ void inv_pass(double (&x)[M][16], const double (&w)[15], double ninv, double p, double pinv) {
    using T = PassTw<L, P>;
#pragma unroll
    for (int u = T::S - 1; u >= 0; u--) {
        const int sigma = 4 * P + u, rb = T::rb(u);
#pragma unroll
        for (int r0 = 0; r0 < 16; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            const double wv = w[T::offset(u) + (r0 >> (rb + 1))];
            if (sigma == 0) {
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
            } else {
                double t[M], wm[M];
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const double X = x[m][r0], Y = x[m][r1];
                    x[m][r0] = X + Y;
                    t[m] = X - Y; wm[m] = wv;
                }
                mmv<M>(t, wm, p, pinv);
#pragma unroll
                for (int m = 0; m < M; m++) x[m][r1] = t[m];
            }
        }
    }
}

// M polynomials through two alternating LDS buffers: one barrier per polynomial
template <int L, int LO_FROM, int LO_TO, int M>
__device__ __forceinline__ void transpose(double (&x)[M][16], double *lds, int tid, int &phase) {
    constexpr int PL = imin(LO_FROM, LO_TO);
#pragma unroll
    for (int m = 0; m < M; m++) {
        double *buf = lds + (phase & 1) * NttShape<L>::LDS_WORDS;
        phase++;
#pragma unroll
        for (int r = 0; r < 16; r++) buf[lds_pad<PL>(elem_index<LO_FROM>(tid, r))] = x[m][r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) x[m][r] = buf[lds_pad<PL>(elem_index<LO_TO>(tid, r))];
    }
}

// forward transform; `w` holds the pass-0 twiddles on entry.  `pre()` is invoked once, right
// before the last transpose, so that the caller can start fetching what it needs after the NTT.
template <int L, int M, typename PRE, int P = 0>
__device__ __forceinline__ void ntt_fwd(double (&x)[M][16], double (&w)[15], const double *__restrict__ tw, double p, double pinv,
                                        double *lds, int tid, int &phase, PRE pre) {
    if constexpr (P + 1 < NttShape<L>::NP) {
        double wn[15];
        load_tw<L, P + 1>(wn, tw, tid);          // in flight during this pass and the transpose
        fwd_pass<L, P, M>(x, w, p, pinv);
        if constexpr (P + 2 == NttShape<L>::NP) pre();
        transpose<L, pass_lo(L, P), pass_lo(L, P + 1), M>(x, lds, tid, phase);
        ntt_fwd<L, M, PRE, P + 1>(x, wn, tw, p, pinv, lds, tid, phase, pre);
    } else {
        fwd_pass<L, P, M>(x, w, p, pinv);
    }
}
template <int L, int M, bool BIG, int P = NttShape<L>::NP - 1>
__device__ __forceinline__ void ntt_inv(double (&x)[M][16], double (&w)[15], const double *__restrict__ itw, double p, double pinv,
                                        double *lds, int tid, int &phase) {
    if constexpr (P > 0) {
        double wn[15];
        load_tw<L, P - 1>(wn, itw, tid);
        inv_pass<L, P, M>(x, w, 0.0, p, pinv);
        if constexpr (BIG) {
#pragma unroll
            for (int m = 0; m < M; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) x[m][r] = red(x[m][r], p, pinv);
        }
        transpose<L, pass_lo(L, P), pass_lo(L, P - 1), M>(x, lds, tid, phase);
        ntt_inv<L, M, BIG, P - 1>(x, wn, itw, p, pinv, lds, tid, phase);
    } else {
        inv_pass<L, P, M>(x, w, itw[0], p, pinv);
    }
}

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
        double y[9] = {z3 + z4, tmp4, tmp5, tmp6, tmp7, z1, z2, z3, z4};
        const double w[9] = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]};
        mmv<9>(y, w, p, pinv);
        const double z3b = y[7] + y[0], z4b = y[8] + y[0];
        x0 = y[4] + y[5] + z4b;    // out 1 = tmp7' + z1' + z4
        x1 = y[3] + y[6] + z3b;    // out 3 = tmp6' + z2' + z3
        x2 = y[2] + y[6] + z4b;    // out 5 = tmp5' + z2' + z4
        x3 = y[1] + y[5] + z3b;    // out 7 = tmp4' + z1' + z3
    }
}
template <int HALF> struct HalfC { static constexpr int NC = HALF ? 9 : 3, FIRST = HALF ? 3 : 0; };

struct Work { u32 blk, line, poly, prime, half; };
// blockIdx -> work item; the two halves of an item sit 8 apart so they land on the same XCD
__device__ __forceinline__ Work decode(u32 idx, u32 k) {
    const u32 w = ((idx >> 4) << 3) | (idx & 7);
    Work o;
    o.half = (idx >> 3) & 1;
    o.prime = w % k;
    u32 t = w / k;
    o.poly = t & 1;
    t >>= 1;
    o.line = t & 7;
    o.blk = t >> 3;
    return o;
}

template <int L, bool BIG, int HALF>
__device__ __forceinline__ void rows_body(const u64 *__restrict__ in, double *__restrict__ mid, const double *__restrict__ consts,
                                          const double *__restrict__ tw, const Work &wk, double p, double pinv, u32 k, double *lds) {
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP, NC = HalfC<HALF>::NC, FIRST = HalfC<HALF>::FIRST;
    const int tid = threadIdx.x;
    const size_t poly_words = (size_t)k * N, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + 8 * wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * N;
    double w0[15];
    load_tw<L, 0>(w0, tw, tid);
    double x[4][16];
    // d_m +- d_(7-m) in integers (inputs are < 2^47), then one exact move to double.  Two line
    // pairs (64 loads per thread) are kept in flight; the compiler fences stop it from hoisting all
    // 128 loads to the top, which would not fit the register file.
    constexpr u64 OFF = 1ULL << 48;
    u64 ra[2][16], rb[2][16];
    auto issue = [&](int m) {
        const u64 *a = in + base + (size_t)m * ct_words + tid, *b = in + base + (size_t)(7 - m) * ct_words + tid;
#pragma unroll
        for (int r = 0; r < 16; r++) {       // pass-0 mapping: coefficient r*TP + tid
            ra[m & 1][r] = a[r * TP];
            rb[m & 1][r] = b[r * TP];
        }
    };
    auto combine = [&](int m) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u64 A = ra[m & 1][r], B = rb[m & 1][r];
            x[m][r] = HALF ? u52_to_f64(A + OFF - B) - (double)OFF : u52_to_f64(A + B);
        }
    };
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
    const double *cp = consts + (size_t)FIRST * k * N + (size_t)wk.prime * N + tid;
    const size_t cstride = (size_t)k * N;
    double cn[9];
    auto fetch = [&](int r) {
#pragma unroll
        for (int i = 0; i < NC; i++) cn[i] = cp[(size_t)i * cstride + r * TP];
    };
    int phase = 0;
    ntt_fwd<L, 4>(x, w0, tw, p, pinv, lds, tid, phase, [&] { fetch(0); });
#pragma unroll
    for (int r = 0; r < 16; r++) {
        double c[9];
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = cn[i];
        if (r + 1 < 16) fetch(r + 1);
        line_half<HALF>(x[0][r], x[1][r], x[2][r], x[3][r], c, p, pinv);
        if (BIG) {
#pragma unroll
            for (int m = 0; m < 4; m++) x[m][r] = red(x[m][r], p, pinv);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        double *o = mid + base + (size_t)(2 * m + HALF) * ct_words + tid;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r * TP] = x[m][r];
    }
}

template <int L, bool BIG>
__global__ __launch_bounds__(NttShape<L>::TP, 2) void k_dct_rows(const u64 *__restrict__ in, double *__restrict__ mid,
                                                                  const double *__restrict__ consts, const double *__restrict__ tw_all,
                                                                  const Modulus *__restrict__ mods, u32 k) {
    __shared__ double lds[2 * NttShape<L>::LDS_WORDS];
    const Work wk = decode(blockIdx.x, k);
    const double p = (double)mods[wk.prime].q, pinv = 1.0 / p;
    const double *tw = tw_all + (size_t)wk.prime * NttShape<L>::N;
    if (wk.half) rows_body<L, BIG, 1>(in, mid, consts, tw, wk, p, pinv, k, lds);
    else rows_body<L, BIG, 0>(in, mid, consts, tw, wk, p, pinv, k, lds);
}

template <int L, bool BIG, int HALF>
__device__ __forceinline__ void cols_body(const double *__restrict__ mid, u64 *__restrict__ out, const double *__restrict__ consts,
                                          const double *__restrict__ itw, const Work &wk, double p, double pinv, u32 k, double *lds) {
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP, NC = HalfC<HALF>::NC, FIRST = HalfC<HALF>::FIRST;
    constexpr int LASTP = NttShape<L>::NP - 1;
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
    fetch(0);
    double wl[15];
    load_tw<L, LASTP>(wl, itw, tid);
    double x[4][16];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const double *a = mid + base + (size_t)m * row_stride + tid, *b = mid + base + (size_t)(7 - m) * row_stride + tid;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const double A = a[r * TP], B = b[r * TP];
            x[m][r] = HALF ? A - B : A + B;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        double c[9], sc[4];
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = cn[i];
#pragma unroll
        for (int m = 0; m < 4; m++) sc[m] = sn[m];
        if (r + 1 < 16) fetch(r + 1);
        line_half<HALF>(x[0][r], x[1][r], x[2][r], x[3][r], c, p, pinv);
        double y[4] = {x[0][r], x[1][r], x[2][r], x[3][r]};
        mmv<4>(y, sc, p, pinv);
#pragma unroll
        for (int m = 0; m < 4; m++) x[m][r] = y[m];
    }
    int phase = 0;
    ntt_inv<L, 4, BIG>(x, wl, itw, p, pinv, lds, tid, phase);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        u64 *o = out + base + (size_t)(2 * m + HALF) * row_stride + tid;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            double v = x[m][r];
            v = v < 0.0 ? v + p : v;
            o[r * TP] = f64_to_u52(v);
        }
    }
}

template <int L, bool BIG>
__global__ __launch_bounds__(NttShape<L>::TP, 2) void k_dct_cols(const double *__restrict__ mid, u64 *__restrict__ out,
                                                                  const double *__restrict__ consts, const double *__restrict__ itw_all,
                                                                  const Modulus *__restrict__ mods, u32 k) {
    __shared__ double lds[2 * NttShape<L>::LDS_WORDS];
    const Work wk = decode(blockIdx.x, k);   // line = column index
    const double p = (double)mods[wk.prime].q, pinv = 1.0 / p;
    const double *itw = itw_all + (size_t)wk.prime * NttShape<L>::N;
    if (wk.half) cols_body<L, BIG, 1>(mid, out, consts, itw, wk, p, pinv, k, lds);
    else cols_body<L, BIG, 0>(mid, out, consts, itw, wk, p, pinv, k, lds);
}

__global__ void k_consts_to_f64(const ulonglong2 *__restrict__ in, double *__restrict__ out, const Modulus *__restrict__ mods, u32 k, u32 n, u32 total) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u64 q = mods[(i / n) % k].q;
    const u64 w = in[i].x;
    out[i] = w > q / 2 ? -(double)(q - w) : (double)w;
}

}  // namespace

bool fhe_dct_f64_supported(const fhe_ctx *c) {
    return c && c->qb.d_tw_f64 && c->max_prime_bits <= 47 && (c->logn == 10 || c->logn == 12 || c->logn == 13);
}

int fhe_dct_f64_make_consts(const fhe_ctx *c, fhe_dct_plan *plan, hipStream_t st) {
    const u32 total = DCT_NCONST * c->k * c->n;
    HIP_TRY(hipMalloc(&plan->d_consts_f64, sizeof(double) * total));
    k_consts_to_f64<<<(total + 255) / 256, 256, 0, st>>>(plan->d_consts, plan->d_consts_f64, c->qb.d_mod, c->k, c->n, total);
    KERNEL_CHECK();
    return FHE_OK;
}

int fhe_dct_f64_launch(const fhe_ctx *c, const fhe_dct_plan *plan, const u64 *in, u64 *out, u64 n_blocks, double *mid, hipStream_t st) {
    const u64 items = n_blocks * 8 * 2 * c->k;   // (block, line, poly, prime), multiple of 8
    const u64 grid = items * 2;
    if (grid > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many blocks for one launch");
    const bool big = c->max_prime_bits > 40;
#define LAUNCH(LL, BB)                                                                                                        \
    do {                                                                                                                      \
        k_dct_rows<LL, BB><<<(unsigned)grid, NttShape<LL>::TP, 0, st>>>(in, mid, plan->d_consts_f64, c->qb.d_tw_f64, c->qb.d_mod, c->k);  \
        k_dct_cols<LL, BB><<<(unsigned)grid, NttShape<LL>::TP, 0, st>>>(mid, out, plan->d_consts_f64, c->qb.d_itw_f64, c->qb.d_mod, c->k); \
    } while (0)
    switch (c->logn) {
        case 10: if (big) LAUNCH(10, true); else LAUNCH(10, false); break;
        case 12: if (big) LAUNCH(12, true); else LAUNCH(12, false); break;
        case 13: if (big) LAUNCH(13, true); else LAUNCH(13, false); break;
        default: return fail(FHE_ERR_PARAM, "fused FP64 path supports n in [1024, 8192]");
    }
#undef LAUNCH
    KERNEL_CHECK();
    return FHE_OK;
}
