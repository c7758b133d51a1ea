// dct_wave.hip -- fused DCT+quant kernels with a WAVE-SYNCHRONOUS 4096-point NTT (n = 4096 only).
//
// One wave owns one polynomial: 64 coefficients per lane in registers.  Stages 0..5 pair registers
// (twiddles are wave-uniform, fetched through the scalar cache), one 64x64 transpose through a
// wave-private LDS buffer (no workgroup barrier), stages 6..11 pair registers again.  A workgroup
// is four such waves = the four sums (or differences) of one LL&M line; the only workgroup barriers
// are the two exchanges around the per-slot circuit.  Compared with dct_fused.hip this trades
// twiddle sharing for: 1 LDS round trip per transform instead of 3, 4 barriers per kernel instead
// of 12+, and 32 independent butterflies per stage per lane, which is what the FP64 pipe (32-cycle
// dependent issue) needs at two waves per SIMD.
//
// Layouts: NTT form index j = 64*lane + r' after the forward transform; the intermediate and the
// constant table use position (j & 63) * 64 + (j >> 6), so that after the exchange thread
// (wave w, lane l) works on slots r' in [16w, 16w+16) of all four polynomials with lane-contiguous
// global accesses.
#include "fp64_core.h"

#include <cstdlib>

#pragma clang fp contract(off)

namespace {
using namespace fp64;

constexpr int WL = 12, WN = 4096;
constexpr int KB = 4;                  // butterflies per product batch
constexpr int TBUF = 32 * 65;          // wave-private transpose buffer (doubles): 32 rows, padded to 65
constexpr int LDS_DOUBLES = 4 * TBUF;  // 66,560 B; the exchange region (4*32*64 doubles) aliases its head

__host__ __device__ constexpr int ins0(int b, int bit) { return ((b >> bit) << (bit + 1)) | (b & ((1 << bit) - 1)); }

// one forward (Cooley-Tukey) stage on register bit BIT; tw(r0) yields the twiddle of pair (r0, r0|1<<BIT)
template <int BIT, typename TWF>
__device__ __forceinline__ void fwd_stage(double (&x)[64], TWF tw, double p, double pinv) {
#pragma unroll
    for (int b0 = 0; b0 < 32; b0 += KB) {
        double t[KB], w[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) { const int r0 = ins0(b0 + k, BIT); t[k] = x[r0 | (1 << BIT)]; w[k] = tw(r0); }
        mmv<KB>(t, w, p, pinv);
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int r0 = ins0(b0 + k, BIT);
            const double X = x[r0];
            x[r0] = X + t[k];
            x[r0 | (1 << BIT)] = X - t[k];
        }
    }
}
// one inverse (Gentleman-Sande) stage
template <int BIT, typename TWF>
__device__ __forceinline__ void inv_stage(double (&x)[64], TWF tw, double p, double pinv) {
#pragma unroll
    for (int b0 = 0; b0 < 32; b0 += KB) {
        double t[KB], w[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int r0 = ins0(b0 + k, BIT), r1 = r0 | (1 << BIT);
            const double X = x[r0], Y = x[r1];
            x[r0] = X + Y;
            t[k] = X - Y;
            w[k] = tw(r0);
        }
        mmv<KB>(t, w, p, pinv);
#pragma unroll
        for (int k = 0; k < KB; k++) x[ins0(b0 + k, BIT) | (1 << BIT)] = t[k];
    }
}
// last inverse stage (sigma = 0, register bit 5): sums take n^-1, differences take w*n^-1
__device__ __forceinline__ void inv_last_stage(double (&x)[64], double ninv, double w1, double p, double pinv) {
#pragma unroll
    for (int b0 = 0; b0 < 32; b0 += 4) {
        double t[8], w[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r0 = b0 + k, r1 = r0 | 32;
            const double X = x[r0], Y = x[r1];
            t[k] = X + Y; w[k] = ninv;
            t[4 + k] = X - Y; w[4 + k] = w1;
        }
        mmv<8>(t, w, p, pinv);
#pragma unroll
        for (int k = 0; k < 4; k++) { x[b0 + k] = t[k]; x[(b0 + k) | 32] = t[4 + k]; }
    }
}

// stages 0..5: register r holds index bits 11..6, twiddle 2^s + (r >> (6-s)) is wave-uniform
template <int S> __device__ __forceinline__ void fwd_hi(double (&x)[64], const double *__restrict__ tw, double p, double pinv) {
    constexpr int BIT = 5 - S;
    fwd_stage<BIT>(x, [&](int r0) { return tw[(1 << S) + (r0 >> (BIT + 1))]; }, p, pinv);
    if constexpr (S < 5) fwd_hi<S + 1>(x, tw, p, pinv);
}
// stages 6..11: register r' holds index bits 5..0, index bits 11..6 are the lane
template <int S> __device__ __forceinline__ void fwd_lo(double (&x)[64], const double *__restrict__ tw, int lane, double p, double pinv) {
    constexpr int BIT = 11 - S;
    fwd_stage<BIT>(x, [&](int r0) { return tw[(1 << S) + (lane << (S - 6)) + (r0 >> (BIT + 1))]; }, p, pinv);
    asm volatile("" ::: "memory");          // keep the next stage's per-lane twiddle loads from being hoisted up here
    if constexpr (S < 11) fwd_lo<S + 1>(x, tw, lane, p, pinv);
}
template <int S> __device__ __forceinline__ void inv_lo(double (&x)[64], const double *__restrict__ itw, int lane, double p, double pinv) {
    constexpr int BIT = 11 - S;
    inv_stage<BIT>(x, [&](int r0) { return itw[(1 << S) + (lane << (S - 6)) + (r0 >> (BIT + 1))]; }, p, pinv);
    asm volatile("" ::: "memory");
    if constexpr (S > 6) inv_lo<S - 1>(x, itw, lane, p, pinv);
}
template <int S> __device__ __forceinline__ void inv_hi(double (&x)[64], const double *__restrict__ itw, double p, double pinv) {
    constexpr int BIT = 5 - S;
    if constexpr (S == 0) {
        inv_last_stage(x, itw[0], itw[1], p, pinv);
    } else {
        inv_stage<BIT>(x, [&](int r0) { return itw[(1 << S) + (r0 >> (BIT + 1))]; }, p, pinv);
        inv_hi<S - 1>(x, itw, p, pinv);
    }
}

// (lane a, register b) <-> (lane b, register a) through a wave-private buffer, in two halves so that
// eight waves fit the CU's LDS.  DS operations of one wave execute in order; the waits make the
// cross-lane visibility explicit for the compiler and the hardware alike.
__device__ __forceinline__ void wave_transpose64(double (&x)[64], double *buf, int lane) {
    double tmp[32];
#pragma unroll
    for (int r = 0; r < 32; r++) buf[r * 65 + lane] = x[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < 32) {
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = buf[lane * 65 + r];
#pragma unroll
        for (int r = 0; r < 32; r++) tmp[r] = buf[lane * 65 + 32 + r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 32; r++) buf[r * 65 + lane] = x[32 + r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane >= 32) {
#pragma unroll
        for (int r = 0; r < 64; r++) x[r] = buf[(lane - 32) * 65 + r];
    } else {
#pragma unroll
        for (int r = 0; r < 32; r++) x[32 + r] = tmp[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int HALF>
__device__ __forceinline__ void circuit_rows(double (&y)[4][16], const double *__restrict__ cp, size_t cstride, double p, double pinv) {
    constexpr int NC = HalfC<HALF>::NC, FIRST = HalfC<HALF>::FIRST;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        double c[9];
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = cp[(size_t)(FIRST + i) * cstride + s * 64];
        line_half<HALF>(y[0][s], y[1][s], y[2][s], y[3][s], c, p, pinv);
    }
}
template <int HALF>
__device__ __forceinline__ void circuit_cols(double (&y)[4][16], const double *__restrict__ cp, const double *__restrict__ sp, size_t cstride, double p, double pinv) {
    constexpr int NC = HalfC<HALF>::NC, FIRST = HalfC<HALF>::FIRST;
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
        double c0[9], c1[9], sc[8];
#pragma unroll
        for (int i = 0; i < NC; i++) { c0[i] = cp[(size_t)(FIRST + i) * cstride + s * 64]; c1[i] = cp[(size_t)(FIRST + i) * cstride + (s + 1) * 64]; }
#pragma unroll
        for (int m = 0; m < 4; m++) { sc[m] = sp[(size_t)(16 * m) * cstride + s * 64]; sc[4 + m] = sp[(size_t)(16 * m) * cstride + (s + 1) * 64]; }
        line_half<HALF>(y[0][s], y[1][s], y[2][s], y[3][s], c0, p, pinv);
        line_half<HALF>(y[0][s + 1], y[1][s + 1], y[2][s + 1], y[3][s + 1], c1, p, pinv);
        double v[8] = {y[0][s], y[1][s], y[2][s], y[3][s], y[0][s + 1], y[1][s + 1], y[2][s + 1], y[3][s + 1]};
        mmv<8>(v, sc, p, pinv);
#pragma unroll
        for (int m = 0; m < 4; m++) { y[m][s] = v[m]; y[m][s + 1] = v[4 + m]; }
    }
}

__global__ __launch_bounds__(256, 2) void k_rows_wave(const u64 *__restrict__ in, double *__restrict__ mid, const double *__restrict__ consts,
                                                      const double *__restrict__ tw_all, const Modulus *__restrict__ mods, u32 k) {
    __shared__ double lds[LDS_DOUBLES];
    const Work wk = decode(blockIdx.x, k);
    const double p = (double)mods[wk.prime].q, pinv = 1.0 / p;
    const double *tw = tw_all + (size_t)wk.prime * WN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = wave;
    const size_t poly_words = (size_t)k * WN, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + 8 * wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * WN;
    const u64 *a = in + base + (size_t)m * ct_words + lane, *b = in + base + (size_t)(7 - m) * ct_words + lane;
    constexpr u64 OFF = 1ULL << 48;
    double x[64];
#pragma unroll
    for (int g = 0; g < 4; g++) {           // coefficient j = 64 r + lane; 32 loads in flight per group
        u64 ra[16], rb[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { ra[i] = a[(16 * g + i) * 64]; rb[i] = b[(16 * g + i) * 64]; }
#pragma unroll
        for (int i = 0; i < 16; i++) x[16 * g + i] = wk.half ? u52_to_f64(ra[i] + OFF - rb[i]) - (double)OFF : u52_to_f64(ra[i] + rb[i]);
        asm volatile("" ::: "memory");
    }
    fwd_hi<0>(x, tw, p, pinv);
    wave_transpose64(x, lds + wave * TBUF, lane);
    fwd_lo<6>(x, tw, lane, p, pinv);
    // exchange: afterwards thread (wave, lane) holds slots r' in [16 wave, 16 wave + 16) of all four polynomials
    double y[4][16];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        __syncthreads();
#pragma unroll
        for (int rl = 0; rl < 32; rl++) lds[(m * 32 + rl) * 64 + lane] = x[32 * c + rl];
        __syncthreads();
        if ((wave >> 1) == c) {
#pragma unroll
            for (int mm_ = 0; mm_ < 4; mm_++)
#pragma unroll
                for (int s = 0; s < 16; s++) y[mm_][s] = lds[(mm_ * 32 + (16 * (wave & 1) + s)) * 64 + lane];
        }
    }
    const size_t cstride = (size_t)k * WN;
    const size_t pos = (size_t)(16 * wave) * 64 + lane;
    const double *cp = consts + (size_t)wk.prime * WN + pos;
    if (wk.half) circuit_rows<1>(y, cp, cstride, p, pinv);
    else circuit_rows<0>(y, cp, cstride, p, pinv);
#pragma unroll
    for (int mm_ = 0; mm_ < 4; mm_++) {
        double *o = mid + base + (size_t)(2 * mm_ + wk.half) * ct_words + pos;
#pragma unroll
        for (int s = 0; s < 16; s++) o[s * 64] = y[mm_][s];
    }
}

__global__ __launch_bounds__(256, 2) void k_cols_wave(const double *__restrict__ mid, u64 *__restrict__ out, const double *__restrict__ consts,
                                                      const double *__restrict__ itw_all, const Modulus *__restrict__ mods, u32 k) {
    __shared__ double lds[LDS_DOUBLES];
    const Work wk = decode(blockIdx.x, k);   // line = column
    const double p = (double)mods[wk.prime].q, pinv = 1.0 / p;
    const double *itw = itw_all + (size_t)wk.prime * WN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = wave;
    const size_t poly_words = (size_t)k * WN, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * WN;
    const size_t row_stride = 8 * ct_words, cstride = (size_t)k * WN;
    const size_t pos = (size_t)(16 * wave) * 64 + lane;
    double y[4][16];
#pragma unroll
    for (int mm_ = 0; mm_ < 4; mm_++) {
        const double *a = mid + base + (size_t)mm_ * row_stride + pos, *b = mid + base + (size_t)(7 - mm_) * row_stride + pos;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const double A = a[s * 64], B = b[s * 64];
            y[mm_][s] = wk.half ? A - B : A + B;
        }
    }
    const double *cp = consts + (size_t)wk.prime * WN + pos;
    const double *sp = consts + (size_t)(12 + 8 * wk.half + wk.line) * cstride + (size_t)wk.prime * WN + pos;
    if (wk.half) circuit_cols<1>(y, cp, sp, cstride, p, pinv);
    else circuit_cols<0>(y, cp, sp, cstride, p, pinv);
    // exchange back: wave m collects polynomial m at j = 64 lane + r'
    double x[64];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        __syncthreads();
        if ((wave >> 1) == c) {
#pragma unroll
            for (int mm_ = 0; mm_ < 4; mm_++)
#pragma unroll
                for (int s = 0; s < 16; s++) lds[(mm_ * 32 + (16 * (wave & 1) + s)) * 64 + lane] = y[mm_][s];
        }
        __syncthreads();
#pragma unroll
        for (int rl = 0; rl < 32; rl++) x[32 * c + rl] = lds[(m * 32 + rl) * 64 + lane];
    }
    __syncthreads();                        // the exchange region aliases the other waves' transpose buffers
    inv_lo<11>(x, itw, lane, p, pinv);
    wave_transpose64(x, lds + wave * TBUF, lane);
    inv_hi<5>(x, itw, p, pinv);
    u64 *o = out + base + (size_t)(2 * m + wk.half) * row_stride + lane;
#pragma unroll
    for (int r = 0; r < 64; r++) {
        double v = x[r];
        v = v < 0.0 ? v + p : v;
        o[r * 64] = f64_to_u52(v);
    }
}

// Shoup-pair table in the u64 kernels' slot order -> centred doubles at position (j & 63)*64 + (j >> 6)
__global__ void k_consts_to_wave(const ulonglong2 *__restrict__ in, double *__restrict__ out, const Modulus *__restrict__ mods, u32 k, u32 n, u32 total) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u32 pos = i % n, rowbase = i - pos;
    const u32 j = ((pos & 63) << 6) | (pos >> 6);
    const u32 src = (j & 15) * (n >> 4) + (j >> 4);
    const u64 q = mods[(i / n) % k].q;
    const u64 w = in[rowbase + src].x;
    out[i] = w > q / 2 ? -(double)(q - w) : (double)w;
}

}  // namespace

bool fhe_dct_wave_supported(const fhe_ctx *c) {
    return c && c->qb.d_tw_f64 && c->logn == 12 && c->max_prime_bits <= 40;
}
int fhe_dct_wave_make_consts(const fhe_ctx *c, fhe_dct_plan *plan, hipStream_t st) {
    const u32 total = DCT_NCONST * c->k * c->n;
    HIP_TRY(hipMalloc(&plan->d_consts_wave, sizeof(double) * total));
    k_consts_to_wave<<<(total + 255) / 256, 256, 0, st>>>(plan->d_consts, plan->d_consts_wave, c->qb.d_mod, c->k, c->n, total);
    KERNEL_CHECK();
    return FHE_OK;
}
int fhe_dct_wave_launch(const fhe_ctx *c, const fhe_dct_plan *plan, const u64 *in, u64 *out, u64 n_blocks, double *mid, hipStream_t st) {
    const u64 grid = n_blocks * 8 * 2 * c->k * 2;
    if (grid > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many blocks for one launch");
    k_rows_wave<<<(unsigned)grid, 256, 0, st>>>(in, mid, plan->d_consts_wave, c->qb.d_tw_f64, c->qb.d_mod, c->k);
    k_cols_wave<<<(unsigned)grid, 256, 0, st>>>(mid, out, plan->d_consts_wave, c->qb.d_itw_f64, c->qb.d_mod, c->k);
    KERNEL_CHECK();
    return FHE_OK;
}
