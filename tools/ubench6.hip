// ubench6.hip -- where the time of the 8192-point forward transform kernel goes: the library's pseudo-Mersenne pair kernel
// (ntt_core.h) with the butterflies and / or the LDS transposes taken out.  Timing only (random tables).  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#include "../fully-homomorphic-image-processing_amd/csrc/ntt_core.h"

// transpose through HALF the buffer: two rounds, round h moves the registers whose old index has bit 3 = h
template <int L, int LO_FROM, int LO_TO>
__device__ __forceinline__ void transpose_half(u64 (&x)[16], u64 *lds, int tid) {
    constexpr int PL = imin(LO_FROM, LO_TO), SB = LO_FROM + 3;      // SB: the coefficient-index bit that says which round
    auto compact = [](int j) { return ((j >> (SB + 1)) << SB) | (j & ((1 << SB) - 1)); };
#pragma unroll
    for (int h = 0; h < 2; h++) {
        __syncthreads();
#pragma unroll
        for (int r = 8 * h; r < 8 * h + 8; r++) lds[lds_pad<PL>(compact(elem_index<LO_FROM>(tid, r)))] = x[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = elem_index<LO_TO>(tid, r);
            if (((j >> SB) & 1) == h) x[r] = lds[lds_pad<PL>(compact(j))];
        }
    }
}
template <int L, typename C, bool EARLY, bool HALF, int P = 0>
__device__ __forceinline__ void regs_half(u64 (&x)[1][16], const ulonglong2 *__restrict__ tw, const PmMod &m, u64 *lds, int tid, PmPassTw *pre = nullptr) {
    PmPassTw t0;
    PmPassTw &t = (P == 0 || !EARLY) ? t0 : *pre;
    ntt_fwd_pass_pm<L, P, 1, 16, C::LIM, C::CS, EARLY ? pm_fwd_pre(L, P) : 0>(x, t, tw, m, tid);
    if constexpr (P + 1 < NttShape<L>::NP) {
        PmPassTw nx;
        if constexpr (EARLY) {
            constexpr int PRE = pm_fwd_pre(L, P + 1);
            pm_tw_load<L, P + 1, 0>(nx, tw, tid);
            if constexpr (PRE > 1) pm_tw_load<L, P + 1, 1>(nx, tw, tid);
            if constexpr (PRE > 2) pm_tw_load<L, P + 1, 2>(nx, tw, tid);
            PM_FENCE();
        }
        if constexpr (HALF) transpose_half<L, pass_lo(L, P), pass_lo(L, P + 1)>(x[0], lds, tid);
        else ntt_transpose<pass_lo(L, P), pass_lo(L, P + 1)>(x[0], lds, tid);
        regs_half<L, C, EARLY, HALF, P + 1>(x, tw, m, lds, tid, &nx);
    }
}
template <int L, typename C, int WPS, bool EARLY, bool HALF>
__global__ __launch_bounds__(NttShape<L>::TP, WPS) void kh(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base) {
    __shared__ u64 lds[HALF ? NttShape<L>::LDS_WORDS / 2 : NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u64 rp = blockIdx.x;
    const u32 prime = (u32)(rp % base.count);
    const PmMod m = base.pm[prime];
    u64 x[1][16];
    load_coeff<L>(x[0], in + rp * N, tid);
    regs_half<L, C, EARLY, HALF>(x, base.tw_pm + (size_t)prime * N, m, lds, tid);
#pragma unroll
    for (int r = 0; r < 16; r++) x[0][r] = canon_pm(x[0][r], m);
    store_slots<L>(x[0], out + rp * N, tid);
}
template <int WPS, bool EARLY, bool HALF> int runh(const u64 *in, u64 *out, RnsBase base, u64 polys) {
    constexpr int L = 13;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    kh<L, PmA, WPS, EARLY, HALF><<<(unsigned)polys, NttShape<L>::TP>>>(in, out, base);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) kh<L, PmA, WPS, EARLY, HALF><<<(unsigned)polys, NttShape<L>::TP>>>(in, out, base);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("M=1, %s LDS buffer, twiddles %s, %d waves/SIMD      %8.3f ms   %6.0f GB/s read+write\n", HALF ? "half" : "full", EARLY ? "early" : "late ", WPS, ms, 2.0 * polys * 65536 / ms / 1e6);
    return 0;
}
template <int L, int M, typename C, int VAR, int P = 0>
__device__ __forceinline__ void regs_var(u64 (&x)[M][16], const ulonglong2 *__restrict__ tw, const PmMod &m, u64 *lds, int tid, PmPassTw *pre = nullptr) {
    PmPassTw t0;
    PmPassTw &t = (P == 0) ? t0 : *pre;
    if constexpr (VAR & 1) ntt_fwd_pass_pm<L, P, M, 16, C::LIM, C::CS, pm_fwd_pre(L, P)>(x, t, tw, m, tid);
    if constexpr (P + 1 < NttShape<L>::NP) {
        PmPassTw nx;
        if constexpr (VAR & 1) {
            constexpr int PRE = pm_fwd_pre(L, P + 1);
            pm_tw_load<L, P + 1, 0>(nx, tw, tid);
            if constexpr (PRE > 1) pm_tw_load<L, P + 1, 1>(nx, tw, tid);
            if constexpr (PRE > 2) pm_tw_load<L, P + 1, 2>(nx, tw, tid);
            PM_FENCE();
        }
        if constexpr (VAR & 2) {
#pragma unroll
            for (int j = 0; j < M; j++) ntt_transpose<pass_lo(L, P), pass_lo(L, P + 1)>(x[j], lds, tid);
        }
        regs_var<L, M, C, VAR, P + 1>(x, tw, m, lds, tid, &nx);
    }
}
template <int L, int M, typename C, int VAR>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base, u32 pair_stride) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u32 prime = blockIdx.x % base.count;
    const u64 g = blockIdx.x / base.count;
    const PmMod m = base.pm[prime];
    u64 x[M][16];
#pragma unroll
    for (int j = 0; j < M; j++) load_coeff<L>(x[j], in + ((M * g + j) * pair_stride + prime) * N, tid);
    regs_var<L, M, C, VAR>(x, base.tw_pm + (size_t)prime * N, m, lds, tid);
#pragma unroll
    for (int j = 0; j < M; j++) {
        if constexpr (VAR & 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) x[j][r] = canon_pm(x[j][r], m);
        }
        store_slots<L>(x[j], out + ((M * g + j) * pair_stride + prime) * N, tid);
    }
}
// persistent workgroups, one polynomial at a time, the NEXT polynomial's coefficients fetched before the current one is transformed
template <int L, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void kp(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base, u64 total) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    u64 x[1][16], nx[16];
    u64 rp = blockIdx.x;
    if (rp >= total) return;
    load_coeff<L>(x[0], in + rp * N, tid);
    for (; rp < total; rp += gridDim.x) {
        const u64 rn = rp + gridDim.x;
        if (rn < total) load_coeff<L>(nx, in + rn * N, tid);
        PM_FENCE();
        const u32 prime = (u32)(rp % base.count);
        const PmMod m = base.pm[prime];
        ntt_fwd_regs_pm<L, 1, 16, C::LIM, C::CS>(x, base.tw_pm + (size_t)prime * N, m, lds, tid);
#pragma unroll
        for (int r = 0; r < 16; r++) x[0][r] = canon_pm(x[0][r], m);
        store_slots<L>(x[0], out + rp * N, tid);
#pragma unroll
        for (int r = 0; r < 16; r++) x[0][r] = nx[r];
    }
}
int runp(const u64 *in, u64 *out, RnsBase base, u64 polys, unsigned grid) {
    constexpr int L = 13;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    kp<L, PmA><<<grid, NttShape<L>::TP>>>(in, out, base, polys);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) kp<L, PmA><<<grid, NttShape<L>::TP>>>(in, out, base, polys);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("persistent, next polynomial prefetched, grid %5u  %8.3f ms   %6.0f GB/s read+write\n", grid, ms, 2.0 * polys * 65536 / ms / 1e6);
    return 0;
}
// the same forward transform with 2^LE coefficients per thread (LE = 3: 1024 threads per 8192-point polynomial, twice the waves)
template <int L, int LE> struct G {
    static constexpr int N = 1 << L, E = 1 << LE, TP = N >> LE, NP = (L + LE - 1) / LE, LDS_WORDS = N + (N >> LE);
    static constexpr int lo(int p) { return (L - LE * p - LE) < 0 ? 0 : (L - LE * p - LE); }
    static constexpr int stages(int p) { return (L - LE * p) > LE ? LE : (L - LE * p); }
};
template <int L, int LE, int P, int U>
__device__ __forceinline__ void g_stage(u64 (&x)[1 << LE], const ulonglong2 *__restrict__ tw, const PmMod &m, u64 off, int tid) {
    using T = G<L, LE>;
    constexpr int E = T::E, LO = T::lo(P), sigma = LE * P + U, rb = (L - 1 - sigma) - LO, CNT = 1 << (LE - 1 - rb);
    const int th = (P == 0) ? 0 : (tid >> LO);
    ulonglong2 w[CNT];
#pragma unroll
    for (int i = 0; i < CNT; i++) w[i] = tw[(1 << sigma) + (i << (sigma - (LE - 1 - rb))) + th];
#pragma unroll
    for (int b = 0; b < E / 2; b++) {
        const int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
        const u64 X = x[r0], T2 = mul_pm(x[r1], w[r0 >> (rb + 1)], m);
        x[r0] = X + T2;
        x[r1] = X - T2 + off;
    }
}
template <int L, int LE, int P, int U = 0>
__device__ __forceinline__ void g_pass(u64 (&x)[1 << LE], const ulonglong2 *__restrict__ tw, const PmMod &m, u64 off, int tid) {
    g_stage<L, LE, P, U>(x, tw, m, off, tid);
    if constexpr (U + 1 < G<L, LE>::stages(P)) g_pass<L, LE, P, U + 1>(x, tw, m, off, tid);
}
template <int L, int LE, int P = 0>
__device__ __forceinline__ void g_fwd(u64 (&x)[1 << LE], const ulonglong2 *__restrict__ tw, const PmMod &m, u64 off, u64 *lds, int tid) {
    using T = G<L, LE>;
    g_pass<L, LE, P>(x, tw, m, off, tid);
    if constexpr (P + 1 < T::NP) {
        constexpr int LF = T::lo(P), LT = T::lo(P + 1), PL = LF < LT ? LF : LT, E = T::E;
        auto idx = [](int lo_, int tid_, int r) { return ((tid_ >> lo_) << (lo_ + LE)) | (r << lo_) | (tid_ & ((1 << lo_) - 1)); };
        auto pad = [](int j) { return j + ((j >> (PL + LE)) << PL); };
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; r++) lds[pad(idx(LF, tid, r))] = x[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; r++) x[r] = lds[pad(idx(LT, tid, r))];
        g_fwd<L, LE, P + 1>(x, tw, m, off, lds, tid);
    }
}
template <int L, int LE, int WPS>
__global__ __launch_bounds__((G<L, LE>::TP), WPS) void kg(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base) {
    using T = G<L, LE>;
    __shared__ u64 lds[T::LDS_WORDS];
    const int tid = threadIdx.x;
    const u64 rp = blockIdx.x;
    const u32 prime = (u32)(rp % base.count);
    const PmMod m = base.pm[prime];
    u64 x[T::E];
#pragma unroll
    for (int r = 0; r < T::E; r++) x[r] = in[rp * T::N + r * T::TP + tid];
    g_fwd<L, LE>(x, base.tw_pm + (size_t)prime * T::N, m, m.q << 3, lds, tid);
#pragma unroll
    for (int r = 0; r < T::E; r++) out[rp * T::N + r * T::TP + tid] = canon_pm(x[r], m);
}
template <int LE, int WPS> int rung(const u64 *in, u64 *out, RnsBase base, u64 polys) {
    constexpr int L = 13;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    kg<L, LE, WPS><<<(unsigned)polys, G<L, LE>::TP>>>(in, out, base);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) kg<L, LE, WPS><<<(unsigned)polys, G<L, LE>::TP>>>(in, out, base);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("%2d coefficients per thread (%4d threads), launch bound %d waves/SIMD   %8.3f ms   %6.0f GB/s read+write\n", 1 << LE, G<L, LE>::TP, WPS, ms, 2.0 * polys * 65536 / ms / 1e6);
    return 0;
}
template <int M, int VAR> int run(const char *name, const u64 *in, u64 *out, RnsBase base, u64 polys) {
    constexpr int L = 13;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(polys / M);
    k<L, M, PmA, VAR><<<grid, NttShape<L>::TP>>>(in, out, base, base.count);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) k<L, M, PmA, VAR><<<grid, NttShape<L>::TP>>>(in, out, base, base.count);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("%-44s M=%d  %8.3f ms   %6.0f GB/s read+write\n", name, M, ms, 2.0 * polys * 65536 / ms / 1e6);
    return 0;
}
int main() {
    const u64 q = 0x7fffffff380001ULL;
    const u32 n = 8192, cnt = 4;
    const u64 polys = 2048ULL * 2 * cnt;
    u64 *in, *out; CHK(hipMalloc(&in, polys * n * 8)); CHK(hipMalloc(&out, polys * n * 8));
    CHK(hipMemset(in, 1, polys * n * 8));
    std::vector<ulonglong2> tw(cnt * n);
    u64 s = 88172645463325252ULL;
    for (auto &t : tw) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; t.x = s % q; s ^= s << 13; s ^= s >> 7; s ^= s << 17; t.y = s % q; }
    ulonglong2 *dtw; CHK(hipMalloc(&dtw, tw.size() * 16)); CHK(hipMemcpy(dtw, tw.data(), tw.size() * 16, hipMemcpyHostToDevice));
    PmMod pm[4]; for (auto &p : pm) { p.q = q; p.delta = (u32)((1ULL << 55) - q); p.sh = 23; p.mb = (1u << 23) - 1; p.pad = 0; }
    PmMod *dpm; CHK(hipMalloc(&dpm, sizeof pm)); CHK(hipMemcpy(dpm, pm, sizeof pm, hipMemcpyHostToDevice));
    RnsBase base{nullptr, nullptr, nullptr, cnt, dtw, dtw, dpm};
    run<2, 3>("full transform", in, out, base, polys);
    run<2, 1>("butterflies, no LDS transposes", in, out, base, polys);
    run<2, 2>("LDS transposes, no butterflies", in, out, base, polys);
    run<2, 0>("load + store only", in, out, base, polys);
    rung<4, 4>(in, out, base, polys);
    rung<3, 4>(in, out, base, polys);
    rung<3, 8>(in, out, base, polys);
    rung<2, 8>(in, out, base, polys);
    runh<4, true, false>(in, out, base, polys);
    runh<4, false, false>(in, out, base, polys);
    runh<4, true, true>(in, out, base, polys);
    runh<4, false, true>(in, out, base, polys);
    runh<6, true, true>(in, out, base, polys);
    runh<6, false, true>(in, out, base, polys);
    run<1, 3>("full transform", in, out, base, polys);
    run<1, 1>("butterflies, no LDS transposes", in, out, base, polys);
    run<1, 2>("LDS transposes, no butterflies", in, out, base, polys);
    run<1, 0>("load + store only", in, out, base, polys);
    return 0;
}
