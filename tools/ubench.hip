// ubench.hip -- instruction-rate probes on gfx950 for the modular-arithmetic design choices
// (u64 Shoup butterflies vs exact FP64-FMA modular multiplication).  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
#define ITERS 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP> __global__ __launch_bounds__(256) void k(u64 *out, u64 seed, double dseed) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[8]; double d[8]; u32 w[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (tid + i + 1); d[i] = dseed * (tid + i + 1); w[i] = (u32)a[i]; }
    const u64 q = 0xffffee001ULL, wc = 0x123456789ULL % q, wp = (u64)(((unsigned __int128)wc << 64) / q);
    const double p = (double)q, pinv = 1.0 / p, wd = 12345678901.0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) w[i] = w[i] * (w[i] | 3u);                                     // v_mul_lo_u32
            if (OP == 1) w[i] = __umulhi(w[i], w[i] | 0x80000001u);                      // v_mul_hi_u32
            if (OP == 2) a[i] = (u64)(u32)a[i] * (u32)(a[i] >> 32) + a[i];               // v_mad_u64_u32
            if (OP == 3) w[i] = __umul24(w[i], w[i] | 5u) + 1;                           // v_mul_u32_u24 (+add)
            if (OP == 4) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);                     // v_fma_f64
            if (OP == 5) d[i] = d[i] * 1.0000001;                                        // v_mul_f64
            if (OP == 6) d[i] = d[i] + 1.5;                                              // v_add_f64
            if (OP == 7) d[i] = __builtin_rint(d[i] * 1.0000001);                        // v_rndne_f64 + mul
            if (OP == 8) a[i] = a[i] + (a[i] >> 7);                                      // 64-bit add + shift
            if (OP == 9) { a[i] = a[i] * wc - __umul64hi(a[i], wp) * q; }                // Shoup lazy modmul
            if (OP == 10) {                                                              // u64 Harvey butterfly pair(i, i^1) approx
                u64 X = a[i] >= 2 * q ? a[i] - 2 * q : a[i];
                u64 T = a[i ^ 1] * wc - __umul64hi(a[i ^ 1], wp) * q;
                a[i] = X + T;
            }
            if (OP == 11) {                                                              // FP64 modmul: exact, centred
                double h = d[i] * wd, l = __builtin_fma(d[i], wd, -h);
                double qq = __builtin_rint(h * pinv);
                d[i] = __builtin_fma(-qq, p, h) + l;
            }
            if (OP == 12) {                                                              // FP64 butterfly (one output)
                double y = d[i ^ 1];
                double h = y * wd, l = __builtin_fma(y, wd, -h);
                double qq = __builtin_rint(h * pinv);
                double t = __builtin_fma(-qq, p, h) + l;
                d[i] = d[i] * 0.5 + t;
            }
            if (OP == 13) { u64 s = a[i] + a[i ^ 1]; a[i] = s >= q ? s - q : s; }        // addmod
        }
    }
    u64 acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i] + (u64)d[i] + w[i];
    out[tid] = acc;
}

template <int OP> int run(const char *name, u64 *out, double ops_per_iter_elem) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, 0x9E3779B97F4A7C15ULL, 1.000001);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) k<OP><<<blocks, threads>>>(out, 0x9E3779B97F4A7C15ULL + r, 1.000001);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    double n = 5.0 * blocks * threads * (double)ITERS * 8;
    double rate = n / (ms * 1e-3);
    // lanes per clock per CU at 2.4 GHz nominal
    printf("%-28s %8.3f ms  %9.2f Gop/s  => %6.2f lane-ops/clk/CU @2.4GHz (x%.0f instr)\n", name, ms / 5, rate / 1e9, rate / 256 / 2.4e9, ops_per_iter_elem);
    return 0;
}

int main() {
    u64 *out; CHK(hipMalloc(&out, sizeof(u64) * 256 * 8 * 256));
    run<0>("v_mul_lo_u32", out, 1); run<1>("v_mul_hi_u32", out, 1); run<2>("v_mad_u64_u32", out, 1);
    run<3>("v_mul_u32_u24+add", out, 2); run<4>("v_fma_f64", out, 1); run<5>("v_mul_f64", out, 1);
    run<6>("v_add_f64", out, 1); run<7>("v_mul_f64+v_rndne_f64", out, 2); run<8>("u64 add+shift", out, 3);
    run<9>("shoup lazy modmul (u64)", out, 1); run<10>("u64 half-butterfly", out, 1);
    run<11>("fp64 exact modmul", out, 1); run<12>("fp64 half-butterfly", out, 1); run<13>("u64 addmod", out, 1);
    return 0;
}
