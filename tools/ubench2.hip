// ubench2.hip -- FP64 issue/latency behaviour at LOW occupancy (1 and 2 waves per SIMD), the regime
// of the fused DCT kernels: cycles per v_fma_f64 as a function of independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int CH> __global__ __launch_bounds__(256) void k(double *out, double s, int iters) {
    double d[CH];
    for (int i = 0; i < CH; i++) d[i] = s * (threadIdx.x + i + 1);
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);
    }
    long long t1 = clock64();
    double acc = 0;
    for (int i = 0; i < CH; i++) acc += d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (double)(t1 - t0) * 1e-300;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 20] = t1 - t0;
}
template <int CH> int run(double *out, int wg_per_cu) {
    const int iters = 20000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<CH><<<256 * wg_per_cu, 256>>>(out, 1.0, 100);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<CH><<<256 * wg_per_cu, 256>>>(out, 1.0, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc; CHK(hipMemcpy(&cyc, ((long long *)out) + (1 << 20), 8, hipMemcpyDeviceToHost));
    printf("chains=%d waves/SIMD=%d: %.2f s_memtime-ticks per fma per wave, %.2f ns per fma per wave, chip %.1f Tfma/s\n", CH, wg_per_cu,
           (double)cyc / ((double)iters * CH), ms * 1e6 / ((double)iters * CH), 256.0 * wg_per_cu * 256 * iters * CH / (ms * 1e-3) / 1e12);
    return 0;
}
int main() {
    double *out; CHK(hipMalloc(&out, sizeof(double) * ((1 << 20) + 16)));
    for (int w = 1; w <= 2; w++) { run<1>(out, w); run<2>(out, w); run<4>(out, w); run<8>(out, w); run<16>(out, w); }
    return 0;
}
