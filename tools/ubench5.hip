// ubench5.hip -- butterflies on a pseudo-Mersenne product against the library's Shoup butterflies (csrc/modarith.h).
// Every SEAL 2.3 default prime (and the auxiliary primes the library picks) is q = 2^b - delta with delta < 2^26.  With a
// twiddle w stored beside w2 = w 2^31 mod q, a product x w mod q for ANY x < 2^62 is
//     x = xl + 2^31 xh;   S = xl w + xh w2  (below 2^(b+32), four v_mad_u64_u32 in two chains, no carries between them);
//     S = zl + 2^b zh  (zh < 2^32);   result = zl + zh delta  (one v_mad_u64_u32)          in [0, 2^b + 2^32 delta)
// -- five multiply-adds and no v_mul_hi_u32, where the Shoup product with the approximate high word takes seven and two.
// Checks the product against the exact value, then times full forward butterflies (X + T, X - T + c q).
// Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#include "../fully-homomorphic-image-processing_amd/csrc/modarith.h"

struct PmMod { u32 delta, sh, mb; };       // q = 2^b - delta, sh = b - 32, mb = 2^sh - 1
__device__ __forceinline__ u64 pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ u64 mul_pm(u64 x, u64 w, u64 w2, PmMod m) {
    const u32 xl = (u32)x & 0x7fffffffu, xh = __builtin_amdgcn_alignbit((u32)(x >> 32), (u32)x, 31);
    const u32 wl = (u32)w, wh = (u32)(w >> 32), vl = (u32)w2, vh = (u32)(w2 >> 32);
    const u64 A = (u64)xh * vl + (u64)xl * wl;
    u64 B = (u64)xl * wh + (A >> 32);
    asm("" : "+v"(B));                      // keeps the compiler from re-associating the addend out of the multiply-add
    B = (u64)xh * vh + B;
    const u32 zh = __builtin_amdgcn_alignbit((u32)(B >> 32), (u32)B, m.sh);
    u64 zl = pack((u32)A, (u32)B & m.mb);
    asm("" : "+v"(zl));
    return (u64)zh * m.delta + zl;
}

template <int OP> __global__ __launch_bounds__(256) void k(u64 *out, const ulonglong2 *tw, const ulonglong2 *tw2, u64 q, PmMod pm, u64 seed) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[8];
    for (int i = 0; i < 8; i++) a[i] = (seed * (tid + i + 1)) >> 9;
    const ulonglong2 t = OP == 0 ? tw[tid & 255] : tw2[tid & 255];
    const u64 nq = 0 - q, q4 = 4 * q, q2 = 2 * q;
    const u32 zero = fhe_opaque_zero;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            if (OP == 0) {          // the library's LAZY forward butterfly (ntt_core.h ntt_fwd_pass4), inputs kept small by a shift
                const u64 X = a[i] >> 2;
                const u64 S = mul_shoup_lazy4_acc(a[i + 1] >> 2, t.x, t.y, nq, zero, X);
                a[i] = S; a[i + 1] = (X << 1) + q4 - S;
            }
            if (OP == 1) {
                const u64 X = a[i] >> 2, T = mul_pm(a[i + 1] >> 2, t.x, t.y, pm);
                a[i] = X + T; a[i + 1] = X - T + q2;
            }
        }
    }
    u64 acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i];
    out[tid] = acc;
}

__global__ void k_check(const ulonglong2 *tw2, u64 q, PmMod pm, u64 seed, u32 *bad, u64 *maxr) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    const ulonglong2 t = tw2[tid & 255];
    u64 x = seed * (2 * tid + 1), mx = 0;
    for (int it = 0; it < 64; it++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        u64 xs = x >> 2;
        if (it == 0) xs = (1ULL << 62) - 1; if (it == 1) xs = 0; if (it == 2) xs = 0x7fffffffULL; if (it == 3) xs = 0x3fffffff80000000ULL; if (it == 4) xs = q - 1; if (it == 5) xs = 4 * q - 1;
        const u64 r = mul_pm(xs, t.x, t.y, pm);
        const u64 want = (u64)(((unsigned __int128)xs * t.x) % q);
        if (r % q != want) atomicAdd(bad + 0, 1);
        if (r >= 2 * q) atomicAdd(bad + 1, 1);
        mx = r > mx ? r : mx;
    }
    atomicMax(maxr, mx);
}

template <int OP> int run(const char *name, u64 *out, const ulonglong2 *tw, const ulonglong2 *tw2, u64 q, PmMod pm) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, tw, tw2, q, pm, 0x9E3779B97F4A7C15ULL);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) k<OP><<<blocks, threads>>>(out, tw, tw2, q, pm, 0x9E3779B97F4A7C15ULL + r);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double n = 5.0 * blocks * threads * (double)ITERS * 4, rate = n / (ms * 1e-3);
    printf("%-52s %8.3f ms  %9.2f G butterflies/s  => %6.2f lane-butterflies/clk/CU @2.4GHz\n", name, ms / 5, rate / 1e9, rate / 256 / 2.4e9);
    return 0;
}

int main() {
    const u64 qs[4] = {0x7fffffffba0001ULL, 0x3fffffffd60001ULL, 0x7ffffffef00001ULL, 0x3ffffffffc60001ULL};   // 55, 54, 55 (delta 2^24.1), 58 bits
    u64 *out; CHK(hipMalloc(&out, sizeof(u64) * 256 * 8 * 256));
    ulonglong2 *tw, *tw2; CHK(hipMalloc(&tw, sizeof(ulonglong2) * 256)); CHK(hipMalloc(&tw2, sizeof(ulonglong2) * 256));
    u32 *bad; CHK(hipMalloc(&bad, 16));
    u64 *maxr; CHK(hipMalloc(&maxr, 8));
    for (int qi = 0; qi < 4; qi++) {
        const u64 q = qs[qi];
        int b = 64 - __builtin_clzll(q);
        PmMod pm; pm.delta = (u32)((1ULL << b) - q); pm.sh = b - 32; pm.mb = (1u << (b - 32)) - 1;
        ulonglong2 h[256], h2[256];
        u64 s = 88172645463325252ULL;
        for (int i = 0; i < 256; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            u64 w = s % q;
            if (i == 0) w = q - 1; if (i == 1) w = 1; if (i == 2) w = 0;
            h[i].x = w; h[i].y = (u64)(((unsigned __int128)w << 64) / q);
            h2[i].x = w; h2[i].y = (u64)((((unsigned __int128)w) << 31) % q);
        }
        CHK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
        CHK(hipMemcpy(tw2, h2, sizeof(h2), hipMemcpyHostToDevice));
        CHK(hipMemset(bad, 0, 16)); CHK(hipMemset(maxr, 0, 8));
        k_check<<<4096, 256>>>(tw2, q, pm, 0x2545F4914F6CDD1DULL, bad, maxr);
        u32 hb[4]; u64 hm; CHK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hm, maxr, 8, hipMemcpyDeviceToHost));
        printf("q = %#llx (b = %d, delta = %#x): wrong residues %u, results >= 2q %u, largest result / q = %.4f (of %u products)\n",
               q, b, pm.delta, hb[0], hb[1], (double)hm / (double)q, 4096u * 256u * 64u);
        if (qi == 0) {
            run<0>("forward butterfly, Shoup (approximate high word)", out, tw, tw2, q, pm);
            run<1>("forward butterfly, pseudo-Mersenne split product", out, tw, tw2, q, pm);
        }
    }
    return 0;
}
