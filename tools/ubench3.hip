// ubench3: the MEMORY ACCESS PATTERN of the fused DCT kernels without their arithmetic.
// Each workgroup (512 threads) reads the eight 32 KB polynomials of a line pair set (d_m, d_(7-m)),
// combines them into four and writes four 32 KB polynomials -- exactly k_dct_rows' global traffic
// (mode rows) or k_dct_cols' (mode cols: stride between the eight reads is 8 ciphertexts).
// Knobs: work order, load width, partner distance, nontemporal hints.  Reports TB/s of unique bytes.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench3.hip -o tools/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;
typedef u64 v2u64 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int N = 4096, K = 3, TP = 512, E = 8;

struct Work { unsigned blk, line, poly, prime, half; };
// order 0: prime-major (blk, line, poly fastest)  [shipping]; 1: block-major (prime inside block)
__device__ inline Work decode(unsigned idx, int order, int pdist_log) {
    // halves sit 2^pdist_log apart
    const unsigned lowmask = (1u << pdist_log) - 1;
    const unsigned w = ((idx >> (pdist_log + 1)) << pdist_log) | (idx & lowmask);
    Work o;
    o.half = (idx >> pdist_log) & 1;
    const unsigned total = gridDim.x >> 1;
    if (order == 0) {
        const unsigned per_prime = total / K;
        o.prime = w / per_prime;
        unsigned t = w - o.prime * per_prime;
        o.poly = t & 1; t >>= 1; o.line = t & 7; o.blk = t >> 3;
    } else {
        unsigned t = w;
        o.poly = t & 1; t >>= 1; o.prime = t % K; t /= K; o.line = t & 7; o.blk = t >> 3;
    }
    return o;
}

// rows pattern (8 B/lane) + `nconst` table arrays of 32 KB per workgroup (L2-resident, shared by all
// workgroups of a prime), summed into the data so the loads cannot be dropped
template <int MODE>   // 0: as the kernels do; 1: streaming loads/stores nontemporal; 2: tables only (no streaming reads)
__global__ __launch_bounds__(512, 2) void k_pattern_consts(const u64 *__restrict__ in, u64 *__restrict__ out, const u64 *__restrict__ table, int nconst, int unique_only, int replicas = 1) {
    const Work wk = decode(blockIdx.x, 0, 3);
    table += (size_t)((blockIdx.x >> 3) % replicas) * K * 76 * N;
    const int tid = threadIdx.x;
    const size_t poly_words = (size_t)K * N, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + 8 * wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * N;
    u64 x[4][E];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        // unique_only: each half reads only "its" four polynomials (no partner re-read)
        const u64 *a = in + base + (size_t)(unique_only ? 2 * m + wk.half : m) * ct_words;
        const u64 *b = in + base + (size_t)(7 - m) * ct_words;
#pragma unroll
        for (int r = 0; r < E; r++) {
            u64 A = MODE == 2 ? (u64)tid : MODE == 1 ? __builtin_nontemporal_load(a + r * TP + tid) : a[r * TP + tid];
            u64 B = (unique_only || MODE == 2) ? 0 : MODE == 1 ? __builtin_nontemporal_load(b + r * TP + tid) : b[r * TP + tid];
            x[m][r] = A + B;
        }
    }
    const u64 *tp = table + (size_t)wk.prime * 76 * N + (size_t)(wk.half * 16) * N + tid;
    for (int i = 0; i < nconst; i++) {       // one batch of 8 loads at a time, like fetch(r) in the kernels
#pragma unroll
        for (int r = 0; r < E; r++) x[0][r] += tp[(size_t)i * N + r * TP];
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        u64 *o = out + base + (size_t)(2 * m + wk.half) * ct_words;
#pragma unroll
        for (int r = 0; r < E; r++) { if (MODE == 1) __builtin_nontemporal_store(x[m][r], o + r * TP + tid); else if (MODE != 2 || x[m][r] == 12345) o[r * TP + tid] = x[m][r]; }
    }
}

// L2-hit bandwidth per CU: every workgroup reads the same 8 arrays of 32 KB, all loads independent
template <int WIDTH>
__global__ __launch_bounds__(512, 2) void k_tables_unrolled(const u64 *__restrict__ table, u64 *__restrict__ out, int reps) {
    const int tid = threadIdx.x;
    u64 acc = 0;
    for (int it = 0; it < reps; it++) {
        const u64 *tp = table + (size_t)(it & 3) * 8 * N;
        if (WIDTH == 8) {
            u64 v[64];
#pragma unroll
            for (int i = 0; i < 64; i++) v[i] = tp[(size_t)i * TP + tid];
#pragma unroll
            for (int i = 0; i < 64; i++) acc += v[i];
        } else {
            v2u64 v[32];
#pragma unroll
            for (int i = 0; i < 32; i++) v[i] = ((const v2u64 *)tp)[(size_t)i * TP + tid];
#pragma unroll
            for (int i = 0; i < 32; i++) acc += v[i].x + v[i].y;
        }
    }
    if (acc == 12345) out[tid] = acc;
}

template <int WIDTH, bool NT>
__global__ __launch_bounds__(512, 2) void k_pattern(const u64 *__restrict__ in, u64 *__restrict__ out, int cols, int order, int pdist_log, int single) {
    const Work wk = decode(blockIdx.x, order, pdist_log);
    if (single && wk.half) return;
    const int tid = threadIdx.x;
    const size_t poly_words = (size_t)K * N, ct_words = 2 * poly_words;
    const size_t line_stride = cols ? 8 * ct_words : ct_words;       // distance between the 8 polynomials read
    const size_t base = ((size_t)wk.blk * 64 + (cols ? wk.line : 8 * wk.line)) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * N;
    u64 x[4][E];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const u64 *a = in + base + (size_t)m * line_stride, *b = in + base + (size_t)(7 - m) * line_stride;
        if (WIDTH == 8) {
#pragma unroll
            for (int r = 0; r < E; r++) {
                u64 A = NT ? __builtin_nontemporal_load(a + r * TP + tid) : a[r * TP + tid];
                u64 B = NT ? __builtin_nontemporal_load(b + r * TP + tid) : b[r * TP + tid];
                x[m][r] = wk.half ? A - B : A + B;
            }
        } else {
#pragma unroll
            for (int r = 0; r < E; r += 2) {       // thread owns coefficients (r/2)*1024 + 2*tid, +1
                const v2u64 *a2 = (const v2u64 *)(a + (r / 2) * 2 * TP) + tid, *b2 = (const v2u64 *)(b + (r / 2) * 2 * TP) + tid;
                v2u64 A = NT ? __builtin_nontemporal_load(a2) : *a2;
                v2u64 B = NT ? __builtin_nontemporal_load(b2) : *b2;
                x[m][r] = wk.half ? A.x - B.x : A.x + B.x;
                x[m][r + 1] = wk.half ? A.y - B.y : A.y + B.y;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        u64 *o = out + base + (size_t)(2 * m + wk.half) * line_stride;
        if (WIDTH == 8) {
#pragma unroll
            for (int r = 0; r < E; r++) { if (NT) __builtin_nontemporal_store(x[m][r], o + r * TP + tid); else o[r * TP + tid] = x[m][r]; }
        } else {
#pragma unroll
            for (int r = 0; r < E; r += 2) {
                v2u64 v; v.x = x[m][r]; v.y = x[m][r + 1];
                v2u64 *o2 = (v2u64 *)(o + (r / 2) * 2 * TP) + tid;
                if (NT) __builtin_nontemporal_store(v, o2); else *o2 = v;
            }
        }
    }
}

__global__ void k_copy(const ulonglong2 *__restrict__ in, ulonglong2 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    const size_t words = (size_t)blocks * 64 * 2 * K * N;
    u64 *in, *out;
    CK(hipMalloc(&in, words * 8));
    CK(hipMalloc(&out, words * 8));
    CK(hipMemset(in, 1, words * 8));
    CK(hipMemset(out, 0, words * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)blocks * 8 * 2 * K * 2;
    auto run = [&](const char *name, auto launch, double bytes) {
        for (int i = 0; i < 2; i++) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 5;
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %8.3f ms  %6.2f TB/s (unique bytes)  = %7.0f blocks/s-equivalent per kernel\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e12, blocks / (ms / reps * 1e-3));
    };
    const double uniq = (double)words * 8 * 2;    // read everything once + write everything once
    run("streaming copy (16 B/lane)", [&] { k_copy<<<256 * 8, 512>>>((const ulonglong2 *)in, (ulonglong2 *)out, words / 2); }, uniq);
    for (int cols = 0; cols < 2; cols++)
        for (int order = 0; order < 2; order++)
            for (int pd = 3; pd <= 3; pd++) {
                char nm[128];
                snprintf(nm, sizeof nm, "%s order=%d  8B/lane", cols ? "cols" : "rows", order);
                run(nm, [&] { k_pattern<8, false><<<grid, TP>>>(in, out, cols, order, pd, 0); }, uniq);
                snprintf(nm, sizeof nm, "%s order=%d 16B/lane", cols ? "cols" : "rows", order);
                run(nm, [&] { k_pattern<16, false><<<grid, TP>>>(in, out, cols, order, pd, 0); }, uniq);
                snprintf(nm, sizeof nm, "%s order=%d 16B/lane nontemporal", cols ? "cols" : "rows", order);
                run(nm, [&] { k_pattern<16, true><<<grid, TP>>>(in, out, cols, order, pd, 0); }, uniq);
            }
    for (int pd = 0; pd <= 6; pd += 3) {
        char nm[128];
        snprintf(nm, sizeof nm, "rows order=0 16B/lane partner distance 2^%d", pd);
        run(nm, [&] { k_pattern<16, false><<<grid, TP>>>(in, out, 0, 0, pd, 0); }, uniq);
    }
    run("rows order=0 16B/lane, even halves only (half the writes)", [&] { k_pattern<16, false><<<grid, TP>>>(in, out, 0, 0, 3, 1); }, uniq * 0.75);
    u64 *table;
    CK(hipMalloc(&table, (size_t)64 * K * 76 * N * 8));
    CK(hipMemset(table, 1, (size_t)64 * K * 76 * N * 8));
    for (int uo = 1; uo >= 0; uo--)
        for (int nc = 0; nc <= 16; nc += 4) {
            char nm[128];
            snprintf(nm, sizeof nm, "rows 8B/lane %s + %2d const arrays (32 KB each)", uo ? "unique reads only" : "with partner re-read", nc);
            run(nm, [&] { k_pattern_consts<0><<<grid, TP>>>(in, out, table, nc, uo); }, uniq);
        }
    for (int nc = 0; nc <= 16; nc += 8) {
        char nm[128];
        snprintf(nm, sizeof nm, "rows 8B/lane partner, NONTEMPORAL streams + %2d const arrays", nc);
        run(nm, [&] { k_pattern_consts<1><<<grid, TP>>>(in, out, table, nc, 0); }, uniq);
        snprintf(nm, sizeof nm, "tables only (no streaming at all)    %2d const arrays", nc);
        run(nm, [&] { k_pattern_consts<2><<<grid, TP>>>(in, out, table, nc, 0); }, uniq);
    }
    {
        const double tb = (double)512 * 4 * 4 * 8 * N * 8;     // 2048 workgroups x 4 reps x 256 KB
        run("L2-hit tables, 64 independent 8 B/lane loads, 2048 WGs x4", [&] { k_tables_unrolled<8><<<2048, TP>>>(table, out, 4); }, tb);
        run("L2-hit tables, 32 independent 16 B/lane loads, 2048 WGs x4", [&] { k_tables_unrolled<16><<<2048, TP>>>(table, out, 4); }, tb);
    }
    for (int rep = 1; rep <= 1; rep *= 4) {
        char nm[128];
        snprintf(nm, sizeof nm, "tables only, 8 const arrays, %2d table replicas", rep);
        run(nm, [&] { k_pattern_consts<2><<<grid, TP>>>(in, out, table, 8, 0, rep); }, uniq);
        snprintf(nm, sizeof nm, "rows partner + 8 const arrays, %2d table replicas", rep);
        run(nm, [&] { k_pattern_consts<0><<<grid, TP>>>(in, out, table, 8, 0, rep); }, uniq);
    }
    return 0;
}
