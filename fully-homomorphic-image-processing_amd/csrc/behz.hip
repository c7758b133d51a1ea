// behz.hip -- seal::Evaluator::multiply / square (full-RNS BEHZ) and relinearize on gfx950.
//
// Algorithm: Bajard, Eynard, Hasan, Zucca, "A Full RNS Variant of FV like Somewhat Homomorphic
// Encryption Schemes" (SAC 2016) with the conventions of SEAL 2.3:
//   0. FastBConv of m~*c from q to Bsk u {m~}, m~ = 2^32
//   1. small Montgomery reduction: r = -x q^-1 mod m~ (centred), c' = (x + q r)/m~ in Bsk
//   2. NTT in q and Bsk, tensor product sum_{a+b=o} c'1_a c'2_b, inverse NTT, times t
//   3. fast floor: (t D - FastBConv([t D]_q -> Bsk)) q^-1 in Bsk
//   4. Shenoy-Kumaresan conversion Bsk -> q with m_sk
// The result is a function of (inputs, q, t, m~) only; the auxiliary primes (k NTT primes + m_sk, the first primes
// = 1 mod 2^17 below 2^58, or below 2^61 when those are too small: fhe_behz_build) just have to be large enough.
// Reference call sites: homo/fhe_resize.h:174-179,197-198; homo/fhe_decode.h:67-97,235,239.
#include "internal.h"

#include <cstring>

#include "host_math.h"

#define BK FHE_MAX_K

struct BehzDev {   // lives in device memory; every access is wave-uniform (scalar loads)
    u32 k;         // |q-base|; Bsk has k+1 primes, index k = m_sk
    Modulus q[BK], b[BK + 1];
    u64 r64q[BK];                          // floor(2^64 / q_i) for single-word reductions
    // every base-conversion multiplier is a context constant: (value, Shoup companion) pairs
    ulonglong2 mt_inv_punct[BK];           // m~ * (q/q_i)^-1 mod q_i
    ulonglong2 inv_punct[BK];              // (q/q_i)^-1 mod q_i
    ulonglong2 punct_q_mod_b[BK][BK + 1];  // (q/q_i) mod b_j
    u64 punct_q_mod_mt[BK];                // (q/q_i) mod 2^32
    u64 neg_inv_q_mod_mt;                  // -q^-1 mod 2^32
    ulonglong2 q_mod_b[BK + 1], inv_mt_mod_b[BK + 1], inv_q_mod_b[BK + 1];
    ulonglong2 t_mod_q[BK], t_mod_b[BK + 1];
    ulonglong2 inv_punct_B[BK];            // (B/b_j)^-1 mod b_j
    ulonglong2 punct_B_mod_q[BK][BK];      // (B/b_j) mod q_i
    ulonglong2 punct_B_mod_msk[BK];
    ulonglong2 inv_B_mod_msk;
    ulonglong2 B_mod_q[BK];
    // Lazy base conversions: every sum_i y_i * c_ij is accumulated as a 128-bit integer and reduced ONCE
    // (four 32x32 multiply-adds per term + seven for the reduction, against ten per Shoup product), with the
    // trailing constant factor of each step folded into the matrix entries:
    u64 mu2_q[BK], mu2_b[BK + 1];          // floor(2^(bits+63) / m): wide Barrett constant, valid for z < 2^(bits+63)
    u32 sh_q[BK], sh_b[BK + 1];            // bits - 1
    u64 ext_q2b[BK][BK + 1];               // (q/q_i) * m~^-1 mod b_j               (step 0+1)
    u64 ext_q_b[BK + 1];                   // q * m~^-1 mod b_j
    ulonglong2 t_inv_punct[BK];            // t * (q/q_i)^-1 mod q_i                 (step 3)
    u64 flo_q2b[BK][BK + 1];               // b_j - (q/q_i) * q^-1 mod b_j
    u64 flo_t_b[BK + 1];                   // t * q^-1 mod b_j
    u64 back_B2q[BK][BK];                  // (B/b_j) mod q_i                        (step 4)
    u64 back_pos[BK], back_neg[BK];        // q_i - B mod q_i,  B mod q_i
    u64 back_B2msk[BK];                    // (B/b_j) mod m_sk
};

typedef unsigned __int128 u128;

// Constants of k_behz_floor_back_pm (below): every modulus of the step is pseudo-Mersenne (ntt_core.h), every constant sits
// beside its multiple by the power of two its variable operand is split at.
constexpr int PM_GROUP_Y = 4;                        // y-terms per group of columns (with the D_b / remainder term in the first)
constexpr int PM_SPLIT_Y = 28, PM_SPLIT_Z = 29;      // y_i < 2^55 = yl + 2^28 yh;  z_j, folded D_b < 2^59 = zl + 2^29 zh
struct BehzPmDev {
    PmMod q[BK], b[BK + 1];
    ulonglong2 t_inv_punct[BK];            // (w, w 2^31 mod q_i): mul_pm
    ulonglong2 flo_t_b[BK + 1];            // (c, c 2^29 mod b_j)
    ulonglong2 flo_q2b[BK][BK + 1];        // (c, c 2^28 mod b_j)
    ulonglong2 inv_punct_B[BK];            // (w, w 2^31 mod b_j): mul_pm
    ulonglong2 back_B2msk[BK];             // (c, c 2^29 mod m_sk)
    ulonglong2 inv_B_mod_msk;              // (w, w 2^31 mod m_sk): mul_pm
    ulonglong2 mt_inv_punct[BK];           // (w, w 2^31 mod q_i): mul_pm            (k_behz_to_bsk_pm)
    ulonglong2 ext_q_b[BK + 1];            // (c, c 2^29 mod b_j)
    ulonglong2 ext_q2b[BK][BK + 1];        // (c, c 2^28 mod b_j)
    u64 punct_q_mod_mt[BK], neg_inv_q_mod_mt;
    ulonglong2 back_pos[BK], back_neg[BK]; // (c, c 2^29 mod q_i)
    ulonglong2 back_B2q[BK][BK];           // [j][i]: (c, c 2^29 mod q_i)
};

struct BehzTables {
    BaseTables aux;     // NTT tables of Bsk (k+1 primes)
    BehzDev host;       // host copy (k, moduli)
    BehzDev *dev = nullptr;
    int aux_bits = 61;
    bool wide_dot = false;   // 58-bit auxiliary primes and q-primes <= 58 bits: dot products of <= 8 terms need no inner reduction
    BehzPmDev *pm_dev = nullptr;     // tables of k_behz_floor_back_pm; null when a modulus does not qualify or a sum could overflow
};

namespace {

__device__ __forceinline__ u64 reduce64(u64 x, u64 q, u64 r64) {   // x mod q for any x < 2^64
    u64 r = x - __umul64hi(x, r64) * q;
    r = csub(r, 2 * q);
    return csub(r, q);
}

// z mod m for z < 2^(bits(m)+63): x = floor(z / 2^(bits-1)) < 2^64, qhat = floor(x mu2 / 2^64) with
// mu2 = floor(2^(bits+63) / m); z/m - qhat < 3, so the remainder estimate lies in [0, 3m)
__device__ __forceinline__ u64 reduce128(u128 z, u64 m, u64 mu2, u32 sh) {
    const u64 x = (u64)(z >> sh);
    u64 r = (u64)z - __umul64hi(x, mu2) * m;
    r = csub(r, 2 * m);
    return csub(r, m);
}

// Terms of a 128-bit dot product between reductions.  Every term is below 2^122 (operands below 2^61), the reduced
// carry below 2^61, and reduce128 takes z < 2^(bits(m)+63); three terms + carry stay below 2^124 for 61-bit moduli and,
// for a smaller modulus q_i, the terms z_j c (c < q_i) are below 2^(61+bits(q_i)): again three fit.
constexpr int DOT_CHUNK = 3;
// With auxiliary primes of 58 bits and q-primes of at most 58 bits no intermediate reduction is needed at all for up to
// 8 terms: a term is below 2^116, the start value below 2^116, and reduce128 takes z < 2^(bits(m)+63) -- 2^(58+63) for
// the auxiliary moduli (9 x 2^116 < 2^120), and for a q-prime of b bits the terms z_j c (c < q_i) are below 2^(58+b)
// against 2^(b+63): 32 of them fit.  BehzTables::wide_dot says when this holds; the kernels take the chunk as a
// template parameter (WIDE_CHUNK = no reduction inside a dot product of <= 8 terms).
constexpr int WIDE_CHUNK = 9;
// k_behz_to_bsk in WIDE mode writes its sums for the multiplier this chip has: every operand is below 2^58, so it
// splits into two 29-bit halves, and a dot product of <= 9 terms becomes three 64-bit columns
//   value = ll + mid 2^29 + hh 2^58,   ll = sum xl cl,  mid = sum (xl ch + xh cl),  hh = sum xh ch      (each < 2^63)
// that v_mad_u64_u32 accumulates directly: four multiply-adds per term and NO carry chain, where the 128-bit
// accumulator costs four multiply-adds plus a four-word add with carries through VCC per term.  The columns are joined
// once per sum; the integer is the same.  (In k_behz_floor_back the same form needs 148 VGPRs -- three waves per SIMD
// instead of four -- and measured 287 against 269 us per 256 products: it keeps its 128-bit accumulators.)
constexpr u64 M29 = (1ULL << 29) - 1;
struct Dot58 {
    u64 ll, mid, hh;
    __device__ __forceinline__ void init(u64 x, u64 c) {          // x, c < 2^58
        const u32 xl = (u32)(x & M29), xh = (u32)(x >> 29), cl = (u32)(c & M29), ch = (u32)(c >> 29);
        ll = (u64)xl * cl;
        mid = (u64)xl * ch + (u64)xh * cl;
        hh = (u64)xh * ch;
    }
    __device__ __forceinline__ void mac(u32 xl, u32 xh, u32 cl, u32 ch) {
        ll = (u64)xl * cl + ll;
        mid = (u64)xl * ch + mid;
        mid = (u64)xh * cl + mid;
        hh = (u64)xh * ch + hh;
    }
    __device__ __forceinline__ unsigned __int128 value() const {
        return (unsigned __int128)ll + ((unsigned __int128)mid << 29) + ((unsigned __int128)hh << 58);
    }
};

// x * c mod m for a context constant c = (value, Shoup companion); any x < 2^64; result in [0, m)
__device__ __forceinline__ u64 mulc(u64 x, const ulonglong2 c, u64 m) { return mul_shoup(x, c.x, c.y, m); }
// lazy variant, result in [0, 2m)
__device__ __forceinline__ u64 mulc_lazy(u64 x, const ulonglong2 c, u64 m) { return mul_shoup_lazy(x, c.x, c.y, m); }

// The base-conversion kernels are latency-bound, not multiply-bound, when their ~100 context constants come in
// through dependent scalar loads inside loops of run-time length.  They are therefore templated on the number of
// q-primes K (every loop unrolls, every table offset is a compile-time constant) and handle CPT coefficients per
// thread, so that every constant fetched serves CPT coefficients and the independent chains overlap.
// steps 0+1 for every coefficient of every input polynomial: in [polys][k][n] -> out [polys][k+1][n]
// CPT coefficients per thread: every table constant fetched (scalar loads) serves CPT coefficients.
constexpr int CPT = 4;          // floor/back kernel
#ifndef TO_BSK_CPT
#define TO_BSK_CPT 1
#endif
template <int K, int CPT, int CH>
__global__ __launch_bounds__(256) void k_behz_to_bsk(const u64 *__restrict__ in, u64 *__restrict__ out, const BehzDev *__restrict__ Tp, u32 n, u64 n_polys) {
    const BehzDev &T = *Tp;       // wave-uniform: scalar loads at compile-time offsets
    const u32 stride = gridDim.x * blockDim.x;             // n == CPT * stride
    const u32 c0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 p = blockIdx.y; p < n_polys; p += gridDim.y) {
        u64 y[CPT][K], r[CPT];
#pragma unroll
        for (int e = 0; e < CPT; e++) r[e] = 0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            const ulonglong2 w = T.mt_inv_punct[i];
            const u64 qi = T.q[i].q, pm = T.punct_q_mod_mt[i];
#pragma unroll
            for (int e = 0; e < CPT; e++) {
                y[e][i] = mul_shoup(in[(p * K + i) * n + c0 + e * stride], w.x, w.y, qi);      // canonical: used as an integer below
                r[e] += (y[e][i] & 0xffffffffULL) * pm;
            }
        }
        const u64 nq = T.neg_inv_q_mod_mt;
#pragma unroll
        for (int e = 0; e < CPT; e++) r[e] = ((r[e] & 0xffffffffULL) * nq) & 0xffffffffULL;
        if constexpr (CH == WIDE_CHUNK) {
            u32 yl[CPT][K], yh[CPT][K];
#pragma unroll
            for (int i = 0; i < K; i++) {
#pragma unroll
                for (int e = 0; e < CPT; e++) { yl[e][i] = (u32)(y[e][i] & M29); yh[e][i] = (u32)(y[e][i] >> 29); }
            }
#pragma unroll
            for (int j = 0; j <= K; j++) {
                const u64 bq = T.b[j].q, mu2 = T.mu2_b[j], eq = T.ext_q_b[j];
                const u32 sh = T.sh_b[j];
                Dot58 d[CPT];
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    const u64 rb = r[e] >= 0x80000000ULL ? r[e] + bq - 0x100000000ULL : r[e];    // centred remainder
                    d[e].init(rb, eq);
                }
#pragma unroll
                for (int i = 0; i < K; i++) {
                    const u64 cij = T.ext_q2b[i][j];
                    const u32 cl = (u32)(cij & M29), ch = (u32)(cij >> 29);
#pragma unroll
                    for (int e = 0; e < CPT; e++) d[e].mac(yl[e][i], yh[e][i], cl, ch);
                }
#pragma unroll
                for (int e = 0; e < CPT; e++) out[(p * (K + 1) + j) * n + c0 + e * stride] = reduce128(d[e].value(), bq, mu2, sh);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j <= K; j++) {
            const u64 bq = T.b[j].q, mu2 = T.mu2_b[j], eq = T.ext_q_b[j];
            const u32 sh = T.sh_b[j];
            u128 acc[CPT];
#pragma unroll
            for (int e = 0; e < CPT; e++) {
                const u64 rb = r[e] >= 0x80000000ULL ? r[e] + bq - 0x100000000ULL : r[e];    // centred remainder
                acc[e] = (u128)rb * eq;
            }
#pragma unroll
            for (int i = 0; i < K; i++) {
                const u64 cij = T.ext_q2b[i][j];
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    if (i > 0 && i % CH == 0) acc[e] = reduce128(acc[e], bq, mu2, sh);
                    acc[e] += (u128)y[e][i] * cij;
                }
            }
#pragma unroll
            for (int e = 0; e < CPT; e++) out[(p * (K + 1) + j) * n + c0 + e * stride] = reduce128(acc[e], bq, mu2, sh);
        }
    }
}

// Which entry of the B operand batch pair c multiplies: c itself, or -- for an operand batch shared between pairs
// (fhe_multiply_prepared_shared: one xfract per output column, one yfract per output row) -- (c / div) % cnt.
struct BMap {
    u64 div, cnt, off;      // cnt == 0: identity; off: absolute number of the launch's first pair
    __device__ __forceinline__ u64 operator()(u64 c) const { return cnt ? ((c + off) / div) % cnt : c; }
};
// tensor product in NTT form over one base: A [count][sa][nb][n], Bm [count][sb][nb][n] -> D [count][sa+sb-1][nb][n]
__global__ __launch_bounds__(256) void k_behz_tensor(const u64 *__restrict__ A, const u64 *__restrict__ Bm, u64 *__restrict__ D,
                                                     const Modulus *__restrict__ mods, const u64 *__restrict__ mu2, u32 nb, u32 n, u32 sa, u32 sb, u64 count,
                                                     BMap bm) {
    const u32 so = sa + sb - 1;
    for (u64 cp = blockIdx.y; cp < count * nb; cp += gridDim.y) {
        const u64 c = cp / nb, cb = bm(c);
        const u32 j = (u32)(cp % nb);
        const Modulus m = mods[j];
        for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
            for (u32 o = 0; o < so; o++) {
                u128 acc = 0;
                u32 terms = 0;
                const u32 lo = o >= sb ? o - sb + 1 : 0, hi = o < sa ? o : sa - 1;
                for (u32 ja = lo; ja <= hi; ja++) {
                    const u64 x = A[((c * sa + ja) * nb + j) * n + s], y = Bm[((cb * sb + (o - ja)) * nb + j) * n + s];
                    acc += (u128)x * y;                               // below 2^122 each; reduced every three terms
                    if (++terms == 3) { acc = reduce128(acc, m.q, mu2[j], m.s1); terms = 0; }
                }
                D[((c * so + o) * nb + j) * n + s] = reduce128(acc, m.q, mu2[j], m.s1);
            }
        }
    }
}

// tensor product fused into the inverse transform: one workgroup forms output polynomial o of
// ciphertext pair c at base prime j from the NTT-form operands and transforms it back, so the
// NTT-form product never goes to memory.  A [count][sa][nb][n], Bm [count][sb][nb][n] (NTT order)
// -> D [count][sa+sb-1][nb][n] (coefficient form, canonical).
// WIDE (every modulus of the base <= 58 bits, at most 12 terms): the products stay in [0, 5q) (mul_barrett_lazy5) and are
// summed as plain integers (12 x 5q < 2^64); the sum is brought below 4q -- the range the inverse transform takes -- by
// at most four conditional subtractions per output instead of three per term.
__device__ __forceinline__ u64 fold_below_4q(u64 acc, u32 terms, u64 q4) {     // acc < 5 terms q, terms <= 12
#pragma unroll
    for (int s = 3; s >= 0; s--)
        if (5 * terms > (4u << s)) acc = csub(acc, q4 << s);
    return acc;
}
template <int L, bool WIDE>
__global__ __launch_bounds__(NttShape<L>::TP) void k_behz_tensor_intt(const u64 *__restrict__ A, const u64 *__restrict__ Bm, u64 *__restrict__ D,
                                                                       RnsBase base, u32 sa, u32 sb, u64 groups, BMap bm) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u32 nb = base.count, so = sa + sb - 1;
    // Work order: the `so` output polynomials of one (pair c, prime j) read the same sa + sb operand polynomials
    // (an operand enters up to min(sa, sb) outputs), so their workgroups are placed 8 apart in blockIdx -- same XCD,
    // dispatched back to back -- and the repeated reads hit that XCD's L2 instead of going to memory again.
    const u64 bid = blockIdx.x, chunk = bid / (8 * so), rem = bid % (8 * so);
    const u32 o = (u32)(rem >> 3);
    const u64 g = chunk * 8 + (rem & 7);               // c * nb + j
    if (g >= groups) return;
    const u32 j = (u32)(g % nb);
    const u64 c = g / nb;
    const u64 id = (c * so + o) * nb + j;              // output polynomial
    const Modulus m = base.mod[j];
    const BarrettLazy bl = barrett_lazy(m);
    u64 acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0;
    const u32 lo = o >= sb ? o - sb + 1 : 0, hi = o < sa ? o : sa - 1;
    for (u32 ja = lo; ja <= hi; ja++) {
        u64 xa[16], xb[16];
        load_slots<L>(xa, A + ((c * sa + ja) * nb + j) * N, tid);
        load_slots<L>(xb, Bm + ((bm(c) * sb + (o - ja)) * nb + j) * N, tid);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = WIDE ? acc[r] + mul_barrett_lazy5(xa[r], xb[r], bl) : addmod(acc[r], mul_barrett(xa[r], xb[r], m), m.q);
    }
    if constexpr (WIDE) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = fold_below_4q(acc[r], hi - lo + 1, 4 * m.q);
    }
    ntt_inv_regs4<L, WIDE>(acc, base.itw + (size_t)j * N, ntt_mod(m.q), lds, tid);               // [0, 4q) in, [0, 4q) out
    // WIDE: the only reader is k_behz_floor_back<.., WIDE_CHUNK>, whose Shoup products take any 64-bit value and whose
    // 128-bit sums have room for a start value below 4 b_j (2^118 + 8 x 2^116 < 2^121): no canonical form needed
    if constexpr (!WIDE) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = csub(csub(acc[r], 2 * m.q), m.q);
    }
    store_coeff<L>(acc, D + id * N, tid);
}

// The same for TWO ciphertext pairs (2 cc, 2 cc + 1) per workgroup at n >= 8192: the two output polynomials share the
// prime, so every twiddle pair of the inverse transform is fetched once for both (ntt_inv_regs4m).  groups = (count / 2) * nb.
template <int L, bool WIDE>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_behz_tensor_intt2(const u64 *__restrict__ A, const u64 *__restrict__ Bm, u64 *__restrict__ D,
                                                                           RnsBase base, u32 sa, u32 sb, u64 groups, BMap bm) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid = threadIdx.x;
    const u32 nb = base.count, so = sa + sb - 1;
    const u64 bid = blockIdx.x, chunk = bid / (8 * so), rem = bid % (8 * so);      // same XCD-aware order as k_behz_tensor_intt
    const u32 o = (u32)(rem >> 3);
    const u64 g = chunk * 8 + (rem & 7);               // cc * nb + j
    if (g >= groups) return;
    const u32 j = (u32)(g % nb);
    const u64 cc = g / nb;
    const Modulus m = base.mod[j];
    const BarrettLazy bl = barrett_lazy(m);
    u64 acc[2][16];
    const u32 lo = o >= sb ? o - sb + 1 : 0, hi = o < sa ? o : sa - 1;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const u64 c = 2 * cc + h, cb = bm(c);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[h][r] = 0;
        for (u32 ja = lo; ja <= hi; ja++) {
            const u64 *pa = A + ((c * sa + ja) * nb + j) * N + tid, *pb = Bm + ((cb * sb + (o - ja)) * nb + j) * N + tid;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {       // eight slots at a time: 64 accumulator + 32 operand VGPRs
                u64 xa[8], xb[8];
#pragma unroll
                for (int r = 0; r < 8; r++) { xa[r] = pa[(r0 + r) * TP]; xb[r] = pb[(r0 + r) * TP]; }
#pragma unroll
                for (int r = 0; r < 8; r++)
                    acc[h][r0 + r] = WIDE ? acc[h][r0 + r] + mul_barrett_lazy5(xa[r], xb[r], bl) : addmod(acc[h][r0 + r], mul_barrett(xa[r], xb[r], m), m.q);
            }
        }
        if constexpr (WIDE) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[h][r] = fold_below_4q(acc[h][r], hi - lo + 1, 4 * m.q);
        }
    }
    ntt_inv_regs4m<L, 2, WIDE>(acc, base.itw + (size_t)j * N, ntt_mod(m.q), lds, tid);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if constexpr (!WIDE) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[h][r] = csub(csub(acc[h][r], 2 * m.q), m.q);
        }
        store_coeff<L>(acc[h], D + (((2 * cc + h) * so + o) * nb + j) * N, tid);
    }
}

// The same step on the pseudo-Mersenne inverse passes (ntt_core.h; C = the class of the base), M ciphertext pairs
// (M cc + h) per workgroup.  The products (mulvv_pm: seven multiply-adds, each below C::RQ / 16 q <= 6q) are summed as
// integers (at most 12 terms: 72 q < 2^62 on a 55-bit base, 18 q on a 58-bit one), one fold_pm brings a sum below (17/16) q, and the transform leaves its outputs below C::RQ / 16 q: k_behz_floor_back<.., WIDE_CHUNK> takes them
// as they are (its Shoup products accept any 64-bit value, its 128-bit sums a start value below 4 b_j).
// SQ: the product of a batch with itself (A == Bm, equal sizes, identity map): a_i a_j and a_j a_i are the same integer, so each
// cross term is formed once and added twice -- the same integer sum, half of a square's slot products and operand loads.  A
// separate instantiation: as a run-time flag the branch cost every product its operand prefetch (profiles/EXPERIMENTS.md).
template <int L, int M, typename C, bool SQ>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_behz_tensor_intt_pm(const u64 *__restrict__ A, const u64 *__restrict__ Bm, u64 *__restrict__ D,
                                                                             RnsBase base, u32 sa, u32 sb, u64 groups, BMap bm) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid = threadIdx.x;
    const u32 nb = base.count, so = sa + sb - 1;
    const u64 bid = blockIdx.x, chunk = bid / (8 * so), rem = bid % (8 * so);      // same XCD-aware order as k_behz_tensor_intt
    const u32 o = (u32)(rem >> 3);
    const u64 g = chunk * 8 + (rem & 7);               // cc * nb + j
    if (g >= groups) return;
    const u32 j = (u32)(g % nb);
    const u64 cc = g / nb;
    const PmMod m = base.pm[j];
    u64 acc[M][16];
    const u32 lo = o >= sb ? o - sb + 1 : 0;
    const u32 hi = SQ ? (o + 1) / 2 : (o < sa ? o + 1 : sa);       // one past the last term; SQ: the cross terms ja < o - ja
#pragma unroll
    for (int h = 0; h < M; h++) {
        const u64 c = M * cc + h, cb = SQ ? c : bm(c);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[h][r] = 0;
        // all 32 operand loads of a term in flight before its first product (M = 1: 32 accumulator + 64 operand VGPRs);
        // left to itself hipcc keeps two or three in flight and every pair of slots waits out a memory round trip
        constexpr int G = M == 1 ? 16 : 8;
        for (u32 ja = lo; ja < hi; ja++) {
            const u64 *pa = A + ((c * sa + ja) * nb + j) * N + tid, *pb = Bm + ((cb * sb + (o - ja)) * nb + j) * N + tid;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += G) {
                u64 xa[G], xb[G];
#pragma unroll
                for (int r = 0; r < G; r++) { xa[r] = pa[(r0 + r) * TP]; xb[r] = pb[(r0 + r) * TP]; }
                PM_FENCE();
#pragma unroll
                for (int r = 0; r < G; r++) acc[h][r0 + r] += mulvv_pm(xa[r], xb[r], m);      // canonical operands; below C::RQ / 16 q
            }
        }
        if constexpr (SQ) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[h][r] += acc[h][r];                               // every cross term counts twice
            if (!(o & 1)) {                                                                    // the middle term a_(o/2)^2, once
                const u64 *pa = A + ((c * sa + o / 2) * nb + j) * N + tid;
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += G) {
                    u64 xa[G];
#pragma unroll
                    for (int r = 0; r < G; r++) xa[r] = pa[(r0 + r) * TP];
                    PM_FENCE();
#pragma unroll
                    for (int r = 0; r < G; r++) acc[h][r0 + r] += mulvv_pm(xa[r], xa[r], m);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) acc[h][r] = fold_pm(acc[h][r], m);
    }
    ntt_inv_regs_pm<L, M, PM_FOLDED, C::XB, C::LIM, C::RQ>(acc, base.itw_pm + (size_t)j * N, m, lds, tid);
#pragma unroll
    for (int h = 0; h < M; h++) store_coeff<L>(acc[h], D + (((M * cc + h) * so + o) * nb + j) * N, tid);
}

// steps 2(tail: times t) + 3 + 4: Dq [polys][k][n], Db [polys][k+1][n] (coefficient form) -> out [polys][k][n]
template <int K, int CH>
__global__ __launch_bounds__(256) void k_behz_floor_back(const u64 *__restrict__ Dq, const u64 *__restrict__ Db, u64 *__restrict__ out,
                                                         const BehzDev *__restrict__ Tp, u32 n, u64 n_polys) {
    const BehzDev &T = *Tp;       // wave-uniform: scalar loads at compile-time offsets
    const u64 msk = T.b[K].q;
    const u32 stride = gridDim.x * blockDim.x;             // n == CPT * stride
    const u32 c0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 p = blockIdx.y; p < n_polys; p += gridDim.y) {
        u64 y[CPT][K], f[CPT][K + 1], z[CPT][K];
#pragma unroll
        for (int i = 0; i < K; i++) {
            const ulonglong2 w = T.t_inv_punct[i];
            const u64 qi = T.q[i].q;
#pragma unroll
            for (int e = 0; e < CPT; e++) y[e][i] = mul_shoup(Dq[(p * K + i) * n + c0 + e * stride], w.x, w.y, qi);   // [t D]_q (q/q_i)^-1, canonical
        }
#pragma unroll
        for (int j = 0; j <= K; j++) {         // fast floor: (t D - FastBConv([t D]_q)) q^-1 in Bsk, one 128-bit sum per prime
            const u64 bq = T.b[j].q, mu2 = T.mu2_b[j], ft = T.flo_t_b[j];
            const u32 sh = T.sh_b[j];
            u128 acc[CPT];
#pragma unroll
            for (int e = 0; e < CPT; e++) acc[e] = (u128)Db[(p * (K + 1) + j) * n + c0 + e * stride] * ft;
#pragma unroll
            for (int i = 0; i < K; i++) {
                const u64 cij = T.flo_q2b[i][j];
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    if (i > 0 && i % CH == 0) acc[e] = reduce128(acc[e], bq, mu2, sh);
                    acc[e] += (u128)y[e][i] * cij;
                }
            }
#pragma unroll
            for (int e = 0; e < CPT; e++) f[e][j] = reduce128(acc[e], bq, mu2, sh);
        }
        u64 a_abs[CPT];
        bool neg[CPT];
        {
            const u64 mu2 = T.mu2_b[K];
            const u32 sh = T.sh_b[K];
            u128 acc[CPT];
#pragma unroll
            for (int e = 0; e < CPT; e++) acc[e] = 0;
#pragma unroll
            for (int j = 0; j < K; j++) {
                const ulonglong2 w = T.inv_punct_B[j];
                const u64 bq = T.b[j].q, cj = T.back_B2msk[j];
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    z[e][j] = mul_shoup(f[e][j], w.x, w.y, bq);                              // canonical integer in [0, b_j)
                    if (j > 0 && j % CH == 0) acc[e] = reduce128(acc[e], msk, mu2, sh);
                    acc[e] += (u128)z[e][j] * cj;
                }
            }
            const ulonglong2 ib = T.inv_B_mod_msk;
#pragma unroll
            for (int e = 0; e < CPT; e++) {
                const u64 conv_sk = reduce128(acc[e], msk, mu2, sh);
                const u64 alpha = mul_shoup(conv_sk + msk - f[e][K], ib.x, ib.y, msk);
                neg[e] = alpha > (msk >> 1);
                a_abs[e] = neg[e] ? msk - alpha : alpha;                                     // |alpha_sk| <= k
            }
        }
#pragma unroll
        for (int i = 0; i < K; i++) {
            const u64 qi = T.q[i].q, mu2 = T.mu2_q[i], bn = T.back_neg[i], bp = T.back_pos[i];
            const u32 sh = T.sh_q[i];
            u128 acc[CPT];
#pragma unroll
            for (int e = 0; e < CPT; e++) acc[e] = (u128)a_abs[e] * (neg[e] ? bn : bp);
#pragma unroll
            for (int j = 0; j < K; j++) {
                const u64 cji = T.back_B2q[j][i];
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    if (j > 0 && j % CH == 0) acc[e] = reduce128(acc[e], qi, mu2, sh);
                    acc[e] += (u128)z[e][j] * cji;
                }
            }
#pragma unroll
            for (int e = 0; e < CPT; e++) out[(p * K + i) * n + c0 + e * stride] = reduce128(acc[e], qi, mu2, sh);
        }
    }
}

// The same step on pseudo-Mersenne arithmetic (ntt_core.h).  A dot product sum_t x_t c_t mod m with context constants c_t
// is kept as TWO 64-bit columns, S = A + 2^32 B with
//     A = sum xl cl + xh c2l,   B = sum xl ch + xh c2h      (x = xl + 2^s xh, c2 = c 2^s mod m; four v_mad_u64_u32 per term)
// -- no carry chain and no 128-bit accumulator -- and reduced once: S = zl + 2^b zh, result zl + zh delta (one
// multiply-add), where the 128-bit form spends a four-word add per term and a seven-multiply Barrett reduction per sum.
// Products with one constant are mul_pm; canonical values (the y_i, z_j, alpha and the outputs -- SEAL's fast base
// conversions take the canonical residues, and bit-exactness rests on that) are fold_pm + one conditional subtraction.
// The host checks with the actual constants that no column can overflow and that every zh stays below 2^32
// (behz_pm_tables); sums over the z_j go two terms at a time for that reason.  PCPT coefficients per thread.
struct PmAcc { u64 A, B; };
__device__ __forceinline__ void pm_mac(PmAcc &s, u32 xl, u32 xh, const ulonglong2 c) {
    s.A = (u64)xl * (u32)c.x + s.A;
    s.A = (u64)xh * (u32)c.y + s.A;
    s.B = (u64)xl * (u32)(c.x >> 32) + s.B;
    s.B = (u64)xh * (u32)(c.y >> 32) + s.B;
}
__device__ __forceinline__ u64 pm_acc_reduce(const PmAcc &s, const PmMod &m) {
    const u64 B = s.B + (s.A >> 32);
    const u32 zh = __builtin_amdgcn_alignbit((u32)(B >> 32), (u32)B, m.sh);
    u64 zl = pm_pack((u32)s.A, (u32)B & m.mb);
    asm("" : "+v"(zl));
    return (u64)zh * m.delta + zl;
}
__device__ __forceinline__ u64 canon_fold_pm(u64 v, const PmMod &m) { return csub(fold_pm(v, m), m.q); }
constexpr int PCPT = 2;
// steps 0 + 1 the same way: in [polys][k][n] (canonical) -> out [polys][k+1][n] (canonical: the forward transforms take it so)
template <int K>
__global__ __launch_bounds__(256) void k_behz_to_bsk_pm(const u64 *__restrict__ in, u64 *__restrict__ out, const BehzPmDev *__restrict__ Tp, u32 n, u64 n_polys) {
    const BehzPmDev &T = *Tp;
    const u32 stride = gridDim.x * blockDim.x;             // n == PCPT * stride
    const u32 c0 = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr u32 MY = (1u << PM_SPLIT_Y) - 1, MZ = (1u << PM_SPLIT_Z) - 1;
    for (u64 p = blockIdx.y; p < n_polys; p += gridDim.y) {
        u32 yl[PCPT][K], yh[PCPT][K];
        u64 r[PCPT];
#pragma unroll
        for (int e = 0; e < PCPT; e++) r[e] = 0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            const PmMod m = T.q[i];
            const ulonglong2 w = T.mt_inv_punct[i];
            const u64 pm = T.punct_q_mod_mt[i];
#pragma unroll
            for (int e = 0; e < PCPT; e++) {
                const u64 y = canon_fold_pm(mul_pm(in[(p * K + i) * n + c0 + e * stride], w, m), m);      // canonical: used as an integer below
                r[e] += (y & 0xffffffffULL) * pm;
                yl[e][i] = (u32)y & MY;
                yh[e][i] = (u32)(y >> PM_SPLIT_Y);
            }
        }
        const u64 nq = T.neg_inv_q_mod_mt;
#pragma unroll
        for (int e = 0; e < PCPT; e++) r[e] = ((r[e] & 0xffffffffULL) * nq) & 0xffffffffULL;
#pragma unroll
        for (int j = 0; j <= K; j++) {
            const PmMod m = T.b[j];
            const ulonglong2 eq = T.ext_q_b[j];
            PmAcc acc[PCPT];
            u64 tot[PCPT];
#pragma unroll
            for (int e = 0; e < PCPT; e++) {
                const u64 rb = r[e] >= 0x80000000ULL ? r[e] + m.q - 0x100000000ULL : r[e];    // centred remainder
                acc[e].A = 0; acc[e].B = 0;
                tot[e] = 0;
                pm_mac(acc[e], (u32)rb & MZ, (u32)(rb >> PM_SPLIT_Z), eq);
            }
#pragma unroll
            for (int i = 0; i < K; i++) {
                const ulonglong2 c = T.ext_q2b[i][j];
#pragma unroll
                for (int e = 0; e < PCPT; e++) {
                    if (i > 0 && i % PM_GROUP_Y == 0) { tot[e] += pm_acc_reduce(acc[e], m); acc[e].A = 0; acc[e].B = 0; }      // K > 4: a second group of columns
                    pm_mac(acc[e], yl[e][i], yh[e][i], c);
                }
            }
#pragma unroll
            for (int e = 0; e < PCPT; e++) out[(p * (K + 1) + j) * n + c0 + e * stride] = canon_fold_pm(tot[e] + pm_acc_reduce(acc[e], m), m);
        }
    }
}
// Steps 0 + 1 FUSED INTO THE FORWARD TRANSFORMS (round 4; opt-in, FHE_BEHZ_FUSED_PREPARE=1): one launch prepares an operand in
// both bases.  MEASURED (P8192, 256 products, profiles/EXPERIMENTS.md): HBM traffic per 2x2 product 14.6 MB -> 11.0 MB (7.9x ->
// 6.0x the algorithmic bytes), launch 274 us against 57 + 102 + 78 us for the three launches it replaces, multiply 2x2 -5 %:
// the product's kernels are VALU-issue-bound at 0.5-0.6 of their issue floor, so the recomputed y_i cost more than the saved
// bytes return.  The default therefore stays k_behz_to_bsk_pm + two transform launches; this kernel is kept for parameter
// sets / chips where the balance differs, and parity-tested (tests/test_gpu_parity.py fallback switches).  A workgroup is
// (input polynomial p, role r): role r < K + 1 is auxiliary prime b_r -- it reads the K residue polynomials x_i of p at its
// threads' sixteen coefficient positions, forms the canonical y_i, the small-Montgomery remainder and ITS column of the
// base extension exactly as k_behz_to_bsk_pm does (same constants, same grouping: same integers), and runs straight into the
// forward transform mod b_r with the extended residues still in registers; role K + 1 + i is the plain forward transform
// of x_i mod q_i.  The extended polynomials never exist in memory in coefficient form: per input polynomial the separate
// launches moved K + (K+1) [k_behz_to_bsk_pm] + 2 (K+1) + 2 K [two transform launches] = 6 K + 3 residue polynomials
// through HBM, this one K (read once: the 2 K + 1 workgroups of p sit 8 apart in blockIdx -- same XCD, dispatched back to
// back, the re-reads hit its L2 -- same placement as k_behz_tensor_intt) + 2 K + 1 written.  The price is the y_i
// computed K + 1 times instead of once (K mul_pm + canonicalisation per coefficient and auxiliary prime).
// Sixteen coefficients per thread in two halves of eight, so that the two-column accumulators of a half (32 VGPRs), the
// operand of the term in flight (16) and the finished residues (32) stay inside the transform's register budget.
template <int L, int K, typename CQ, typename CB>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_behz_prepare_pm(const u64 *__restrict__ in, u64 *__restrict__ xq, u64 *__restrict__ xb,
                                                                         RnsBase qbase, RnsBase bbase, const BehzPmDev *__restrict__ Tp, u64 n_polys) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, R = 2 * K + 1;
    const int tid = threadIdx.x;
    const u64 bid = blockIdx.x, chunk = bid / (8 * R), rem = bid % (8 * R);
    const u32 role = (u32)(rem >> 3);
    const u64 p = chunk * 8 + (rem & 7);
    if (p >= n_polys) return;
    u64 x[1][16];
    if (role > K) {                                         // q-base: the plain forward transform (k_ntt_fwd_pm)
        const u32 i = role - (K + 1);
        const PmMod m = qbase.pm[i];
        load_coeff<L>(x[0], in + (p * K + i) * N, tid);
        ntt_fwd_regs_pm<L, 1, 16, CQ::LIM, CQ::CS>(x, qbase.tw_pm + (size_t)i * N, m, lds, tid);
#pragma unroll
        for (int r = 0; r < 16; r++) x[0][r] = canon_pm(x[0][r], m);
        store_slots<L>(x[0], xq + (p * K + i) * N, tid);
        return;
    }
    const BehzPmDev &T = *Tp;
    const u32 j = role;
    const PmMod mb = T.b[j];
    constexpr u32 MY = (1u << PM_SPLIT_Y) - 1, MZ = (1u << PM_SPLIT_Z) - 1;
    // the 2 K groups of eight operand loads (half h, residue i) run one group AHEAD of the arithmetic: a group's round trip to
    // L2 / HBM is covered by the previous group's products instead of being waited out 2 K times per workgroup
    u64 v[2][8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[0][e] = in[(p * K + 0) * N + elem_index<L - 4>(tid, e)];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        PmAcc acc[8];
        u64 tot[8], rm[8];                                  // rm: the small-Montgomery column mod 2^32 (only its low word is used)
#pragma unroll
        for (int e = 0; e < 8; e++) { acc[e].A = 0; acc[e].B = 0; tot[e] = 0; rm[e] = 0; }
#pragma unroll
        for (int i = 0; i < K; i++) {
            const int g = h * K + i;                        // this group; the next one is (h, i + 1) or (1, 0)
            if (g + 1 < 2 * K) {
                const int hn = (i + 1 < K) ? h : 1, in_ = (i + 1 < K) ? i + 1 : 0;
#pragma unroll
                for (int e = 0; e < 8; e++) v[(g + 1) & 1][e] = in[(p * K + in_) * N + elem_index<L - 4>(tid, 8 * hn + e)];
            }
            PM_FENCE();
            const PmMod m = T.q[i];
            const ulonglong2 w = T.mt_inv_punct[i], c = T.ext_q2b[i][j];
            const u32 pm = (u32)T.punct_q_mod_mt[i];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (i > 0 && i % PM_GROUP_Y == 0) { tot[e] += pm_acc_reduce(acc[e], mb); acc[e].A = 0; acc[e].B = 0; }      // K > 4: a second group of columns
                const u64 y = canon_fold_pm(mul_pm(v[g & 1][e], w, m), m);                    // canonical: used as an integer below
                rm[e] = (u64)(u32)y * pm + rm[e];                                              // one multiply-add; mod 2^32 at the end
                pm_mac(acc[e], (u32)y & MY, (u32)(y >> PM_SPLIT_Y), c);
            }
            PM_FENCE();
        }
        const u32 nq = (u32)T.neg_inv_q_mod_mt;
        const ulonglong2 eq = T.ext_q_b[j];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const u64 r = (u64)((u32)rm[e] * nq);
            const u64 rb = r >= 0x80000000ULL ? r + mb.q - 0x100000000ULL : r;                 // centred remainder
            pm_mac(acc[e], (u32)rb & MZ, (u32)(rb >> PM_SPLIT_Z), eq);
            x[0][8 * h + e] = canon_fold_pm(tot[e] + pm_acc_reduce(acc[e], mb), mb);           // the value k_behz_to_bsk_pm stores
        }
    }
    ntt_fwd_regs_pm<L, 1, 16, CB::LIM, CB::CS>(x, bbase.tw_pm + (size_t)j * N, mb, lds, tid);
#pragma unroll
    for (int r = 0; r < 16; r++) x[0][r] = canon_pm(x[0][r], mb);
    store_slots<L>(x[0], xb + (p * (K + 1) + j) * N, tid);
}

// steps 2(tail: times t) + 3 + 4 for PCPT coefficients (c0, c0 + stride) of ONE output polynomial: Dq -> its k residue polynomials
// (coefficient form, any values the inverse transform leaves), Db -> its k + 1; res = the canonical residues of the product
template <int K>
__device__ __forceinline__ void floor_back_pm_at(const BehzPmDev &T, const u64 *__restrict__ Dq, const u64 *__restrict__ Db, u32 n, u32 c0, u32 stride,
                                                 u64 (&res)[PCPT][K]) {
    constexpr u32 MY = (1u << PM_SPLIT_Y) - 1, MZ = (1u << PM_SPLIT_Z) - 1;
    u32 yl[PCPT][K], yh[PCPT][K], zl[PCPT][K], zh[PCPT][K];
    u64 f[PCPT][K + 1];
#pragma unroll
    for (int i = 0; i < K; i++) {          // [t D]_q (q/q_i)^-1, canonical
        const PmMod m = T.q[i];
        const ulonglong2 w = T.t_inv_punct[i];
#pragma unroll
        for (int e = 0; e < PCPT; e++) {
            const u64 y = canon_fold_pm(mul_pm(Dq[(size_t)i * n + c0 + e * stride], w, m), m);
            yl[e][i] = (u32)y & MY;
            yh[e][i] = (u32)(y >> PM_SPLIT_Y);
        }
    }
#pragma unroll
    for (int j = 0; j <= K; j++) {         // fast floor: (t D - FastBConv([t D]_q)) q^-1 in Bsk, one two-column sum per prime
        const PmMod m = T.b[j];
        const ulonglong2 ft = T.flo_t_b[j];
        PmAcc acc[PCPT];
#pragma unroll
        for (int e = 0; e < PCPT; e++) {
            const u64 d = fold_pm(Db[(size_t)j * n + c0 + e * stride], m);       // any 64-bit value -> below (17/16) b_j
            acc[e].A = 0; acc[e].B = 0;
            f[e][j] = 0;
            pm_mac(acc[e], (u32)d & MZ, (u32)(d >> PM_SPLIT_Z), ft);
        }
#pragma unroll
        for (int i = 0; i < K; i++) {
            const ulonglong2 c = T.flo_q2b[i][j];
#pragma unroll
            for (int e = 0; e < PCPT; e++) {
                if (i > 0 && i % PM_GROUP_Y == 0) { f[e][j] += pm_acc_reduce(acc[e], m); acc[e].A = 0; acc[e].B = 0; }      // K > 4: a second group of columns
                pm_mac(acc[e], yl[e][i], yh[e][i], c);
            }
        }
#pragma unroll
        for (int e = 0; e < PCPT; e++) f[e][j] += pm_acc_reduce(acc[e], m);              // below 1.5 b_j per group: 3 b_j
    }
    const PmMod mk = T.b[K];
    u64 conv[PCPT];
#pragma unroll
    for (int e = 0; e < PCPT; e++) conv[e] = 0;
#pragma unroll
    for (int j0 = 0; j0 < K; j0 += 2) {
        PmAcc acc[PCPT];
#pragma unroll
        for (int e = 0; e < PCPT; e++) { acc[e].A = 0; acc[e].B = 0; }
#pragma unroll
        for (int j = j0; j < j0 + 2 && j < K; j++) {
            const PmMod m = T.b[j];
            const ulonglong2 w = T.inv_punct_B[j], cj = T.back_B2msk[j];
#pragma unroll
            for (int e = 0; e < PCPT; e++) {
                const u64 z = canon_fold_pm(mul_pm(f[e][j], w, m), m);                   // canonical integer in [0, b_j)
                zl[e][j] = (u32)z & MZ;
                zh[e][j] = (u32)(z >> PM_SPLIT_Z);
                pm_mac(acc[e], zl[e][j], zh[e][j], cj);
            }
        }
#pragma unroll
        for (int e = 0; e < PCPT; e++) conv[e] += pm_acc_reduce(acc[e], mk);             // each below 1.5 m_sk
    }
    u32 al[PCPT], ah[PCPT];                // |alpha_sk| (at most k for a product; split like the z_j all the same)
    bool neg[PCPT];
    {
        const ulonglong2 ib = T.inv_B_mod_msk;
#pragma unroll
        for (int e = 0; e < PCPT; e++) {
            const u64 alpha = canon_fold_pm(mul_pm(conv[e] + 4 * mk.q - f[e][K], ib, mk), mk);      // f_K below 3 m_sk, conv below 6 m_sk: below 2^62
            neg[e] = alpha > (mk.q >> 1);
            const u64 a_abs = neg[e] ? mk.q - alpha : alpha;
            al[e] = (u32)a_abs & MZ;
            ah[e] = (u32)(a_abs >> PM_SPLIT_Z);
        }
    }
#pragma unroll
    for (int i = 0; i < K; i++) {
        const PmMod m = T.q[i];
        const ulonglong2 bn = T.back_neg[i], bp = T.back_pos[i];
        u64 tot[PCPT];
#pragma unroll
        for (int e = 0; e < PCPT; e++) {
            PmAcc a0;
            a0.A = 0; a0.B = 0;
            pm_mac(a0, al[e], ah[e], neg[e] ? bn : bp);
            tot[e] = pm_acc_reduce(a0, m);
        }
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += 2) {
            PmAcc acc[PCPT];
#pragma unroll
            for (int e = 0; e < PCPT; e++) { acc[e].A = 0; acc[e].B = 0; }
#pragma unroll
            for (int j = j0; j < j0 + 2 && j < K; j++) {
                const ulonglong2 cji = T.back_B2q[j][i];
#pragma unroll
                for (int e = 0; e < PCPT; e++) pm_mac(acc[e], zl[e][j], zh[e][j], cji);
            }
#pragma unroll
            for (int e = 0; e < PCPT; e++) tot[e] += pm_acc_reduce(acc[e], m);
        }
#pragma unroll
        for (int e = 0; e < PCPT; e++) res[e][i] = canon_fold_pm(tot[e], m);
    }
}
template <int K>
__global__ __launch_bounds__(256) void k_behz_floor_back_pm(const u64 *__restrict__ Dq, const u64 *__restrict__ Db, u64 *__restrict__ out,
                                                            const BehzPmDev *__restrict__ Tp, u32 n, u64 n_polys) {
    const BehzPmDev &T = *Tp;     // wave-uniform: scalar loads at compile-time offsets
    const u32 stride = gridDim.x * blockDim.x;             // n == PCPT * stride
    const u32 c0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 p = blockIdx.y; p < n_polys; p += gridDim.y) {
        u64 res[PCPT][K];
        floor_back_pm_at<K>(T, Dq + p * K * n, Db + p * (K + 1) * n, n, c0, stride, res);
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
            for (int e = 0; e < PCPT; e++) out[(p * K + i) * n + c0 + e * stride] = res[e][i];
    }
}

// Cubic's tail fused into the floor / back conversion of its three products (homo/fhe_resize.h:176-188: a t3, b t2, c t, their sum
// times encode(0.5) = x^-1, plus B): per output coefficient the three products' residues are formed, added, rotated by one
// position (coefficient j takes S[j + 1], the last one -S[0]) and B is added -- the three size-so products never exist in memory
// (per output polynomial 3 x k written + 3 x k read again + one launch less).  Da / Db / Dc: [count][size_ab | size_c][2k + 1 as
// D_q then D_b per batch][n] as fhe_behz_tensor_shared leaves them; out through `mo`, B through `mB` (index maps of circuits.hip).
template <int K>
__global__ __launch_bounds__(256) void k_behz_floor3_combine_pm(const u64 *__restrict__ Aq, const u64 *__restrict__ Ab, const u64 *__restrict__ Bq2,
                                                                const u64 *__restrict__ Bb2, const u64 *__restrict__ Cq, const u64 *__restrict__ Cb,
                                                                u32 size_ab, u32 size_c, const u64 *__restrict__ Bct, CMap mB, u32 size_b,
                                                                u64 *__restrict__ out, CMap mo, const BehzPmDev *__restrict__ Tp, u32 n, u64 n_polys) {
    const BehzPmDev &T = *Tp;
    const u32 stride = gridDim.x * blockDim.x;
    const u32 c0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 p = blockIdx.y; p < n_polys; p += gridDim.y) {
        const u64 ct = p / size_ab;
        const u32 poly = (u32)(p % size_ab);
        u64 S[PCPT][K], r[PCPT][K];
        floor_back_pm_at<K>(T, Aq + p * K * n, Ab + p * (K + 1) * n, n, c0, stride, S);
        floor_back_pm_at<K>(T, Bq2 + p * K * n, Bb2 + p * (K + 1) * n, n, c0, stride, r);
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
            for (int e = 0; e < PCPT; e++) S[e][i] = addmod(S[e][i], r[e][i], T.q[i].q);
        if (poly < size_c) {
            const u64 pc = ct * size_c + poly;
            floor_back_pm_at<K>(T, Cq + pc * K * n, Cb + pc * (K + 1) * n, n, c0, stride, r);
#pragma unroll
            for (int i = 0; i < K; i++)
#pragma unroll
                for (int e = 0; e < PCPT; e++) S[e][i] = addmod(S[e][i], r[e][i], T.q[i].q);
        }
        const u64 *pB = poly < size_b ? Bct + (mB(ct) * size_b + poly) * K * n : nullptr;
        u64 *po = out + (mo(ct) * size_ab + poly) * K * n;
#pragma unroll
        for (int e = 0; e < PCPT; e++) {
            const u32 c = c0 + e * stride, j = c ? c - 1 : n - 1;         // x^-1: coefficient c lands at c - 1, coefficient 0 at n - 1 negated
#pragma unroll
            for (int i = 0; i < K; i++) {
                const u64 q = T.q[i].q;
                u64 v = S[e][i];
                if (!c) v = v ? q - v : 0;
                po[(size_t)i * n + j] = pB ? addmod(v, pB[(size_t)i * n + j], q) : v;
            }
        }
    }
}

// relinearisation ------------------------------------------------------------------------------
// digits: for ciphertext c, source prime i, digit d, target prime ii: ((c2_i >> (dbc d)) & mask) mod q_ii
__global__ __launch_bounds__(256) void k_relin_digits(const u64 *__restrict__ ct, u64 stride, u64 *__restrict__ dig, const BehzDev *__restrict__ Tp,
                                                      u32 n, u32 nd, u32 dbc, u64 count, u32 src_poly) {
    const BehzDev &T = *Tp;
    const u32 k = T.k;
    const u64 mask = dbc >= 64 ? ~0ULL : ((1ULL << dbc) - 1);
    const u64 units = count * k * nd;
    for (u64 u = blockIdx.y; u < units; u += gridDim.y) {
        const u32 d = (u32)(u % nd);
        const u32 i = (u32)((u / nd) % k);
        const u64 c = u / ((u64)nd * k);
        const u64 *c2 = ct + c * stride + ((u64)src_poly * k + i) * n;
        for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
            const u64 v = (c2[s] >> (dbc * d)) & mask;
            for (u32 ii = 0; ii < k; ii++) dig[(u * k + ii) * n + s] = reduce64(v, T.q[ii].q, T.r64q[ii]);
        }
    }
}
// acc[c][pp][ii][s] = sum_{i,d} dig[c][i][d][ii][s] * evk[i][d][pp][ii][s]
__global__ __launch_bounds__(256) void k_relin_accum(const u64 *__restrict__ dig, const u64 *__restrict__ evk, u64 *__restrict__ acc,
                                                     const BehzDev *__restrict__ Tp, u32 n, u32 nd, u64 count) {
    const BehzDev &T = *Tp;
    const u32 k = T.k;
    for (u64 u = blockIdx.y; u < count * k; u += gridDim.y) {
        const u32 ii = (u32)(u % k);
        const u64 c = u / k;
        const Modulus m = T.q[ii];
        for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
            u64 a0 = 0, a1 = 0;
            for (u32 i = 0; i < k; i++)
                for (u32 d = 0; d < nd; d++) {
                    const u64 x = dig[(((c * k + i) * nd + d) * k + ii) * n + s];
                    const u64 *e = evk + ((((u64)i * nd + d) * 2) * k + ii) * n + s;
                    a0 = addmod(a0, mul_barrett(x, e[0], m), m.q);
                    a1 = addmod(a1, mul_barrett(x, e[(u64)k * n], m), m.q);
                }
            acc[((c * 2 + 0) * k + ii) * n + s] = a0;
            acc[((c * 2 + 1) * k + ii) * n + s] = a1;
        }
    }
}
// dst may be the ciphertext itself (in place) or a compact size-2 batch (fhe_relinearize_to)
__global__ __launch_bounds__(256) void k_relin_add(const u64 *ct, u64 stride, u64 *out, u64 out_stride, const u64 *__restrict__ acc, const BehzDev *__restrict__ Tp, u32 n, u64 count) {
    const BehzDev &T = *Tp;
    const u32 k = T.k;
    for (u64 u = blockIdx.y; u < count * 2 * k; u += gridDim.y) {
        const u32 ii = (u32)(u % k);
        const u32 pp = (u32)((u / k) & 1);
        const u64 c = u / (2 * k);
        const u64 *src = ct + c * stride + ((u64)pp * k + ii) * n;
        u64 *dst = out + c * out_stride + ((u64)pp * k + ii) * n;
        for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x)
            dst[s] = addmod(src[s], acc[u * n + s], T.q[ii].q);
    }
}

// Relinearisation on pseudo-Mersenne bases (C = class of the q-base): three launches instead of five.
// (1) digit extraction fused into the forward transforms: workgroup (c, i, d, ii) reads c2_i, takes digit d and transforms it
// modulo q_ii (the k workgroups of one digit sit next to each other: the repeated reads of c2_i hit L2); a digit of
// dbc >= bits(q_ii) bits is folded first.  dig [count][k][nd][k][n], NTT form, canonical.
template <int L, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_relin_fwd_pm(const u64 *__restrict__ ct, u64 stride, u64 *__restrict__ dig, RnsBase base, u32 nd, u32 dbc, u32 src_poly, u32 npow) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    // npow > 1: the polynomials src_poly .. src_poly + npow - 1 at once (one evaluator.relinearize of a size src_poly + npow ciphertext:
    // every step's source polynomial is untouched by the steps before it, fhe_relinearize_n); row i' = power * k + prime
    const u32 k = base.count, ii = blockIdx.x % k, kk = k * npow;
    const u64 u = blockIdx.x / k;                  // (c * kk + i') * nd + d
    const u32 d = (u32)(u % nd), ip = (u32)((u / nd) % kk), i = ip % k;
    src_poly += ip / k;
    const u64 c = u / ((u64)nd * kk);
    const u64 mask = (1ULL << dbc) - 1;            // dbc <= 60
    const PmMod m = base.pm[ii];
    u64 x[1][16];
    load_coeff<L>(x[0], ct + c * stride + ((u64)src_poly * k + i) * N, tid);
    const bool wide = dbc >= m.sh + 32;            // the digit may reach q_ii
#pragma unroll
    for (int r = 0; r < 16; r++) {
        x[0][r] = (x[0][r] >> (dbc * d)) & mask;
        if (wide) x[0][r] = fold_pm(x[0][r], m);
    }
    ntt_fwd_regs_pm<L, 1, PM_FOLDED, C::LIM, C::CS>(x, base.tw_pm + (size_t)ii * N, m, lds, tid);
#pragma unroll
    for (int r = 0; r < 16; r++) x[0][r] = canon_pm(x[0][r], m);
    store_slots<L>(x[0], dig + (u * k + ii) * N, tid);
}
// (2) acc[c][pp][ii][s] = sum_{i,d} dig * evk with mulvv_pm products summed as integers (k nd <= 20 terms of at most 6q), one fold
template <typename C>
__global__ __launch_bounds__(256) void k_relin_accum_pm(const u64 *__restrict__ dig, const u64 *__restrict__ evk, u64 *__restrict__ acc,
                                                        RnsBase base, u32 n, u32 nd, u64 count, u32 npow) {
    const u32 k = base.count, kk = k * npow;      // rows (power, source prime): the key sets of consecutive powers follow each other
    for (u64 u = blockIdx.y; u < count * k; u += gridDim.y) {
        const u32 ii = (u32)(u % k);
        const u64 c = u / k;
        const PmMod m = base.pm[ii];
        for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
            u64 a0 = 0, a1 = 0;
            for (u32 i = 0; i < kk; i++)
                for (u32 d = 0; d < nd; d++) {
                    const u64 x = dig[(((c * kk + i) * nd + d) * k + ii) * n + s];
                    const u64 *e = evk + ((((u64)i * nd + d) * 2) * k + ii) * n + s;
                    a0 += mulvv_pm(x, e[0], m);
                    a1 += mulvv_pm(x, e[(u64)k * n], m);
                }
            acc[((c * 2 + 0) * k + ii) * n + s] = fold_pm(a0, m);
            acc[((c * 2 + 1) * k + ii) * n + s] = fold_pm(a1, m);
        }
    }
}
// (3) inverse transform of acc and the addition into c0 / c1 in one kernel; the sum goes back in place or to a compact size-2 batch
template <int L, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_relin_inv_add_pm(const u64 *ct, u64 stride, u64 *out, u64 out_stride, const u64 *__restrict__ acc, RnsBase base) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u32 k = base.count, ii = blockIdx.x % k, pp = (blockIdx.x / k) & 1;
    const u64 c = blockIdx.x / (2 * k);
    const PmMod m = base.pm[ii];
    u64 x[1][16], y[16];
    u64 *dst = out + c * out_stride + ((u64)pp * k + ii) * N;
    load_slots<L>(x[0], acc + (u64)blockIdx.x * N, tid);
    load_coeff<L>(y, ct + c * stride + ((u64)pp * k + ii) * N, tid);
    ntt_inv_regs_pm<L, 1, PM_FOLDED, C::XB, C::LIM, C::RQ>(x, base.itw_pm + (size_t)ii * N, m, lds, tid);
#pragma unroll
    for (int r = 0; r < 16; r++) y[r] = addmod(y[r], canon_rq_pm<C::RQ>(x[0][r], m), m.q);
    store_coeff<L>(y, dst, tid);
}

// (2) + (3) in one launch (round 5, opt-in FHE_RELIN_FUSED=1: measured 1-3 % slower than the two launches at dbc = 30, level at dbc = 60): workgroup (c, pp, ii) forms acc[c][pp][ii] = sum_{i,d} dig * evk slot by slot in registers -- all
// 32 operand loads of a term in flight before its first product, like the tensor step --, folds once, runs the inverse transform
// and adds c_pp on the way out.  The accumulators never exist in memory (2 polynomials written and read back per ciphertext and
// prime before) and one launch per relinearisation goes away; the digits are read once per pp instead of once (8 more polynomial
// reads per (c, ii), out of L2 for the second reader).  Same integer sums, same fold, same transform: the same bits.
template <int L, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_relin_accum_inv_add_pm(const u64 *ct, u64 stride, u64 *out, u64 out_stride, const u64 *__restrict__ dig,
                                                                                const u64 *__restrict__ evk, RnsBase base, u32 nd) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid = threadIdx.x;
    const u32 k = base.count, ii = blockIdx.x % k, pp = (blockIdx.x / k) & 1;
    const u64 c = blockIdx.x / (2 * k);
    const PmMod m = base.pm[ii];
    u64 acc[1][16], y[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][r] = 0;
    const u32 terms = k * nd;                            // (i, d) pairs, at most 20: 20 x 6q < 2^62 on a 55-bit base (the host passes nd x powers as nd)
    for (u32 t = 0; t < terms; t++) {
        const u64 *pa = dig + ((c * terms + t) * k + ii) * N + tid, *pb = evk + (((u64)t * 2 + pp) * k + ii) * N + tid;
        u64 xa[16], xb[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { xa[r] = pa[r * TP]; xb[r] = pb[r * TP]; }
        PM_FENCE();
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][r] += mulvv_pm(xa[r], xb[r], m);
    }
    u64 *dst = out + c * out_stride + ((u64)pp * k + ii) * N;
    load_coeff<L>(y, ct + c * stride + ((u64)pp * k + ii) * N, tid);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][r] = fold_pm(acc[0][r], m);
    ntt_inv_regs_pm<L, 1, PM_FOLDED, C::XB, C::LIM, C::RQ>(acc, base.itw_pm + (size_t)ii * N, m, lds, tid);
#pragma unroll
    for (int r = 0; r < 16; r++) y[r] = addmod(y[r], canon_rq_pm<C::RQ>(acc[0][r], m), m.q);
    store_coeff<L>(y, dst, tid);
}

inline dim3 grid2(u32 n, u64 rows) { return dim3((n + 255) / 256, (unsigned)(rows < 32768 ? (rows ? rows : 1) : 32768)); }

}  // namespace

// Tables of k_behz_floor_back_pm, when every modulus of the step is pseudo-Mersenne and -- checked here with the actual
// constants and the largest values the variables can take -- no column of a two-column sum can pass 2^64 and every
// quotient part stays below 2^32.  Otherwise pm_dev stays null and the 128-bit kernel runs.
static void behz_pm_tables(const fhe_ctx *c, BehzTables *T, const BehzDev &D, const std::vector<u64> &bsk) {
    using namespace hostmath;
    const u32 k = c->k;
    if (!c->qb.pm_class || !T->aux.pm_class || !T->wide_dot || k > 8 || c->opt.ntt_nopm) return;
    BehzPmDev P;
    std::memset(&P, 0, sizeof P);
    const std::vector<u64> &q = c->qb.primes;
    auto pmmod = [](u64 m) {
        PmMod o;
        const int b = bit_length(m);
        o.q = m; o.delta = (u32)((1ULL << b) - m); o.sh = (u32)(b - 32); o.mb = (1u << (b - 32)) - 1; o.pad = 0;
        return o;
    };
    auto with = [](u64 v, int s, u64 m) { return make_ulonglong2(v, (u64)(((u128)v << s) % m)); };
    for (u32 i = 0; i < k; ++i) {
        P.q[i] = pmmod(q[i]);
        P.t_inv_punct[i] = with(D.t_inv_punct[i].x, 31, q[i]);
        P.mt_inv_punct[i] = with(D.mt_inv_punct[i].x, 31, q[i]);
        P.punct_q_mod_mt[i] = D.punct_q_mod_mt[i];
        P.back_pos[i] = with(D.back_pos[i], PM_SPLIT_Z, q[i]);
        P.back_neg[i] = with(D.back_neg[i], PM_SPLIT_Z, q[i]);
        for (u32 j = 0; j < k; ++j) P.back_B2q[j][i] = with(D.back_B2q[j][i], PM_SPLIT_Z, q[i]);
    }
    for (u32 j = 0; j <= k; ++j) {
        P.b[j] = pmmod(bsk[j]);
        P.flo_t_b[j] = with(D.flo_t_b[j], PM_SPLIT_Z, bsk[j]);
        P.ext_q_b[j] = with(D.ext_q_b[j], PM_SPLIT_Z, bsk[j]);
        for (u32 i = 0; i < k; ++i) {
            P.flo_q2b[i][j] = with(D.flo_q2b[i][j], PM_SPLIT_Y, bsk[j]);
            P.ext_q2b[i][j] = with(D.ext_q2b[i][j], PM_SPLIT_Y, bsk[j]);
        }
    }
    for (u32 j = 0; j < k; ++j) {
        P.inv_punct_B[j] = with(D.inv_punct_B[j].x, 31, bsk[j]);
        P.back_B2msk[j] = with(D.back_B2msk[j], PM_SPLIT_Z, bsk[k]);
    }
    P.inv_B_mod_msk = with(D.inv_B_mod_msk.x, 31, bsk[k]);
    P.neg_inv_q_mod_mt = D.neg_inv_q_mod_mt;
    // worst case of a group of terms (largest variable, its split, its constant pair) under modulus m
    struct Term { u64 vmax; int split; ulonglong2 c; };
    auto fits = [](const std::vector<Term> &g, const PmMod &m) {
        u128 A = 0, B = 0;
        for (const Term &t : g) {
            const u64 xl = (t.vmax >> t.split) ? ((1ULL << t.split) - 1) : t.vmax, xh = t.vmax >> t.split;
            if (xh >> 32) return false;
            A += (u128)xl * (u32)t.c.x + (u128)xh * (u32)t.c.y;
            B += (u128)xl * (u32)(t.c.x >> 32) + (u128)xh * (u32)(t.c.y >> 32);
        }
        if (A >> 64) return false;
        B += A >> 32;
        return !(B >> 64) && !((B >> m.sh) >> 32);
    };
    auto folded = [](const PmMod &m) { const int b = (int)m.sh + 32; return (1ULL << b) + ((u64)m.delta << (64 - b)); };   // fold_pm's bound
    bool ok = true;
    for (u32 i = 0; i < k; ++i) ok = ok && bit_length(q[i]) <= 55;
    for (u32 j = 0; j <= k && ok; ++j) {
        for (u32 i0 = 0; i0 < k && ok; i0 += PM_GROUP_Y) {      // groups of PM_GROUP_Y y-terms, the first with the extra term
            std::vector<Term> g, x;
            if (i0 == 0) {
                g.push_back({folded(P.b[j]), PM_SPLIT_Z, P.flo_t_b[j]});
                x.push_back({bsk[j] - 1, PM_SPLIT_Z, P.ext_q_b[j]});      // base extension: centred remainder + the y_i
            }
            for (u32 i = i0; i < i0 + PM_GROUP_Y && i < k; ++i) {
                g.push_back({q[i] - 1, PM_SPLIT_Y, P.flo_q2b[i][j]});
                x.push_back({q[i] - 1, PM_SPLIT_Y, P.ext_q2b[i][j]});
            }
            ok = fits(g, P.b[j]) && fits(x, P.b[j]);
        }
    }
    for (u32 j0 = 0; j0 < k && ok; j0 += 2) {
        std::vector<Term> g;
        for (u32 j = j0; j < j0 + 2 && j < k; ++j) g.push_back({bsk[j] - 1, PM_SPLIT_Z, P.back_B2msk[j]});
        ok = fits(g, P.b[k]);
        for (u32 i = 0; i < k && ok; ++i) {
            std::vector<Term> h;
            for (u32 j = j0; j < j0 + 2 && j < k; ++j) h.push_back({bsk[j] - 1, PM_SPLIT_Z, P.back_B2q[j][i]});
            ok = fits(h, P.q[i]);
        }
    }
    for (u32 i = 0; i < k && ok; ++i)
        ok = fits({{bsk[k] >> 1, PM_SPLIT_Z, P.back_pos[i]}}, P.q[i]) && fits({{bsk[k] >> 1, PM_SPLIT_Z, P.back_neg[i]}}, P.q[i]);
    if (!ok) return;
    if (hipMalloc((void **)&T->pm_dev, sizeof(BehzPmDev)) != hipSuccess) { T->pm_dev = nullptr; (void)hipGetLastError(); return; }
    if (hipMemcpy(T->pm_dev, &P, sizeof(BehzPmDev), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(T->pm_dev); T->pm_dev = nullptr; }
}

int fhe_behz_build(fhe_ctx *c) {
    using namespace hostmath;
    if (c->behz) return FHE_OK;
    const u32 k = c->k, n = c->n;
    BehzTables *T = new BehzTables();
    BehzDev &D = T->host;
    memset(&D, 0, sizeof(D));
    D.k = k;
    const std::vector<u64> &q = c->qb.primes;
    // auxiliary primes: = 1 (mod 2^17), descending from 2^58 (or 2^61); first one is m_sk, the next k form B.
    // The result of a product does not depend on them (header comment) as long as the value the fast floor leaves in
    // Bsk -- at most min(sa, sb) n t q / 4 in magnitude, sizes up to 2^8 here -- stays below B m_sk / 2.  58-bit primes
    // let the forward transforms run without conditional subtractions (ntt_core.h LAZY); 61-bit ones (what SEAL 2.3
    // takes) remain the fallback when k + 1 of the smaller ones are not enough.
    int q_bits = 0;
    for (u64 qi : q) q_bits += bit_length(qi);
    const int need = q_bits + bit_length(c->t) + (int)c->logn + 8 + 4;
    const int aux_bits = (57 * (int)(k + 1) >= need && !c->opt.behz_aux61) ? 58 : 61;
    T->aux_bits = aux_bits;
    T->wide_dot = aux_bits == 58 && c->max_prime_bits <= 58 && k <= 8 && !c->opt.behz_chunk3;
    std::vector<u64> found;
    for (u64 cand = (1ULL << aux_bits) + 1 - (1ULL << 17); found.size() < k + 1; cand -= (1ULL << 17)) {
        if (!is_prime(cand)) continue;
        bool clash = false;
        for (u64 qi : q) clash |= (qi == cand);
        if (!clash) found.push_back(cand);
    }
    std::vector<u64> bsk(k + 1);
    for (u32 j = 0; j < k; ++j) bsk[j] = found[j + 1];
    bsk[k] = found[0];
    int rc = fhe_build_base(T->aux, bsk, n, c->logn, false);
    if (rc) { delete T; return rc; }
    auto pair = [](u64 v, u64 m) { return make_ulonglong2(v, shoup(v, m)); };
    const u64 mt = 1ULL << 32;
    for (u32 i = 0; i < k; ++i) {
        D.q[i] = c->qb.h_mod[i];
        D.r64q[i] = (u64)((((u128)1) << 64) / q[i]);
        const u64 ip = invmod(prod_mod(q.data(), (int)k, (int)i, q[i]), q[i]);
        D.inv_punct[i] = pair(ip, q[i]);
        D.mt_inv_punct[i] = pair(mulmod(mt % q[i], ip, q[i]), q[i]);
        for (u32 j = 0; j <= k; ++j) D.punct_q_mod_b[i][j] = pair(prod_mod(q.data(), (int)k, (int)i, bsk[j]), bsk[j]);
        D.punct_q_mod_mt[i] = prod_mod(q.data(), (int)k, (int)i, mt);
        D.t_mod_q[i] = pair(c->t % q[i], q[i]);
        D.B_mod_q[i] = pair(prod_mod(bsk.data(), (int)k, -1, q[i]), q[i]);
    }
    {
        const u64 qm = prod_mod(q.data(), (int)k, -1, mt);   // odd
        u64 x = qm;
        for (int it = 0; it < 6; ++it) x = (x * (2 - qm * x)) & (mt - 1);
        D.neg_inv_q_mod_mt = (mt - x) & (mt - 1);
    }
    for (u32 j = 0; j <= k; ++j) {
        D.b[j] = T->aux.h_mod[j];
        const u64 qmb = prod_mod(q.data(), (int)k, -1, bsk[j]);
        D.q_mod_b[j] = pair(qmb, bsk[j]);
        D.inv_q_mod_b[j] = pair(invmod(qmb, bsk[j]), bsk[j]);
        D.inv_mt_mod_b[j] = pair(invmod(mt % bsk[j], bsk[j]), bsk[j]);
        D.t_mod_b[j] = pair(c->t % bsk[j], bsk[j]);
    }
    for (u32 j = 0; j < k; ++j) {
        D.inv_punct_B[j] = pair(invmod(prod_mod(bsk.data(), (int)k, (int)j, bsk[j]), bsk[j]), bsk[j]);
        for (u32 i = 0; i < k; ++i) D.punct_B_mod_q[j][i] = pair(prod_mod(bsk.data(), (int)k, (int)j, q[i]), q[i]);
        D.punct_B_mod_msk[j] = pair(prod_mod(bsk.data(), (int)k, (int)j, bsk[k]), bsk[k]);
    }
    D.inv_B_mod_msk = pair(invmod(prod_mod(bsk.data(), (int)k, -1, bsk[k]), bsk[k]), bsk[k]);
    auto mu2_of = [](u64 m, u64 &mu2, u32 &sh) {
        const int bits = bit_length(m);
        sh = (u32)(bits - 1);
        mu2 = (u64)((((u128)1) << (bits + 63)) / m);
    };
    for (u32 i = 0; i < k; ++i) {
        mu2_of(q[i], D.mu2_q[i], D.sh_q[i]);
        D.t_inv_punct[i] = pair(mulmod(c->t % q[i], D.inv_punct[i].x, q[i]), q[i]);
        D.back_neg[i] = D.B_mod_q[i].x;
        D.back_pos[i] = (q[i] - D.B_mod_q[i].x) % q[i];
        for (u32 j = 0; j < k; ++j) D.back_B2q[j][i] = D.punct_B_mod_q[j][i].x;
    }
    for (u32 j = 0; j <= k; ++j) {
        mu2_of(bsk[j], D.mu2_b[j], D.sh_b[j]);
        D.ext_q_b[j] = mulmod(D.q_mod_b[j].x, D.inv_mt_mod_b[j].x, bsk[j]);
        D.flo_t_b[j] = mulmod(c->t % bsk[j], D.inv_q_mod_b[j].x, bsk[j]);
        for (u32 i = 0; i < k; ++i) {
            D.ext_q2b[i][j] = mulmod(D.punct_q_mod_b[i][j].x, D.inv_mt_mod_b[j].x, bsk[j]);
            D.flo_q2b[i][j] = (bsk[j] - mulmod(D.punct_q_mod_b[i][j].x, D.inv_q_mod_b[j].x, bsk[j])) % bsk[j];
        }
    }
    for (u32 j = 0; j < k; ++j) D.back_B2msk[j] = D.punct_B_mod_msk[j].x;
    behz_pm_tables(c, T, D, bsk);
    if (hipMalloc((void **)&T->dev, sizeof(BehzDev)) != hipSuccess ||
        hipMemcpy(T->dev, &D, sizeof(BehzDev), hipMemcpyHostToDevice) != hipSuccess) {
        fhe_free_base(T->aux);
        delete T;
        return fail(FHE_ERR_HIP, "BEHZ table upload failed");
    }
    c->behz = T;
    return FHE_OK;
}

int fhe_behz_ensure(const fhe_ctx *cc) {
    fhe_ctx *c = const_cast<fhe_ctx *>(cc);
    std::call_once(c->behz_once, [c] {
        // the calling thread may be bound to another device (a host driving several GPUs): the tables belong to the context's
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != c->device) (void)hipSetDevice(c->device);
        c->behz_rc = fhe_behz_build(c);
        if (c->behz_rc) c->behz_err = fhe_last_error();
        if (cur >= 0 && cur != c->device) (void)hipSetDevice(cur);
    });
    if (c->behz_rc) return fail(c->behz_rc, "ct x ct tables: %s", c->behz_err.c_str());
    return FHE_OK;
}

extern "C" int fhe_ctx_has_ctct_tables(const fhe_ctx *c) { return c && c->behz ? 1 : 0; }

extern "C" int fhe_arith_path(const fhe_ctx *c) {
    if (!c) return fail(FHE_ERR_PARAM, "null argument");
    if (c->opt.ntt_nopm) return 0;
    int r = c->qb.pm_class & 3;
    if (fhe_behz_ensure(c) == FHE_OK) r |= ((c->behz->aux.pm_class & 3) << 2) | (c->behz->pm_dev ? 16 : 0);
    return r;
}

void fhe_behz_free(fhe_ctx *c) {
    if (c && c->behz) {
        fhe_free_base(c->behz->aux);
        if (c->behz->pm_dev) (void)hipFree(c->behz->pm_dev);
        if (c->behz->dev) (void)hipFree(c->behz->dev);
        delete c->behz;
        c->behz = nullptr;
    }
}

static size_t mul_words(const fhe_ctx *c, u32 sa, u32 sb, u64 count, bool square) {
    const size_t kn = (size_t)c->k * c->n, bn = (size_t)(c->k + 1) * c->n;
    const size_t so = sa + sb - 1;
    size_t w = count * sa * (kn + bn) + count * so * (kn + bn);
    if (!square) w += count * sb * (kn + bn);
    return w;
}
extern "C" size_t fhe_multiply_scratch_bytes(const fhe_ctx *c, uint32_t sa, uint32_t sb, uint64_t count) {
    if (!c || !sa || !sb) return 0;
    return mul_words(c, sa, sb, count, false) * sizeof(u64);
}

// q-base transforms of `n_rns` RNS polynomials: the FP64 kernels where the context supports them (same
// NTT-form order and canonical residues as the u64 kernels), the u64 kernels otherwise
static int qbase_ntt(bool inverse, const fhe_ctx *c, const u64 *in, u64 *out, u64 n_rns, hipStream_t st) {
    if (n_rns && fhe_rgb_f64_supported(c) && !c->opt.force_u64) return fhe_poly_f64_launch(inverse ? 1 : 0, c, in, out, n_rns, nullptr, st);
    return fhe_ntt_launch(inverse, c, c->qb, in, out, n_rns * c->k, st);
}

// operand preparation (steps 0-2 up to the forward transforms): src [count][s][k][n] -> xq [count][s][k][n] (NTT),
// xb [count][s][k+1][n] (NTT)
template <int K>
static int behz_prepare_fused(const fhe_ctx *c, const u64 *src, u64 n_polys, u64 *xq, u64 *xb, hipStream_t st) {
    const RnsBase qb = c->qb.dev(), bb = c->behz->aux.dev();
    if ((n_polys + 8) * (2 * K + 1) > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    const unsigned grid = (unsigned)(((n_polys + 7) / 8) * 8 * (2 * K + 1));
    DISPATCH_L(c->logn, {
        if (c->qb.pm_class == 1) k_behz_prepare_pm<L, K, PmA, PmB><<<grid, NttShape<L>::TP, 0, st>>>(src, xq, xb, qb, bb, c->behz->pm_dev, n_polys);
        else k_behz_prepare_pm<L, K, PmB, PmB><<<grid, NttShape<L>::TP, 0, st>>>(src, xq, xb, qb, bb, c->behz->pm_dev, n_polys);
    });
    KERNEL_CHECK();
    return FHE_OK;
}
static int behz_prepare(const fhe_ctx *c, const u64 *src, u32 s, u64 count, u64 *xq, u64 *xb, hipStream_t st) {
    const u32 k = c->k, n = c->n;
    // both bases on pseudo-Mersenne arithmetic with two-column conversions: base extension fused into the forward transforms
    if (c->behz->pm_dev && c->qb.pm_class && c->behz->aux.pm_class == 2 && !c->opt.ntt_nopm && c->opt.behz_fused_prepare && k <= 4 &&
        !(fhe_rgb_f64_supported(c) && !c->opt.force_u64)) {
        switch (k) {
            case 1: return behz_prepare_fused<1>(c, src, count * s, xq, xb, st);
            case 2: return behz_prepare_fused<2>(c, src, count * s, xq, xb, st);
            case 3: return behz_prepare_fused<3>(c, src, count * s, xq, xb, st);
            default: return behz_prepare_fused<4>(c, src, count * s, xq, xb, st);
        }
    }
    switch (k) {
#define GO(KK) case KK: if (c->behz->pm_dev) k_behz_to_bsk_pm<KK><<<grid2(n / PCPT, count * s), 256, 0, st>>>(src, xb, c->behz->pm_dev, n, count * s); \
                        else if (c->behz->wide_dot) k_behz_to_bsk<KK, TO_BSK_CPT, WIDE_CHUNK><<<grid2(n / TO_BSK_CPT, count * s), 256, 0, st>>>(src, xb, c->behz->dev, n, count * s); \
                        else k_behz_to_bsk<KK, TO_BSK_CPT, DOT_CHUNK><<<grid2(n / TO_BSK_CPT, count * s), 256, 0, st>>>(src, xb, c->behz->dev, n, count * s); break;
        GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
#undef GO
        default: return fail(FHE_ERR_PARAM, "base extension is built for up to 8 coefficient moduli, not %u", k);
    }
    int r = fhe_ntt_launch(false, c, c->behz->aux, xb, xb, count * s * (k + 1), st);
    if (r) return r;
    return qbase_ntt(false, c, src, xq, count * s, st);
}
// tensor product fused into the inverse transforms over one base: pairs of ciphertext pairs per workgroup at n >= 8192
// (P8192 inverse transform +19 % with shared twiddles), the odd one out and the smaller degrees one per workgroup
static int tensor_intt(const fhe_ctx *c, const u64 *A, const u64 *Bm, u64 *D, const RnsBase base, u32 sa, u32 sb, u64 count, hipStream_t st, BMap bm,
                       bool wide_base, int pm_class) {
    const u32 nb = base.count, so = sa + sb - 1;
    const bool wide = wide_base && (sa < sb ? sa : sb) <= 12 && !c->opt.behz_tensor_canon;     // terms per output <= min(sa, sb); 12 x 5q < 2^64
    const bool single = c->opt.ntt_single || c->opt.behz_tensor_single;      // behz_tensor_single: only this step on the one-polynomial kernel (which has the range-tracking inverse)
    u64 done = 0;
    const int pmc = (wide && c->behz->wide_dot && !c->opt.ntt_nopm && base.pm) ? pm_class : 0;     // pseudo-Mersenne inverse passes
    if (pmc) {
        // one ciphertext pair per workgroup, below (see fhe_ntt_launch: the two-polynomial shape buys nothing on these passes)
    } else if (c->logn >= 13 && !single && count >= 2) {
        const u64 pairs = count / 2;
        switch (c->logn) {
#define GO2(LL, WW) k_behz_tensor_intt2<LL, WW><<<(unsigned)(((pairs * nb + 7) / 8) * 8 * so), NttShape<LL>::TP, 0, st>>>(A, Bm, D, base, sa, sb, pairs * nb, bm)
            case 13: if (wide) GO2(13, true); else GO2(13, false); break;
            default: if (wide) GO2(14, true); else GO2(14, false); break;
#undef GO2
        }
        done = 2 * pairs;
    }
    if (done < count) {
        const u64 rest = count - done;
        const size_t n = c->n;
        // a shared B batch is indexed by the absolute pair number: its pointer stays, the map carries the offset
        const u64 *A2 = A + done * sa * nb * n, *B2 = bm.cnt ? Bm : Bm + done * sb * nb * n;
        u64 *D2 = D + done * so * nb * n;
        if (bm.cnt) bm.off += done;
        if (pmc) {
            const unsigned grid = (unsigned)(((rest * nb + 7) / 8) * 8 * so);
            DISPATCH_L(c->logn, {
                const bool square = A2 == B2 && sa == sb && !bm.cnt && !c->opt.behz_square_full;
                if (pmc == 1 && square) k_behz_tensor_intt_pm<L, 1, PmA, true><<<grid, NttShape<L>::TP, 0, st>>>(A2, B2, D2, base, sa, sb, rest * nb, bm);
                else if (pmc == 1) k_behz_tensor_intt_pm<L, 1, PmA, false><<<grid, NttShape<L>::TP, 0, st>>>(A2, B2, D2, base, sa, sb, rest * nb, bm);
                else if (square) k_behz_tensor_intt_pm<L, 1, PmB, true><<<grid, NttShape<L>::TP, 0, st>>>(A2, B2, D2, base, sa, sb, rest * nb, bm);
                else k_behz_tensor_intt_pm<L, 1, PmB, false><<<grid, NttShape<L>::TP, 0, st>>>(A2, B2, D2, base, sa, sb, rest * nb, bm);
            });
        } else if (wide) { DISPATCH_L(c->logn, (k_behz_tensor_intt<L, true><<<(unsigned)(((rest * nb + 7) / 8) * 8 * so), NttShape<L>::TP, 0, st>>>(A2, B2, D2, base, sa, sb, rest * nb, bm))); }
        else { DISPATCH_L(c->logn, (k_behz_tensor_intt<L, false><<<(unsigned)(((rest * nb + 7) / 8) * 8 * so), NttShape<L>::TP, 0, st>>>(A2, B2, D2, base, sa, sb, rest * nb, bm))); }
    }
    KERNEL_CHECK();
    return FHE_OK;
}

// tensor product + inverse transforms + floor/back-conversion from prepared operands; D = scratch for so polynomials
static int behz_finish(const fhe_ctx *c, const u64 *Aq, const u64 *Ab, u32 sa, const u64 *Bq, const u64 *Bb, u32 sb, u64 *out, u64 count,
                       u64 *Dq, u64 *Db, hipStream_t st, BMap bm = BMap{1, 0, 0}) {
    const u32 k = c->k, n = c->n, so = sa + sb - 1;
    int rc;
    if ((count * (u64)(k + 1) + 8) * so > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    const bool q_f64 = fhe_rgb_f64_supported(c);      // FP64 inverse transforms beat the fused u64 kernel there
    if (q_f64) {
        k_behz_tensor<<<grid2(n, count * k), 256, 0, st>>>(Aq, Bq, Dq, c->qb.d_mod, c->behz->dev->mu2_q, k, n, sa, sb, count, bm);
        if ((rc = qbase_ntt(true, c, Dq, Dq, count * so, st))) return rc;
    } else {
        const RnsBase qb = c->qb.dev();
        if ((rc = tensor_intt(c, Aq, Bq, Dq, qb, sa, sb, count, st, bm, c->max_prime_bits <= 58, c->qb.pm_class))) return rc;
    }
    if ((rc = tensor_intt(c, Ab, Bb, Db, c->behz->aux.dev(), sa, sb, count, st, bm, c->behz->aux_bits <= 58, c->behz->aux.pm_class))) return rc;
    if (c->behz->pm_dev) {
        switch (k) {
#define GO(KK) case KK: k_behz_floor_back_pm<KK><<<grid2(n / PCPT, count * so), 256, 0, st>>>(Dq, Db, out, c->behz->pm_dev, n, count * so); break;
            GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
#undef GO
            default: return fail(FHE_ERR_PARAM, "pseudo-Mersenne floor / back conversion is built for up to 8 coefficient moduli, not %u", k);
        }
        KERNEL_CHECK();
        return FHE_OK;
    }
    switch (k) {
#define GO(KK) case KK: if (c->behz->wide_dot) k_behz_floor_back<KK, WIDE_CHUNK><<<grid2(n / CPT, count * so), 256, 0, st>>>(Dq, Db, out, c->behz->dev, n, count * so); \
                        else k_behz_floor_back<KK, DOT_CHUNK><<<grid2(n / CPT, count * so), 256, 0, st>>>(Dq, Db, out, c->behz->dev, n, count * so); break;
        GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
#undef GO
        default: return fail(FHE_ERR_PARAM, "floor / back conversion is built for up to 8 coefficient moduli, not %u", k);
    }
    KERNEL_CHECK();
    return FHE_OK;
}

// a / b: plain ciphertexts (used when the matching prepared pointer is null); ap / bp: prepared operands
static int behz_multiply(const fhe_ctx *cc, const u64 *a, const u64 *ap, u32 sa, const u64 *b, const u64 *bp, u32 sb, u64 *out, u64 count,
                         void *scratch, size_t scratch_bytes, hipStream_t st) {
    if (!cc || (!a && !ap) || (!b && !bp) || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (sa < 1 || sb < 1) return fail(FHE_ERR_PARAM, "ciphertext sizes must be at least 1");
    if (!count) return FHE_OK;
    const fhe_ctx *c = cc;
    int rc;
    if (int erc = fhe_behz_ensure(c)) return erc;
    const bool square = !ap && !bp && a == b && sa == sb;
    if (!scratch || scratch_bytes < mul_words(c, sa, sb, count, square) * sizeof(u64))
        return fail(FHE_ERR_PARAM, "scratch too small: need fhe_multiply_scratch_bytes()");
    const u32 k = c->k, n = c->n, so = sa + sb - 1;
    const size_t kn = (size_t)k * n, bn = (size_t)(k + 1) * n;
    u64 *p = (u64 *)scratch;
    const u64 *Aq, *Ab, *Bq, *Bb;
    if (ap) { Aq = ap; Ab = ap + count * sa * kn; }
    else {
        u64 *xq = p, *xb = xq + count * sa * kn;
        p = xb + count * sa * bn;
        if ((rc = behz_prepare(c, a, sa, count, xq, xb, st))) return rc;
        Aq = xq; Ab = xb;
    }
    if (square) { Bq = Aq; Bb = Ab; }
    else if (bp) { Bq = bp; Bb = bp + count * sb * kn; }
    else {
        u64 *xq = p, *xb = xq + count * sb * kn;
        p = xb + count * sb * bn;
        if ((rc = behz_prepare(c, b, sb, count, xq, xb, st))) return rc;
        Bq = xq; Bb = xb;
    }
    u64 *Dq = p, *Db = Dq + count * so * kn;
    return behz_finish(c, Aq, Ab, sa, Bq, Bb, sb, out, count, Dq, Db, st);
}

extern "C" size_t fhe_multiply_operand_words(const fhe_ctx *c, uint32_t size, uint64_t count) {
    if (!c) return 0;
    return (size_t)count * size * (2 * (size_t)c->k + 1) * c->n;
}
extern "C" int fhe_multiply_prepare(const fhe_ctx *cc, const uint64_t *a, uint32_t size, uint64_t count, uint64_t *prepared, fhe_stream s) {
    if (!cc || !a || !prepared) return fail(FHE_ERR_PARAM, "null argument");
    if (!size) return fail(FHE_ERR_PARAM, "ciphertext sizes must be at least 1");
    if (!count) return FHE_OK;
    const fhe_ctx *c = cc;
    if (int erc = fhe_behz_ensure(c)) return erc;
    u64 *xq = (u64 *)prepared, *xb = xq + count * size * (size_t)c->k * c->n;
    return behz_prepare(c, (const u64 *)a, size, count, xq, xb, (hipStream_t)s);
}
extern "C" int fhe_multiply_prepared(const fhe_ctx *c, const uint64_t *a, const uint64_t *ap, uint32_t sa, const uint64_t *b, const uint64_t *bp,
                                     uint32_t sb, uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream s) {
    return behz_multiply(c, (const u64 *)a, (const u64 *)ap, sa, (const u64 *)b, (const u64 *)bp, sb, (u64 *)out, count, scratch, scratch_bytes,
                         (hipStream_t)s);
}

// a (plain or prepared) x a prepared operand batch of b_count entries shared between the pairs:
// pair c takes entry (b_first + c / b_div) % b_count
extern "C" int fhe_multiply_prepared_shared(const fhe_ctx *cc, const uint64_t *a, const uint64_t *ap, uint32_t sa, const uint64_t *bp, uint32_t sb,
                                            uint64_t b_count, uint64_t b_div, uint64_t b_first, uint64_t *out, uint64_t count, void *scratch,
                                            size_t scratch_bytes, fhe_stream s) {
    if (!cc || (!a && !ap) || !bp || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (sa < 1 || sb < 1) return fail(FHE_ERR_PARAM, "ciphertext sizes must be at least 1");
    if (!b_count || !b_div) return fail(FHE_ERR_PARAM, "shared operand: b_count and b_div must be positive");
    if (!count) return FHE_OK;
    const fhe_ctx *c = cc;
    if (int erc = fhe_behz_ensure(c)) return erc;
    if (!scratch || scratch_bytes < mul_words(c, sa, sb, count, false) * sizeof(u64))
        return fail(FHE_ERR_PARAM, "scratch too small: need fhe_multiply_scratch_bytes()");
    hipStream_t st = (hipStream_t)s;
    const u32 k = c->k, n = c->n, so = sa + sb - 1;
    const size_t kn = (size_t)k * n, bn = (size_t)(k + 1) * n;
    u64 *p = (u64 *)scratch;
    const u64 *Aq, *Ab;
    int rc;
    if (ap) { Aq = (const u64 *)ap; Ab = Aq + count * sa * kn; }
    else {
        u64 *xq = p, *xb = xq + count * sa * kn;
        p = xb + count * sa * bn;
        if ((rc = behz_prepare(c, (const u64 *)a, sa, count, xq, xb, st))) return rc;
        Aq = xq; Ab = xb;
    }
    const u64 *Bq = (const u64 *)bp, *Bb = Bq + b_count * sb * kn;
    u64 *Dq = p, *Db = Dq + count * so * kn;
    return behz_finish(c, Aq, Ab, sa, Bq, Bb, sb, (u64 *)out, count, Dq, Db, st, BMap{b_div, b_count, (b_first % b_count) * b_div});
}

bool fhe_behz_floor3_supported(const fhe_ctx *c) { return c && fhe_behz_ensure(c) == FHE_OK && c->behz->pm_dev && c->k <= 8 && !fhe_rgb_f64_supported(c); }
size_t fhe_behz_d_words(const fhe_ctx *c, u32 so, u64 count) { return (size_t)count * so * (2 * (size_t)c->k + 1) * c->n; }
static int tensor_both(const fhe_ctx *c, const u64 *Aq, const u64 *Ab, u32 sa, const u64 *Bq, const u64 *Bb, u32 sb, u64 count, u64 *Dq, u64 *Db, hipStream_t st, BMap bm) {
    int rc;
    if ((count * (u64)(c->k + 1) + 8) * (sa + sb - 1) > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    if ((rc = tensor_intt(c, Aq, Bq, Dq, c->qb.dev(), sa, sb, count, st, bm, c->max_prime_bits <= 58, c->qb.pm_class))) return rc;
    return tensor_intt(c, Ab, Bb, Db, c->behz->aux.dev(), sa, sb, count, st, bm, c->behz->aux_bits <= 58, c->behz->aux.pm_class);
}
int fhe_behz_tensor_shared(const fhe_ctx *c, const u64 *a, u32 sa, const u64 *bp, u32 sb, u64 b_count, u64 b_div, u64 b_first, u64 *d, u64 count,
                           u64 *scratch, hipStream_t st) {
    if (int erc = fhe_behz_ensure(c)) return erc;
    const u32 k = c->k, n = c->n, so = sa + sb - 1;
    const size_t kn = (size_t)k * n;
    u64 *xq = scratch, *xb = xq + count * sa * kn;
    int rc;
    if ((rc = behz_prepare(c, a, sa, count, xq, xb, st))) return rc;
    const u64 *Bq = bp, *Bb = Bq + b_count * sb * kn;
    return tensor_both(c, xq, xb, sa, Bq, Bb, sb, count, d, d + count * so * kn, st, BMap{b_div, b_count, (b_first % b_count) * b_div});
}
int fhe_behz_floor3_combine(const fhe_ctx *c, const u64 *da, const u64 *db, const u64 *dc, u32 size_ab, u32 size_c, const u64 *B, CMap mB, u32 size_b,
                            u64 *out, CMap mo, u64 count, hipStream_t st) {
    const u32 k = c->k, n = c->n;
    const size_t kn = (size_t)k * n;
    const u64 np = count * size_ab;
    switch (k) {
#define GO(KK) case KK: k_behz_floor3_combine_pm<KK><<<grid2(n / PCPT, np), 256, 0, st>>>(da, da + count * size_ab * kn, db, db + count * size_ab * kn, dc, dc + count * size_c * kn, \
                                                                                   size_ab, size_c, B, mB, size_b, out, mo, c->behz->pm_dev, n, np); break;
        GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
#undef GO
        default: return fail(FHE_ERR_PARAM, "floor / back conversion is built for up to 8 coefficient moduli, not %u", k);
    }
    KERNEL_CHECK();
    return FHE_OK;
}

extern "C" int fhe_multiply(const fhe_ctx *c, const uint64_t *a, uint32_t sa, const uint64_t *b, uint32_t sb, uint64_t *out,
                            uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream s) {
    return behz_multiply(c, (const u64 *)a, nullptr, sa, (const u64 *)b, nullptr, sb, (u64 *)out, count, scratch, scratch_bytes, (hipStream_t)s);
}
// SEAL special-cases size 2 as (c0^2, 2 c0 c1, c1^2); that is the same ring tensor the generic
// product forms, so square shares the multiply path (transforming the operand once).
extern "C" int fhe_square(const fhe_ctx *c, const uint64_t *a, uint32_t sa, uint64_t *out, uint64_t count, void *scratch,
                          size_t scratch_bytes, fhe_stream s) {
    return behz_multiply(c, (const u64 *)a, nullptr, sa, (const u64 *)a, nullptr, sa, (u64 *)out, count, scratch, scratch_bytes, (hipStream_t)s);
}

extern "C" uint32_t fhe_evk_digits(const fhe_ctx *c, uint32_t dbc) {
    if (!c || !dbc) return 0;
    return (uint32_t)((c->max_prime_bits + dbc - 1) / dbc);
}
extern "C" size_t fhe_relinearize_scratch_bytes(const fhe_ctx *c, uint32_t dbc, uint64_t count) {
    if (!c || !dbc) return 0;
    const size_t kn = (size_t)c->k * c->n;
    return (count * c->k * fhe_evk_digits(c, dbc) * kn + count * 2 * kn) * sizeof(u64);
}
extern "C" size_t fhe_evk_words(const fhe_ctx *c, uint32_t dbc) {
    if (!c || !dbc) return 0;
    return (size_t)c->k * fhe_evk_digits(c, dbc) * 2 * c->k * c->n;
}
static bool relin_pm_ok(const fhe_ctx *c, u32 nd, u32 npow, u64 count);
static int relin_pm(const fhe_ctx *c, const u64 *ct, u64 stride, u32 src_poly, u32 npow, u64 *out2, u64 out_stride, u64 count, const u64 *evk, u32 dbc, u64 *scratch,
                    hipStream_t st);
extern "C" size_t fhe_relinearize_n_scratch_bytes(const fhe_ctx *c, uint32_t size, uint32_t dbc, uint64_t count) {
    if (!c || !dbc || size < 3) return 0;
    const size_t kn = (size_t)c->k * c->n;
    return (count * c->k * fhe_evk_digits(c, dbc) * kn * (size - 2) + count * 2 * kn) * sizeof(u64);
}
extern "C" int fhe_relinearize_n(const fhe_ctx *c, uint64_t *ct, uint32_t size, uint64_t stride, uint64_t *out2, uint64_t out_stride, uint64_t count,
                                 const uint64_t *evk, uint32_t dbc, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!c || !ct || !evk || !out2) return fail(FHE_ERR_PARAM, "null argument");
    if (size < 3 || size > FHE_MAX_POLYS) return fail(FHE_ERR_PARAM, "relinearize: %u polynomials (3 .. %d)", size, FHE_MAX_POLYS);
    if (dbc < 1 || dbc > 60) return fail(FHE_ERR_PARAM, "decomposition bit count out of range");
    // the size - 2 key switches in as FEW passes as the lazy sums allow (k x digits x powers <= 20 terms per pass: all of them at once for
    // a Cubic's size-4 result, and for a sampler's size-6 result at dbc 60; two passes of two powers at dbc 30), the top powers first and in
    // place, the last pass into out2 -- when the caller brought the digits' scratch (fhe_relinearize_n_scratch_bytes);
    // FHE_RELIN_STEPS=1 keeps the sequential steps (A/B measurements, the parity test's second path)
    if (size > 3 && count && !c->opt.relin_steps && scratch && scratch_bytes >= fhe_relinearize_n_scratch_bytes(c, size, dbc, count)) {
        if (int erc = fhe_behz_ensure(c)) return erc;
        const u32 nd = fhe_evk_digits(c, dbc);
        u32 gmax = size - 2;
        while (gmax > 1 && !relin_pm_ok(c, nd, gmax, count)) --gmax;
        if (gmax > 1 && relin_pm_ok(c, nd, gmax, count)) {
            if (stride < (u64)size * c->k * c->n) return fail(FHE_ERR_PARAM, "ciphertext stride smaller than a size-%u ciphertext", size);
            if (out_stride < (u64)2 * c->k * c->n) return fail(FHE_ERR_PARAM, "output stride smaller than a size-2 ciphertext");
            if (!(out2 == ct && out_stride == stride)) {
                const uintptr_t i0 = (uintptr_t)ct, i1 = i0 + ((count - 1) * stride + (u64)size * c->k * c->n) * sizeof(u64);
                const uintptr_t o0 = (uintptr_t)out2, o1 = o0 + ((count - 1) * out_stride + (u64)2 * c->k * c->n) * sizeof(u64);
                if (o0 < i1 && i0 < o1) return fail(FHE_ERR_PARAM, "output range overlaps the input range (only out2 == ct with out_stride == stride may alias)");
            }
            const size_t ew = fhe_evk_words(c, dbc);
            u32 hi = size - 1;                                       // polynomials [lo, hi] per pass, from the top
            while (hi >= 2) {
                const u32 g = hi - 1 < gmax ? hi - 1 : gmax, lo = hi - g + 1;
                const bool last = lo == 2;
                if (int rc = relin_pm(c, (const u64 *)ct, stride, lo, g, last ? (u64 *)out2 : (u64 *)ct, last ? out_stride : stride, count,
                                      (const u64 *)evk + (size_t)(lo - 2) * ew, dbc, (u64 *)scratch, (hipStream_t)s)) return rc;
                hi = lo - 1;
            }
            return FHE_OK;
        }
    }
    const size_t ew = fhe_evk_words(c, dbc);
    for (uint32_t p = size - 1; p >= 3; --p)                 // the top polynomial first, with the keys for s^p, in place
        if (int rc = fhe_relinearize_poly(c, ct, stride, p, ct, stride, count, evk + (size_t)(p - 2) * ew, dbc, scratch, scratch_bytes, s)) return rc;
    return fhe_relinearize_poly(c, ct, stride, 2, out2, out_stride, count, evk, dbc, scratch, scratch_bytes, s);
}
extern "C" int fhe_relinearize(const fhe_ctx *cc, uint64_t *ct3, uint64_t stride, uint64_t count, const uint64_t *evk, uint32_t dbc,
                               void *scratch, size_t scratch_bytes, fhe_stream s) {
    return fhe_relinearize_to(cc, ct3, stride, ct3, stride, count, evk, dbc, scratch, scratch_bytes, s);
}
// The pseudo-Mersenne key switch: polynomials src_poly .. src_poly + npow - 1 with the keys for s^src_poly .. in ONE pass of three
// launches.  npow > 1 is evaluator.relinearize of a size src_poly + npow ciphertext as one sum: SEAL's steps run top polynomial first,
// but step j only ever writes c0 / c1, so the source polynomial of every step is the caller's own -- the result is c + sum over
// the steps of their key-switch terms, and modular addition does not care about the order: the same bits as the sequential steps
// (tests/test_gpu_relin.py compares them), with one inverse transform pair and one addition pass instead of npow.
static bool relin_pm_ok(const fhe_ctx *c, u32 nd, u32 npow, u64 count) {
    const bool f64 = fhe_rgb_f64_supported(c) && !c->opt.force_u64;      // the q-base transforms run on the FP64 kernels there
    return c->qb.pm_class && !c->opt.ntt_nopm && !f64 && (u64)c->k * nd * npow <= 20 && count * c->k * nd * c->k * npow <= 0x7fffffffULL;
}
static int relin_pm(const fhe_ctx *c, const u64 *ct, u64 stride, u32 src_poly, u32 npow, u64 *out2, u64 out_stride, u64 count, const u64 *evk, u32 dbc, u64 *scratch,
                    hipStream_t st) {
    const u32 k = c->k, n = c->n, nd = fhe_evk_digits(c, dbc);
    u64 *dig = scratch, *acc = dig + count * k * nd * k * n * npow;
    const RnsBase base = c->qb.dev();
#define GO_PM(CC)                                                                                                                          \
    DISPATCH_L(c->logn, {                                                                                                                  \
        k_relin_fwd_pm<L, CC><<<(unsigned)(count * k * nd * k * npow), NttShape<L>::TP, 0, st>>>(ct, stride, dig, base, nd, dbc, src_poly, npow); \
        if (!c->opt.relin_fused) {                                                                                                        \
            k_relin_accum_pm<CC><<<grid2(n, count * k), 256, 0, st>>>(dig, evk, acc, base, n, nd, count, npow);                              \
            k_relin_inv_add_pm<L, CC><<<(unsigned)(count * 2 * k), NttShape<L>::TP, 0, st>>>(ct, stride, out2, out_stride, acc, base);     \
        } else {                                                                                                                           \
            k_relin_accum_inv_add_pm<L, CC><<<(unsigned)(count * 2 * k), NttShape<L>::TP, 0, st>>>(ct, stride, out2, out_stride, dig, evk, base, nd * npow); \
        }                                                                                                                                  \
    })
    if (c->qb.pm_class == 1) { GO_PM(PmA); } else { GO_PM(PmB); }
#undef GO_PM
    KERNEL_CHECK();
    return FHE_OK;
}

extern "C" int fhe_relinearize_to(const fhe_ctx *cc, const uint64_t *ct3, uint64_t stride, uint64_t *out2, uint64_t out_stride, uint64_t count,
                                  const uint64_t *evk, uint32_t dbc, void *scratch, size_t scratch_bytes, fhe_stream s) {
    return fhe_relinearize_poly(cc, ct3, stride, 2, out2, out_stride, count, evk, dbc, scratch, scratch_bytes, s);
}
// One key-switch step (SEAL 2.3 relinearize_one_step): polynomial `src_poly` (the last one of a size src_poly + 1 ciphertext) is
// decomposed and folded into c0 / c1 with the keys for s^src_poly; out gets c0', c1' only.
extern "C" int fhe_relinearize_poly(const fhe_ctx *cc, const uint64_t *ct3, uint64_t stride, uint32_t src_poly, uint64_t *out2, uint64_t out_stride, uint64_t count,
                                    const uint64_t *evk, uint32_t dbc, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!cc || !ct3 || !evk || !out2) return fail(FHE_ERR_PARAM, "null argument");
    if (dbc < 1 || dbc > 60) return fail(FHE_ERR_PARAM, "decomposition bit count out of range");
    if (!count) return FHE_OK;
    const fhe_ctx *c = cc;
    int rc;
    if (int erc = fhe_behz_ensure(c)) return erc;
    if (src_poly < 2 || src_poly >= FHE_MAX_POLYS) return fail(FHE_ERR_PARAM, "key-switch source polynomial %u out of range", src_poly);
    if (stride < (u64)(src_poly + 1) * c->k * c->n) return fail(FHE_ERR_PARAM, "ciphertext stride smaller than a size-%u ciphertext", src_poly + 1);
    if (out_stride < (u64)2 * c->k * c->n) return fail(FHE_ERR_PARAM, "output stride smaller than a size-2 ciphertext");
    if (!scratch || scratch_bytes < fhe_relinearize_scratch_bytes(c, dbc, count)) return fail(FHE_ERR_PARAM, "scratch too small");
    // workgroup c reads ciphertext c while it writes output c: only the exact in-place case (same pointer, same stride) and fully
    // disjoint ranges are safe -- with another stride output c lands inside an input another workgroup has not read yet
    if (!(out2 == ct3 && out_stride == stride)) {
        const uintptr_t i0 = (uintptr_t)ct3, i1 = i0 + ((count - 1) * stride + (u64)(src_poly + 1) * c->k * c->n) * sizeof(u64);
        const uintptr_t o0 = (uintptr_t)out2, o1 = o0 + ((count - 1) * out_stride + (u64)2 * c->k * c->n) * sizeof(u64);
        if (o0 < i1 && i0 < o1) return fail(FHE_ERR_PARAM, "output range overlaps the input range (only out2 == ct3 with out_stride == stride may alias)");
    }
    hipStream_t st = (hipStream_t)s;
    const u32 k = c->k, n = c->n, nd = fhe_evk_digits(c, dbc);
    const BehzDev *T = c->behz->dev;
    u64 *dig = (u64 *)scratch, *acc = dig + count * k * nd * k * n;
    if (relin_pm_ok(c, nd, 1, count)) return relin_pm(c, (const u64 *)ct3, stride, src_poly, 1, (u64 *)out2, out_stride, count, (const u64 *)evk, dbc, (u64 *)scratch, st);
    k_relin_digits<<<grid2(n, count * k * nd), 256, 0, st>>>((const u64 *)ct3, stride, dig, T, n, nd, dbc, count, src_poly);
    if ((rc = qbase_ntt(false, c, dig, dig, count * k * nd, st))) return rc;
    k_relin_accum<<<grid2(n, count * k), 256, 0, st>>>(dig, (const u64 *)evk, acc, T, n, nd, count);
    if ((rc = qbase_ntt(true, c, acc, acc, count * 2, st))) return rc;
    k_relin_add<<<grid2(n, count * 2 * k), 256, 0, st>>>((const u64 *)ct3, stride, (u64 *)out2, out_stride, acc, T, n, count);
    KERNEL_CHECK();
    return FHE_OK;
}
