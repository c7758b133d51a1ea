// dct_u64.hip -- encrypted_dct + quantize_fhe (homo/fhe_image.h:196-305) as TWO fused launches in 64-bit
// integer (Shoup / Harvey) arithmetic, for coefficient moduli the exact-FP64 kernels of dct_fused.hip cannot
// take: primes of 48..57 bits -- in particular the moduli SEAL 2.3's coeff_modulus_128 really returns
// (54/55-bit, homo/server_jpeg.cpp:78).  Same dataflow as the FP64 pair:
//   kernel A (rows):    x_m = d_m +- d_(7-m) on coefficients, four joint forward NTTs (every twiddle shared
//                       by the four polynomials), the even / odd half of the LL&M row pass per NTT slot,
//                       four NTT-form row outputs to the intermediate
//   kernel B (columns): the same on the intermediates for a column, per-output scale
//                       encode(0.125)*encode(1/quant) folded into one product, four joint inverse NTTs
// instead of the general path's NTT -> 64-values-per-slot kernel -> inverse NTT (three launches, every
// polynomial through HBM three times each way).
//
// Lazy ranges (q < 2^57 so that 128 q < 2^64).  Every product is mul_shoup_lazy4 (modarith.h: v_mad_u64_u32 only,
// approximate high word): ANY 64-bit operand in, [0, 4q) out.  Forward butterflies, primes <= 56 bits: no
// conditional subtraction at all -- x0' = X + T, x1' = X - T + 4q grow by 4q per stage, (2 + 4 log2 n) q <= 54 q at the
// end, and the per-slot circuit works on those (4 x 54 q < 2^64 / q holds up to 56 bits); 57-bit primes keep one
// conditional subtraction per butterfly (Harvey with doubled ranges: values in [0, 8q)).  Sums inside the per-slot
// circuit are left unreduced (bounds in the comments of line_half); row outputs 0 and 4, which meet no constant, are
// brought below 4q by a product with 1 so that every row output is below 16 q; the scale product brings column
// outputs to [0, 4q), the Gentleman-Sande inverse butterflies track their ranges statically (inv_stage), two conditional subtractions at the store
// give canonical residues.  The ciphertexts are bit-identical to the op-at-a-time evaluation (exact ring
// arithmetic); tests/test_gpu_parity.py compares with the oracle.
#include "internal.h"

#include <cstdlib>

namespace {

struct Work { u32 blk, line, poly, prime, half; };
// the two halves of a line sit 8 apart in blockIdx (same XCD: the partner's re-read of the eight inputs is an
// L2 hit) and the prime is the slowest index (resident workgroups share one prime's twiddles and constants)
__device__ __forceinline__ Work decode(u32 idx, u32 k) {
    const u32 w = ((idx >> 4) << 3) | (idx & 7);
    const u32 per_prime = (gridDim.x >> 1) / k;
    Work o;
    o.half = (idx >> 3) & 1;
    o.prime = w / per_prime;
    u32 t = w - o.prime * per_prime;
    o.poly = t & 1;
    t >>= 1;
    o.line = t & 7;
    o.blk = t >> 3;
    return o;
}

constexpr int LE = 3, E = 8;                     // coefficients per thread
template <int L> struct Sh { static constexpr int N = 1 << L, TP = N >> LE, NP = (L + LE - 1) / LE, LDS_WORDS = N + (N >> LE); };
__host__ __device__ constexpr int p_lo(int L, int p) { return (L - LE * p - LE) < 0 ? 0 : (L - LE * p - LE); }
__host__ __device__ constexpr int p_stages(int L, int p) { return (L - LE * p) > LE ? LE : (L - LE * p); }
template <int LO> __device__ __forceinline__ int e_index(int tid, int r) { return ((tid >> LO) << (LO + LE)) | (r << LO) | (tid & ((1 << LO) - 1)); }
template <int LO> __device__ __forceinline__ int e_pad(int j) { return j + ((j >> (LO + LE)) << LO); }

// twiddles of one register pass: 2^(LE-1-rb) per stage, 7 in all, each a (value, Shoup companion) pair
template <int L, int P> struct Tw {
    static constexpr int LO = p_lo(L, P), S = p_stages(L, P);
    static constexpr int rb(int u) { return (L - 1 - (LE * P + u)) - LO; }
    static constexpr int count(int u) { return 1 << (LE - 1 - rb(u)); }
    static constexpr int offset(int u) { int o = 0; for (int v = 0; v < u; v++) o += count(v); return o; }
};
// twiddles of ONE stage (1, 2 or 4 pairs): fetched one stage ahead of their use, so that at most eight pairs
// are live (a whole pass's seven pairs plus the next pass's would not fit the 128-VGPR budget next to the data)
template <int L, int P, int U>
__device__ __forceinline__ void load_stage(ulonglong2 (&w)[4], const ulonglong2 *__restrict__ tw, int tid) {
    using T = Tw<L, P>;
    const int th = (P == 0) ? 0 : (tid >> T::LO);
#pragma unroll
    for (int i = 0; i < T::count(U); i++) w[i] = tw[(1 << (LE * P + U)) + ((th << (LE - 1 - T::rb(U))) | i)];
}

// the same for the pseudo-Mersenne tables of this geometry (BaseTables::d_tw_pm3: (w, w 2^31 mod q) pairs, the twiddles
// of a stage as [i][th] so that consecutive lanes read consecutive pairs)
template <int L, int P, int U>
__device__ __forceinline__ void load_stage_pm(ulonglong2 (&w)[4], const ulonglong2 *__restrict__ tw, int tid) {
    using T = Tw<L, P>;
    constexpr int sigma = LE * P + U;
    const int th = (P == 0) ? 0 : (tid >> T::LO);
#pragma unroll
    for (int i = 0; i < T::count(U); i++) w[i] = tw[(1 << sigma) + (i << (sigma - (LE - 1 - T::rb(U)))) + th];
}
template <int L, int P, int U, bool PM>
__device__ __forceinline__ void load_stage_any(ulonglong2 (&w)[4], const ulonglong2 *__restrict__ tw, int tid) {
    if constexpr (PM) load_stage_pm<L, P, U>(w, tw, tid);
    else load_stage<L, P, U>(w, tw, tid);
}

// per-workgroup constants of one prime
struct Prime { u64 q, nq, q4, one_p; u32 zero; PmMod pm; };     // nq = 2^64 - q, q4 = 4q, one_p = floor(2^64 / q), zero: modarith.h
__device__ __forceinline__ Prime prime_of(const Modulus &m) {
    Prime o;
    o.q = m.q; o.nq = 0 - m.q; o.q4 = 4 * m.q; o.zero = fhe_opaque_zero;
    o.one_p = one_companion(m);
    o.pm.q = m.q; o.pm.delta = (u32)((1ULL << (m.s1 + 1)) - m.q); o.pm.sh = m.s1 + 1 - 32; o.pm.mb = (1u << (m.s1 + 1 - 32)) - 1; o.pm.pad = 0;
    return o;
}
// bound (in units of q) of the forward transform's outputs
template <int L, bool LAZY, bool PM = false> struct Bn { static constexpr u64 V = PM ? 2 + (L << PmA::CS) : LAZY ? 2 + 4 * L : 8; };

// Cooley-Tukey stage U of pass P on four polynomials.  LAZY: values grow by 4q per stage; otherwise [0, 8q) in and out
template <int L, int P, int U, bool LAZY>
__device__ __forceinline__ void fwd_stage(u64 (&x)[4][E], const ulonglong2 (&w)[4], const Prime &pr) {
    using T = Tw<L, P>;
    constexpr int rb = T::rb(U);
#pragma unroll
    for (int b = 0; b < E / 2; b++) {
        const int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
        const ulonglong2 wv = w[r0 >> (rb + 1)];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const u64 X = LAZY ? x[m][r0] : csub(x[m][r0], pr.q4);
            const u64 S = mul_shoup_lazy4_acc(x[m][r1], wv.x, wv.y, pr.nq, pr.zero, X);      // X + T
            x[m][r0] = S;
            x[m][r1] = (X << 1) + pr.q4 - S;                                                // X - T + 4q
        }
    }
}
// The same on the pseudo-Mersenne product (ntt_core.h; every prime <= 55 bits, class PmA): values grow by 2^CS q = 8q per
// stage from 2q, no conditional subtraction and no fold up to n = 8192 ((2 + 8 x 13) q < 128 q, the product's operand limit)
template <int L, int P, int U>
__device__ __forceinline__ void fwd_stage_pm(u64 (&x)[4][E], const ulonglong2 (&w)[4], const Prime &pr) {
    using T = Tw<L, P>;
    constexpr int rb = T::rb(U);
    static_assert(32 + ((L - 1) << (4 + PmA::CS)) <= PmA::LIM, "forward operands pass the product's limit");
    const u64 off = pr.q << PmA::CS;
#pragma unroll
    for (int b = 0; b < E / 2; b++) {
        const int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
        const ulonglong2 wv = w[r0 >> (rb + 1)];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const u64 X = x[m][r0], T2 = mul_pm(x[m][r1], wv, pr.pm);
            x[m][r0] = X + T2;
            x[m][r1] = X - T2 + off;
        }
    }
}
template <int L, int P, bool LAZY, bool PM, int U = 0>
__device__ __forceinline__ void fwd_pass(u64 (&x)[4][E], const ulonglong2 (&w)[4], const ulonglong2 *__restrict__ tw, const Prime &pr, int tid) {
    if constexpr (U + 1 < Tw<L, P>::S) {
        ulonglong2 wn[4];
        load_stage_any<L, P, U + 1, PM>(wn, tw, tid);
        if constexpr (PM) fwd_stage_pm<L, P, U>(x, w, pr);
        else fwd_stage<L, P, U, LAZY>(x, w, pr);
        fwd_pass<L, P, LAZY, PM, U + 1>(x, wn, tw, pr, tid);
    } else {
        if constexpr (PM) fwd_stage_pm<L, P, U>(x, w, pr);
        else fwd_stage<L, P, U, LAZY>(x, w, pr);
    }
}
// Bound (in units of q) of register r after the stages S-1 ... U of inverse pass P have run (U = S: at entry).  A sum
// X + Y is left unreduced and doubles the bound (both sides of a butterfly share their history, hence their bound); a
// product resets its register to [0, 4q).  Entry: 4 for the first pass executed (the scale products), 8 after an exchange.
template <int L, int P>
__host__ __device__ constexpr int inv_bd(int r, int U) {
    int bd = (P == Sh<L>::NP - 1) ? 4 : 8;
    for (int u = Tw<L, P>::S - 1; u >= U; u--) bd = ((r >> Tw<L, P>::rb(u)) & 1) ? 4 : 2 * bd;
    return bd;
}
// Gentleman-Sande stage U of pass P with static range tracking (q < 2^57: 128q < 2^64): sums stay unreduced inside a
// pass (at most 8 x 2^3 = 64q), differences get the partner's bound added, products take any 64-bit operand; after the
// last stage of a pass every register is brought below 8q again (7 conditional subtractions per 8 coefficients and
// three stages instead of 12).  The last pass ends with both sides through a product: [0, 4q) out.
// n^-1 is merged into the last stage of the transform.
template <int L, int P, int U>
__device__ __forceinline__ void inv_stage(u64 (&x)[4][E], const ulonglong2 (&w)[4], const ulonglong2 ninv, const Prime &pr) {
    using T = Tw<L, P>;
    constexpr int sigma = LE * P + U, rb = T::rb(U);
#pragma unroll
    for (int b = 0; b < E / 2; b++) {
        const int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
        const ulonglong2 wv = w[r0 >> (rb + 1)];
        const int bd = inv_bd<L, P>(r1, U + 1);                        // bound of Y (and of X) before this stage: 4 ... 32
        const u64 off = pr.q4 << (bd == 4 ? 0 : bd == 8 ? 1 : bd == 16 ? 2 : 3);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const u64 X = x[m][r0], Y = x[m][r1];
            const u64 Tm = X + Y;                    // < 2 bd q <= 64q
            const u64 D = X - Y + off;               // (0, 2 bd q)
            x[m][r0] = (sigma == 0) ? mul_shoup_lazy4(Tm, ninv.x, ninv.y, pr.nq, pr.zero) : Tm;
            x[m][r1] = mul_shoup_lazy4(D, wv.x, wv.y, pr.nq, pr.zero);
        }
    }
    if constexpr (U == 0 && P > 0) {
#pragma unroll
        for (int r = 0; r < E; r++) {
            const int bd = inv_bd<L, P>(r, 0);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (bd > 32) x[m][r] = csub(x[m][r], pr.q4 << 3);
                if (bd > 16) x[m][r] = csub(x[m][r], pr.q4 << 2);
                if (bd > 8) x[m][r] = csub(x[m][r], pr.q4 << 1);
            }
        }
    }
}
// Pseudo-Mersenne inverse (class PmA) with the ranges tracked in sixteenths of q like ntt_core.h's pm_inv_plan, for this
// file's geometry (8 registers, passes of three stages): entry 4q (the scale products) for the first pass executed, XB
// after an exchange; a sum stays unreduced, a difference gets 2^s q >= bound(Y) added, an operand is folded (one
// multiply-add) only when the difference would pass the product's limit; registers above XB are folded at a pass's end.
constexpr int PM3_E0 = 64, PM3_XB = 192;
struct Pm3Plan { bool fold_y[LE][E], fold_x[LE][E]; int shift[LE][E]; bool fold_exit[E]; };
template <int L, int P>
__host__ __device__ constexpr Pm3Plan pm3_plan() {
    Pm3Plan pl{};
    int bd[E] = {};
    for (int r = 0; r < E; r++) bd[r] = (P == Sh<L>::NP - 1) ? PM3_E0 : PM3_XB;
    for (int u = Tw<L, P>::S - 1; u >= 0; u--) {
        const int sigma = LE * P + u, rb = Tw<L, P>::rb(u);
        for (int r0 = 0; r0 < E; r0++) {
            if (r0 & (1 << rb)) continue;
            const int r1 = r0 | (1 << rb);
            if (bd[r0] + (16 << pm_ceil_log2_q(bd[r1])) > PmA::LIM) { pl.fold_y[u][r0] = true; bd[r1] = PM_FOLDED; }
            if (bd[r0] + (16 << pm_ceil_log2_q(bd[r1])) > PmA::LIM) { pl.fold_x[u][r0] = true; bd[r0] = PM_FOLDED; }
            pl.shift[u][r0] = pm_ceil_log2_q(bd[r1]);
            bd[r0] = sigma == 0 ? PmA::RQ : bd[r0] + bd[r1];
            bd[r1] = PmA::RQ;
        }
    }
    for (int r = 0; r < E; r++) pl.fold_exit[r] = P > 0 && bd[r] > PM3_XB;
    return pl;
}
template <int L, int P, int U, int B>
__device__ __forceinline__ void inv_bfly_pm(u64 (&x)[4][E], const ulonglong2 (&w)[4], const ulonglong2 ninv, const Prime &pr) {
    constexpr Pm3Plan pl = pm3_plan<L, P>();
    constexpr int sigma = LE * P + U, rb = Tw<L, P>::rb(U);
    constexpr int r0 = ((B >> rb) << (rb + 1)) | (B & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
    const ulonglong2 wv = w[r0 >> (rb + 1)];
    const u64 off = pr.q << pl.shift[U][r0];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        u64 X = x[m][r0], Y = x[m][r1];
        if constexpr (pl.fold_y[U][r0]) Y = fold_pm(Y, pr.pm);
        if constexpr (pl.fold_x[U][r0]) X = fold_pm(X, pr.pm);
        const u64 Tm = X + Y;
        const u64 D = X - Y + off;
        if constexpr (sigma == 0) x[m][r0] = mul_pm(Tm, ninv, pr.pm);
        else x[m][r0] = Tm;
        x[m][r1] = mul_pm(D, wv, pr.pm);
    }
}
template <int L, int P, int R>
__device__ __forceinline__ void inv_exit_pm(u64 (&x)[4][E], const Prime &pr) {
    constexpr Pm3Plan pl = pm3_plan<L, P>();
    if constexpr (pl.fold_exit[R]) {
#pragma unroll
        for (int m = 0; m < 4; m++) x[m][R] = fold_pm(x[m][R], pr.pm);
    }
}
template <int L, int P, int U>
__device__ __forceinline__ void inv_stage_pm(u64 (&x)[4][E], const ulonglong2 (&w)[4], const ulonglong2 ninv, const Prime &pr) {
    inv_bfly_pm<L, P, U, 0>(x, w, ninv, pr);
    inv_bfly_pm<L, P, U, 1>(x, w, ninv, pr);
    inv_bfly_pm<L, P, U, 2>(x, w, ninv, pr);
    inv_bfly_pm<L, P, U, 3>(x, w, ninv, pr);
    if constexpr (U == 0 && P > 0) {
        inv_exit_pm<L, P, 0>(x, pr); inv_exit_pm<L, P, 1>(x, pr); inv_exit_pm<L, P, 2>(x, pr); inv_exit_pm<L, P, 3>(x, pr);
        inv_exit_pm<L, P, 4>(x, pr); inv_exit_pm<L, P, 5>(x, pr); inv_exit_pm<L, P, 6>(x, pr); inv_exit_pm<L, P, 7>(x, pr);
    }
}
template <int L, int P, bool PM, int U = Tw<L, P>::S - 1>
__device__ __forceinline__ void inv_pass(u64 (&x)[4][E], const ulonglong2 (&w)[4], const ulonglong2 *__restrict__ itw, const ulonglong2 ninv, const Prime &pr, int tid) {
    if constexpr (U > 0) {
        ulonglong2 wn[4];
        load_stage_any<L, P, U - 1, PM>(wn, itw, tid);
        if constexpr (PM) inv_stage_pm<L, P, U>(x, w, ninv, pr);
        else inv_stage<L, P, U>(x, w, ninv, pr);
        inv_pass<L, P, PM, U - 1>(x, wn, itw, ninv, pr, tid);
    } else {
        if constexpr (PM) inv_stage_pm<L, P, U>(x, w, ninv, pr);
        else inv_stage<L, P, U>(x, w, ninv, pr);
    }
}

// four polynomials through two alternating LDS buffers; exchanges whose lane-to-index maps keep the wave bits
// of the thread id fixed (both LO <= 6) need no workgroup barrier: a wave only reads what it wrote itself
template <int L, int LO_FROM, int LO_TO>
__device__ __forceinline__ void transpose(u64 (&x)[4][E], u64 *lds, int tid, int &phase) {
    constexpr int PL = LO_FROM < LO_TO ? LO_FROM : LO_TO;
    constexpr bool WAVE_LOCAL = (Sh<L>::TP <= 64) || (LO_FROM <= 6 && LO_TO <= 6);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        u64 *buf = lds + (phase & 1) * Sh<L>::LDS_WORDS;
        phase++;
#pragma unroll
        for (int r = 0; r < E; r++) buf[e_pad<PL>(e_index<LO_FROM>(tid, r))] = x[m][r];
        if constexpr (WAVE_LOCAL) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < E; r++) x[m][r] = buf[e_pad<PL>(e_index<LO_TO>(tid, r))];
    }
}

// `w` holds the twiddles of the first stage of pass P; the first stage of the next pass is fetched before the
// LDS exchange so that its latency hides behind it
template <int L, bool LAZY, bool PM, int P = 0>
__device__ __forceinline__ void ntt_fwd(u64 (&x)[4][E], ulonglong2 (&w)[4], const ulonglong2 *__restrict__ tw, const Prime &pr, u64 *lds, int tid, int &phase) {
    fwd_pass<L, P, LAZY, PM>(x, w, tw, pr, tid);
    if constexpr (P + 1 < Sh<L>::NP) {
        ulonglong2 wn[4];
        load_stage_any<L, P + 1, 0, PM>(wn, tw, tid);
        transpose<L, p_lo(L, P), p_lo(L, P + 1)>(x, lds, tid, phase);
        ntt_fwd<L, LAZY, PM, P + 1>(x, wn, tw, pr, lds, tid, phase);
    }
}
template <int L, bool PM, int P = Sh<L>::NP - 1>
__device__ __forceinline__ void ntt_inv(u64 (&x)[4][E], ulonglong2 (&w)[4], const ulonglong2 *__restrict__ itw, const ulonglong2 ninv, const Prime &pr, u64 *lds, int tid, int &phase) {
    inv_pass<L, P, PM>(x, w, itw, ninv, pr, tid);
    if constexpr (P > 0) {
        ulonglong2 wn[4];
        load_stage_any<L, P - 1, Tw<L, P - 1>::S - 1, PM>(wn, itw, tid);
        transpose<L, p_lo(L, P), p_lo(L, P - 1)>(x, lds, tid, phase);
        ntt_inv<L, PM, P - 1>(x, wn, itw, ninv, pr, lds, tid, phase);
    }
}

// Even / odd half of one LL&M line (homo/fhe_image.h:215-242) on one NTT slot.  In: x[m] < B q (B = Bn in the row
// kernel, 32 in the column kernel).  Out: x[m] = line output 2m + HALF: even outputs 0 and 4 below 4B q (RED04: brought
// below 4q), outputs 2 and 6 below 8q, odd outputs below 16q.  `sub` = B q, the multiple of q added before a
// subtraction.  4B q < 2^64 is the caller's condition (216 q for B = 54, 128 q for B = 32).
template <int HALF, bool RED04, typename CF>
__device__ __forceinline__ void line_half(u64 &x0, u64 &x1, u64 &x2, u64 &x3, const Prime &pr, u64 sub, CF C) {
    auto MUL = [&](u64 v, int cid) { const ulonglong2 w = C(cid); return mul_shoup_lazy4(v, w.x, w.y, pr.nq, pr.zero); };    // any v -> [0, 4q)
    if constexpr (HALF == 0) {                       // x = tmp0..tmp3
        const u64 tmp10 = x0 + x3, tmp13 = x0 - x3 + sub, tmp11 = x1 + x2, tmp12 = x1 - x2 + sub;      // < 2B q
        const u64 z1 = MUL(tmp12 + tmp13, 0);        // operand < 4B q
        x0 = tmp10 + tmp11;                          // out 0, < 4B q
        x2 = tmp10 - tmp11 + 2 * sub;                // out 4, < 4B q
        if constexpr (RED04) { x0 = reduce_lazy4(x0, pr.one_p, pr.nq, pr.zero); x2 = reduce_lazy4(x2, pr.one_p, pr.nq, pr.zero); }
        x1 = z1 + MUL(tmp13, 1);                     // out 2, < 8q
        x3 = z1 + MUL(tmp12, 2);                     // out 6
    } else {                                         // x = tmp7, tmp6, tmp5, tmp4
        const u64 tmp7 = x0, tmp6 = x1, tmp5 = x2, tmp4 = x3;
        const u64 z1 = tmp4 + tmp7, z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;              // < 2B q
        const u64 z5 = MUL(z3 + z4, 3);              // operand < 4B q
        const u64 t4 = MUL(tmp4, 4), t5 = MUL(tmp5, 5), t6 = MUL(tmp6, 6), t7 = MUL(tmp7, 7);
        const u64 m1 = MUL(z1, 8), m2 = MUL(z2, 9), m3 = MUL(z3, 10) + z5, m4 = MUL(z4, 11) + z5;      // m3, m4 < 8q
        x0 = t7 + m1 + m4;                           // out 1, < 16q
        x1 = t6 + m2 + m3;                           // out 3
        x2 = t5 + m2 + m4;                           // out 5
        x3 = t4 + m1 + m3;                           // out 7
    }
}

template <int L, int HALF, bool LAZY, bool PM>
__device__ __forceinline__ void rows_body(const u64 *__restrict__ in, u64 *__restrict__ mid, const ulonglong2 *__restrict__ consts,
                                          const ulonglong2 *__restrict__ tw, const Work &wk, const Prime &pr, u32 k, u64 *lds) {
    constexpr int N = Sh<L>::N, TP = Sh<L>::TP;
    const int tid = threadIdx.x;
    const size_t poly_words = (size_t)k * N, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + 8 * wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * N;
    ulonglong2 w0[4];
    load_stage_any<L, 0, 0, PM>(w0, tw, tid);
    u64 x[4][E];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const u64 *a = in + base + (size_t)m * ct_words + tid, *b = in + base + (size_t)(7 - m) * ct_words + tid;
#pragma unroll
        for (int r = 0; r < E; r++) {                // pass-0 mapping: coefficient r*TP + tid; canonical inputs
            const u64 A = a[r * TP], B = b[r * TP];
            x[m][r] = HALF ? A - B + pr.q : A + B;   // [0, 2q)
        }
    }
    int phase = 0;
    ntt_fwd<L, LAZY, PM>(x, w0, tw, pr, lds, tid, phase);    // below Bn q, slot j = (tid << 3) + r at position r*TP + tid
    const ulonglong2 *cp = consts + (size_t)wk.prime * N + tid;
    const size_t cstride = (size_t)k * N;
#pragma unroll
    for (int r = 0; r < E; r++) {
        auto C = [&](int cid) { return cp[(size_t)cid * cstride + r * TP]; };
        line_half<HALF, true>(x[0][r], x[1][r], x[2][r], x[3][r], pr, Bn<L, LAZY, PM>::V * pr.q, C);
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {                    // row outputs below 16 q
        u64 *o = mid + base + (size_t)(2 * m + HALF) * ct_words + tid;
#pragma unroll
        for (int r = 0; r < E; r++) o[r * TP] = x[m][r];
    }
}

template <int L, int HALF, bool PM>
__device__ __forceinline__ void cols_body(const u64 *__restrict__ mid, u64 *__restrict__ out, const ulonglong2 *__restrict__ consts,
                                          const ulonglong2 *__restrict__ itw, const Work &wk, const Prime &pr, u32 k, u64 *lds) {
    constexpr int N = Sh<L>::N, TP = Sh<L>::TP;
    const int tid = threadIdx.x;
    const size_t poly_words = (size_t)k * N, ct_words = 2 * poly_words;
    const size_t base = ((size_t)wk.blk * 64 + wk.line) * ct_words + (size_t)wk.poly * poly_words + (size_t)wk.prime * N;
    const size_t row_stride = 8 * ct_words, cstride = (size_t)k * N;
    const u64 sub16 = 4 * pr.q4;
    u64 x[4][E];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const u64 *a = mid + base + (size_t)m * row_stride + tid, *b = mid + base + (size_t)(7 - m) * row_stride + tid;
#pragma unroll
        for (int r = 0; r < E; r++) {
            const u64 A = a[r * TP], B = b[r * TP];  // < 16 q each
            x[m][r] = HALF ? A - B + sub16 : A + B;  // < 32 q
        }
    }
    const ulonglong2 *cp = consts + (size_t)wk.prime * N + tid;
    // per-output scale: row 2m+HALF, column wk.line -> constant 12 + 8*row + col
    const ulonglong2 *sp = consts + (size_t)(12 + 8 * HALF + wk.line) * cstride + (size_t)wk.prime * N + tid;
#pragma unroll
    for (int r = 0; r < E; r++) {
        auto C = [&](int cid) { return cp[(size_t)cid * cstride + r * TP]; };
        line_half<HALF, false>(x[0][r], x[1][r], x[2][r], x[3][r], pr, 2 * sub16, C);     // outputs below 128 q
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const ulonglong2 s = sp[(size_t)(16 * m) * cstride + r * TP];
            x[m][r] = mul_shoup_lazy4(x[m][r], s.x, s.y, pr.nq, pr.zero);                 // [0, 4q)
        }
        asm volatile("" ::: "memory");     // one slot's constants in flight at a time (four waves per SIMD hide the round trip)
    }
    int phase = 0;
    ulonglong2 wl[4];
    load_stage_any<L, Sh<L>::NP - 1, Tw<L, Sh<L>::NP - 1>::S - 1, PM>(wl, itw, tid);
    ntt_inv<L, PM>(x, wl, itw, itw[0], pr, lds, tid, phase);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        u64 *o = out + base + (size_t)(2 * m + HALF) * row_stride + tid;
#pragma unroll
        for (int r = 0; r < E; r++) o[r * TP] = PM ? csub(fold_pm(x[m][r], pr.pm), pr.q) : csub(csub(x[m][r], 2 * pr.q), pr.q);
    }
}

__host__ __device__ constexpr int occ_w(int tp, int lds_words) { return ((2 * 2 * lds_words * 8 <= 160 * 1024) ? 2 : 1) * tp / 256 < 1 ? 1 : ((2 * 2 * lds_words * 8 <= 160 * 1024) ? 2 : 1) * tp / 256; }

// PM: pseudo-Mersenne butterflies (every prime <= 55 bits, class PmA) on the tables tw3 / itw3
template <int L, bool LAZY, bool PM>
__global__ __launch_bounds__((Sh<L>::TP), (occ_w(Sh<L>::TP, Sh<L>::LDS_WORDS))) void k_dct_rows_u64(const u64 *__restrict__ in, u64 *__restrict__ mid,
                                                                                      const ulonglong2 *__restrict__ consts, RnsBase base, const ulonglong2 *__restrict__ tw3, u32 k) {
    __shared__ u64 lds[2 * Sh<L>::LDS_WORDS];
    const Work wk = decode(blockIdx.x, k);
    const Prime pr = prime_of(base.mod[wk.prime]);
    const ulonglong2 *tw = (PM ? tw3 : base.tw) + (size_t)wk.prime * Sh<L>::N;
    if (wk.half) rows_body<L, 1, LAZY, PM>(in, mid, consts, tw, wk, pr, k, lds);
    else rows_body<L, 0, LAZY, PM>(in, mid, consts, tw, wk, pr, k, lds);
}
template <int L, bool PM>
__global__ __launch_bounds__((Sh<L>::TP), (occ_w(Sh<L>::TP, Sh<L>::LDS_WORDS))) void k_dct_cols_u64(const u64 *__restrict__ mid, u64 *__restrict__ out,
                                                                                      const ulonglong2 *__restrict__ consts, RnsBase base, const ulonglong2 *__restrict__ itw3, u32 k) {
    __shared__ u64 lds[2 * Sh<L>::LDS_WORDS];
    const Work wk = decode(blockIdx.x, k);
    const Prime pr = prime_of(base.mod[wk.prime]);
    const ulonglong2 *itw = (PM ? itw3 : base.itw) + (size_t)wk.prime * Sh<L>::N;
    if (wk.half) cols_body<L, 1, PM>(mid, out, consts, itw, wk, pr, k, lds);
    else cols_body<L, 0, PM>(mid, out, consts, itw, wk, pr, k, lds);
}

// constants from the u64 kernels' slot order (16 slots per thread) to this file's (8 per thread):
// bit-reversed index j = (t << 3) + r lives at r * (n >> 3) + t here, at (j & 15) * (n >> 4) + (j >> 4) there
__global__ void k_consts_to_le3(const ulonglong2 *__restrict__ in, ulonglong2 *__restrict__ out, u32 n, u32 total) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u32 pos = i % n, rowbase = i - pos, tp = n >> 3, r = pos / tp, t = pos % tp;
    const u32 j = (t << 3) + r;
    out[i] = in[rowbase + (j & 15) * (n >> 4) + (j >> 4)];
}

}  // namespace

bool fhe_dct_u64_supported(const fhe_ctx *c) {
    return c && c->opt.dct_u64_fused && c->max_prime_bits <= 57 && (c->logn == 11 || c->logn == 12 || c->logn == 13);
}

int fhe_dct_u64_make_consts(const fhe_ctx *c, fhe_dct_plan *plan, hipStream_t st) {
    const u32 total = DCT_NCONST * c->k * c->n;
    HIP_TRY(hipMalloc(&plan->d_consts_le3, sizeof(ulonglong2) * total));
    k_consts_to_le3<<<(total + 255) / 256, 256, 0, st>>>(plan->d_consts, plan->d_consts_le3, c->n, total);
    KERNEL_CHECK();
    return FHE_OK;
}

int fhe_dct_u64_launch(const fhe_ctx *c, const fhe_dct_plan *plan, const u64 *in, u64 *out, u64 n_blocks, u64 *mid, hipStream_t st) {
    const u64 grid = n_blocks * 8 * 2 * c->k * 2;      // (block, line, poly, prime) x two halves
    if (grid > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many blocks for one launch");
    const RnsBase base = c->qb.dev();
    const bool lazy = c->max_prime_bits <= 56;     // (2 + 4 log2 n) q x 4 must stay below 2^64
    const bool pm = c->qb.pm_class == 1 && c->qb.d_tw_pm3 && !c->opt.ntt_nopm;      // every prime <= 55 bits and pseudo-Mersenne: (2 + 8 log2 n) q x 4 < 2^64 too
    const ulonglong2 *tw3 = c->qb.d_tw_pm3, *itw3 = c->qb.d_itw_pm3;
    switch (c->logn) {
#define GO(LL) case LL: if (pm) k_dct_rows_u64<LL, true, true><<<(unsigned)grid, Sh<LL>::TP, 0, st>>>(in, mid, plan->d_consts_le3, base, tw3, c->k); \
                        else if (lazy) k_dct_rows_u64<LL, true, false><<<(unsigned)grid, Sh<LL>::TP, 0, st>>>(in, mid, plan->d_consts_le3, base, tw3, c->k); \
                        else k_dct_rows_u64<LL, false, false><<<(unsigned)grid, Sh<LL>::TP, 0, st>>>(in, mid, plan->d_consts_le3, base, tw3, c->k); \
                        if (pm) k_dct_cols_u64<LL, true><<<(unsigned)grid, Sh<LL>::TP, 0, st>>>(mid, out, plan->d_consts_le3, base, itw3, c->k); \
                        else k_dct_cols_u64<LL, false><<<(unsigned)grid, Sh<LL>::TP, 0, st>>>(mid, out, plan->d_consts_le3, base, itw3, c->k); break;
        GO(11) GO(12) GO(13)
#undef GO
        default: return fail(FHE_ERR_PARAM, "fused u64 path supports n in {2048, 4096, 8192}");
    }
    KERNEL_CHECK();
    return FHE_OK;
}
