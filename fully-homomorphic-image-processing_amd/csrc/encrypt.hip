// encrypt.hip -- the servers' own encryptions as batches on the device (include/fhe_hip.h, "server-side encryptions").
// The reference encrypts inside its loops: frac(x), frac(y) per output pixel (homo/fhe_resize.h:230,234,262,266), an encode(0)
// per homomorphic_sin / cos (homo/fhe_decode.h:54,134), the index and the accumulators of server_decode (homo/server_decode.cpp:
// 121,126).  One seal::Encryptor::encrypt is Enc(m) = (Delta m' + pk0 u + e1, pk1 u + e2): here `count` of them are
//   k_enc_sample_u      u from the ChaCha20 stream of (key, first_index + i), as residues          [count][k][n]
//   fhe_ntt_forward     in place
//   k_enc_pk_mul        (pk0 u, pk1 u) per slot                                                    [count][2][k][n]
//   fhe_ntt_inverse     in place
//   k_enc_finish        + e1 / e2 from the same stream, + Delta m' on c0
// -- five launches per BATCH where the Python and the facade encryptors made five per ciphertext plus three uploads.
// All sampling is 32/64-bit integer work (no transcendental functions), so the oracle's restatement draws the same values.
#include "internal.h"

#include <cmath>
#include <cstring>

#include "host_math.h"

namespace {
// floor(2^63 P(|e| <= i)), i = 0..18, for the rounded normal with sigma = 3.19 redrawn beyond 19 (tools/noise_cdt.py recomputes the
// table with 90 digits; tests/test_encrypt_sampler.py compares)
#define FHE_NOISE_CDT_VALUES                                                                                                                                   \
    {0x0ff141e3023416d2ULL, 0x2e4f850f76b8d9a6ULL, 0x488c5acec8fd6db3ULL, 0x5d1ca569fc3e4ccbULL, 0x6bbb5699bdd65b9cULL, 0x75291bf8371e7eccULL, 0x7aad3cf138611a69ULL,  \
     0x7d9aa4d4ab7c76bdULL, 0x7f0368341f79807cULL, 0x7fa0f21e3a554470ULL, 0x7fdf5971c6494be2ULL, 0x7ff5c5a33f74a4e1ULL, 0x7ffd148ddcc40605ULL, 0x7fff3db0052c58c3ULL,  \
     0x7fffd206471c7fcfULL, 0x7ffff61ba7b56e58ULL, 0x7ffffe11d76ecb8aULL, 0x7fffffa9c1e61510ULL, 0x7ffffff3ceaa701fULL}
__constant__ u64 kNoiseCdtDev[FHE_NOISE_CDT_LEN] = FHE_NOISE_CDT_VALUES;      // what the kernels compare against
const u64 kNoiseCdtHost[FHE_NOISE_CDT_LEN] = FHE_NOISE_CDT_VALUES;            // what fhe_noise_cdt reports: the same initialiser

struct ChaChaKey { u32 w[8]; };

__device__ __forceinline__ u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
#define CHACHA_QR(a, b, c, d)                \
    a += b; d ^= a; d = rotl32(d, 16);       \
    c += d; b ^= c; b = rotl32(b, 12);       \
    a += b; d ^= a; d = rotl32(d, 8);        \
    c += d; b ^= c; b = rotl32(b, 7);
// one 64-byte block as eight little-endian u64: state = "expand 32-byte k" | key | block counter (64 bit) | nonce (64 bit), 20 rounds
__device__ __forceinline__ void chacha20_block(const ChaChaKey &key, u64 counter, u64 nonce, u64 out[8]) {
    const u32 s0 = 0x61707865u, s1 = 0x3320646eu, s2 = 0x79622d32u, s3 = 0x6b206574u;
    const u32 s12 = (u32)counter, s13 = (u32)(counter >> 32), s14 = (u32)nonce, s15 = (u32)(nonce >> 32);
    u32 x0 = s0, x1 = s1, x2 = s2, x3 = s3, x4 = key.w[0], x5 = key.w[1], x6 = key.w[2], x7 = key.w[3], x8 = key.w[4], x9 = key.w[5], x10 = key.w[6],
        x11 = key.w[7], x12 = s12, x13 = s13, x14 = s14, x15 = s15;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        CHACHA_QR(x0, x4, x8, x12)
        CHACHA_QR(x1, x5, x9, x13)
        CHACHA_QR(x2, x6, x10, x14)
        CHACHA_QR(x3, x7, x11, x15)
        CHACHA_QR(x0, x5, x10, x15)
        CHACHA_QR(x1, x6, x11, x12)
        CHACHA_QR(x2, x7, x8, x13)
        CHACHA_QR(x3, x4, x9, x14)
    }
    out[0] = (u64)(x0 + s0) | ((u64)(x1 + s1) << 32);
    out[1] = (u64)(x2 + s2) | ((u64)(x3 + s3) << 32);
    out[2] = (u64)(x4 + key.w[0]) | ((u64)(x5 + key.w[1]) << 32);
    out[3] = (u64)(x6 + key.w[2]) | ((u64)(x7 + key.w[3]) << 32);
    out[4] = (u64)(x8 + key.w[4]) | ((u64)(x9 + key.w[5]) << 32);
    out[5] = (u64)(x10 + key.w[6]) | ((u64)(x11 + key.w[7]) << 32);
    out[6] = (u64)(x12 + s12) | ((u64)(x13 + s13) << 32);
    out[7] = (u64)(x14 + s14) | ((u64)(x15 + s15) << 32);
}
__device__ __forceinline__ int draw_ternary(u64 r) { return (int)__umul64hi(r, 3) - 1; }
__device__ __forceinline__ int draw_noise(u64 r) {
    const u64 x = r >> 1;
    int m = 0;
#pragma unroll
    for (int i = 0; i < FHE_NOISE_CDT_LEN; ++i) m += x >= kNoiseCdtDev[i] ? 1 : 0;
    return (r & 1) ? -m : m;
}

// one thread = one ChaCha block = eight consecutive coefficients of one encryption's u, written as its k residues
__global__ __launch_bounds__(256) void k_enc_sample_u(ChaChaKey key, u64 first_index, u64 count, u64 *__restrict__ out, const Modulus *__restrict__ mods, u32 k,
                                                      u32 n) {
    const u64 per = n / 8, total = count * per;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        const u64 e = g / per, b = g % per;
        u64 r[8];
        chacha20_block(key, b, first_index + e, r);
        int v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = draw_ternary(r[i]);
        for (u32 p = 0; p < k; ++p) {
            const u64 q = mods[p].q;
            ulonglong2 *dst = (ulonglong2 *)(out + (e * k + p) * n + b * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ulonglong2 w;
                w.x = v[2 * i] < 0 ? q - 1 : (u64)v[2 * i];
                w.y = v[2 * i + 1] < 0 ? q - 1 : (u64)v[2 * i + 1];
                dst[i] = w;
            }
        }
    }
}

// out[e][j][p][s] = u_ntt[e][p][s] * pk_ntt[j][p][s]
__global__ __launch_bounds__(256) void k_enc_pk_mul(const u64 *__restrict__ u_ntt, const u64 *__restrict__ pk_ntt, u64 *__restrict__ out,
                                                    const Modulus *__restrict__ mods, u32 k, u32 n, u64 count) {
    const u64 rows = count * k;
    for (u64 rp = blockIdx.y; rp < rows; rp += gridDim.y) {
        const u64 e = rp / k;
        const u32 p = (u32)(rp % k);
        const Modulus m = mods[p];
        const u64 *u = u_ntt + rp * n, *p0 = pk_ntt + (u64)p * n, *p1 = pk_ntt + ((u64)k + p) * n;
        u64 *o0 = out + ((e * 2) * k + p) * n, *o1 = out + ((e * 2 + 1) * k + p) * n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const u64 x = u[i];
            o0[i] = mul_barrett(x, p0[i], m);
            o1[i] = mul_barrett(x, p1[i], m);
        }
    }
}

struct EncLift {                      // plaintext lifting of add_plain (fhe_hip.hip fhe_add_plain): Delta m' = delta m (+ q mod t for the upper half)
    u64 threshold;
    u64 delta[FHE_MAX_K], increment[FHE_MAX_K];
};
// one thread = eight consecutive coefficients of polynomial j of encryption e: its noise block, every residue
__global__ __launch_bounds__(256) void k_enc_finish(ChaChaKey key, u64 first_index, u64 count, u64 *__restrict__ ct, const u64 *__restrict__ plain,
                                                    const Modulus *__restrict__ mods, u32 k, u32 n, EncLift L) {
    const u64 per = n / 8, total = count * 2 * per;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        const u64 b = g % per, ej = g / per, e = ej / 2;
        const u32 j = (u32)(ej % 2);
        u64 r[8];
        chacha20_block(key, (u64)(1 + j) * per + b, first_index + e, r);
        int v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = draw_noise(r[i]);
        u64 m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool any = false;
        if (j == 0 && plain) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { m[i] = plain[e * n + b * 8 + i]; any |= m[i] != 0; }
        }
        for (u32 p = 0; p < k; ++p) {
            const Modulus md = mods[p];
            const u64 q = md.q;
            ulonglong2 *dst = (ulonglong2 *)(ct + (ej * k + p) * n + b * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ulonglong2 w = dst[i];
                u64 a[2] = {w.x, w.y};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nv = v[2 * i + h];
                    u64 x = addmod(a[h], nv < 0 ? q - (u64)(-nv) : (u64)nv, q);
                    if (any) {
                        const u64 mm = m[2 * i + h];
                        if (mm) {
                            u64 lift = mul_barrett(L.delta[p], mm % q, md);
                            if (mm >= L.threshold) lift = addmod(lift, L.increment[p], q);
                            x = addmod(x, lift, q);
                        }
                    }
                    a[h] = x;
                }
                w.x = a[0];
                w.y = a[1];
                dst[i] = w;
            }
        }
    }
}

// ---- ONE launch per batch (round 6) ----------------------------------------------------------------------------------------------
// One workgroup = one encryption, n / 16 threads, sixteen coefficients per thread in the transforms' pass-0 mapping (coefficient
// r * TP + tid).  The three draws of the encryption (u, e1, e2: 3 n / 8 ChaCha20 blocks, the SAME blocks of the SAME (key, index)
// stream the five-launch path reads, so the same ciphertext bits) are generated once, six blocks per thread in stream order, and
// exchanged through the transform buffer as bytes: every thread ends up with its sixteen draws of each kind packed into four
// registers (12 VGPRs, alive across the primes).  Then per prime: u's residues -> forward transform -> the two key products per
// slot -> two inverse transforms -> + e1 / e2 (+ Delta m' on c0) -> 2 n residues written.  The only memory traffic is the public
// key (k polynomial pairs, L2-resident across the batch), the plaintext coefficients and the ciphertext itself; u, its transform
// and the products never exist in memory (the five launches moved 3.6 MB per 0.5 MB ciphertext at n = 8192, k = 4).
// A = the arithmetic of the q-base: EncPm<L, C> (pseudo-Mersenne transforms, SEAL's 54 / 55-bit primes) or EncShoup<L> (lazy Shoup
// transforms, any base of primes up to 58 bits).
template <int L, typename C> struct EncPm {
    using Mod = PmMod;
    static constexpr int N = NttShape<L>::N;
    static __device__ __forceinline__ Mod mod(const RnsBase &b, u32 p) { return b.pm[p]; }
    static __device__ __forceinline__ u64 q(const Mod &m) { return m.q; }
    static __device__ __forceinline__ void fwd(u64 (&x)[1][16], const RnsBase &b, u32 p, const Mod &m, u64 *lds, int tid) {
        ntt_fwd_regs_pm<L, 1, 16, C::LIM, C::CS>(x, b.tw_pm + (size_t)p * N, m, lds, tid);
#pragma unroll
        for (int r = 0; r < 16; r++) x[0][r] = fold_pm(x[0][r], m);
    }
    static __device__ __forceinline__ u64 mul(u64 x, u64 w, const Mod &m, const Modulus &) { return mulvv_pm(x, w, m); }      // below RQ / 16 q
    static __device__ __forceinline__ void inv(u64 (&x)[2][16], const RnsBase &b, u32 p, const Mod &m, u64 *lds, int tid) {
        ntt_inv_regs_pm<L, 2, C::RQ, C::XB, C::LIM, C::RQ>(x, b.itw_pm + (size_t)p * N, m, lds, tid);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) x[j][r] = canon_rq_pm<C::RQ>(x[j][r], m);
    }
};
template <int L> struct EncShoup {
    struct Mod { NttMod m; float cs; };
    static constexpr int N = NttShape<L>::N;
    static __device__ __forceinline__ Mod mod(const RnsBase &b, u32 p) { Mod o; o.m = ntt_mod(b.mod[p].q); o.cs = canon_scale(o.m.q); return o; }
    static __device__ __forceinline__ u64 q(const Mod &m) { return m.m.q; }
    static __device__ __forceinline__ void fwd(u64 (&x)[1][16], const RnsBase &b, u32 p, const Mod &m, u64 *lds, int tid) {
        ntt_fwd_regs4<L, true>(x[0], b.tw + (size_t)p * N, m.m, lds, tid);        // below (2 + 4 L) q <= 58 q
#pragma unroll
        for (int r = 0; r < 16; r++) x[0][r] = canon_below_64q(x[0][r], m.m.q, m.cs);
    }
    static __device__ __forceinline__ u64 mul(u64 x, u64 w, const Mod &, const Modulus &md) { return mul_barrett(x, w, md); }  // canonical
    static __device__ __forceinline__ void inv(u64 (&x)[2][16], const RnsBase &b, u32 p, const Mod &m, u64 *lds, int tid) {
        ntt_inv_regs4m<L, 2, true>(x, b.itw + (size_t)p * N, m.m, lds, tid);      // [0, 4q) in and out
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) x[j][r] = csub(csub(x[j][r], 2 * m.m.q), m.m.q);
    }
};
template <int L> using EncPmA = EncPm<L, PmA>;
template <int L> using EncPmB = EncPm<L, PmB>;
__device__ __forceinline__ int enc_byte(const u32 (&w)[4], int r) { return (int)(w[r >> 2] << (24 - 8 * (r & 3))) >> 24; }      // sign-extended byte r

template <int L, typename A, int OCC>
__global__ __launch_bounds__(NttShape<L>::TP, OCC) void k_enc_fused(ChaChaKey key, u64 first_index, u64 *__restrict__ ct, const u64 *__restrict__ plain,
                                                                   const u64 *__restrict__ pk_ntt, RnsBase base, EncLift Lf) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid0 = threadIdx.x;
    const u64 e = blockIdx.x;
    const u32 k = base.count;
    // (1) the draws, in stream order: block g = role * (n / 8) + b of encryption e, eight draws as eight bytes in one LDS word
    constexpr int PER = N / 8;
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {
        const int g = tid0 + j * TP;                              // 6 TP = 3 PER blocks
        u64 r[8];
        chacha20_block(key, (u64)g, first_index + e, r);
        u64 packed = 0;
        if (g < PER) {
#pragma unroll
            for (int i = 0; i < 8; ++i) packed |= (u64)(unsigned char)draw_ternary(r[i]) << (8 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) packed |= (u64)(unsigned char)draw_noise(r[i]) << (8 * i);
        }
        lds[g] = packed;
    }
    __syncthreads();
    u32 du[4], d1[4], d2[4];                                     // byte r of each: the draw at coefficient r * TP + tid
    {
        const signed char *bytes = (const signed char *)lds;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            u32 a = 0, b = 0, c = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int coef = (4 * w + i) * TP + tid0;
                a |= (u32)(unsigned char)bytes[coef] << (8 * i);
                b |= (u32)(unsigned char)bytes[N + coef] << (8 * i);
                c |= (u32)(unsigned char)bytes[2 * N + coef] << (8 * i);
            }
            du[w] = a; d1[w] = b; d2[w] = c;
        }
    }
    __syncthreads();                                             // the buffer goes back to the transforms
    // (2) per prime
#pragma unroll 1
    for (u32 p = 0; p < k; ++p) {
        // the packed draws are loop-invariant: without this the compiler unpacks all 48 bytes ONCE, before the loop, and keeps them in
        // 48 registers across the transforms (800 B of scratch per lane); declared modified here, they are unpacked where they are used
#pragma unroll
        for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(du[w]), "+v"(d1[w]), "+v"(d2[w]));
        int tid = tid0;                                          // likewise the thread index: every LDS / global address of the transforms is a function of
        asm volatile("" : "+v"(tid));                            // it, and hoisted out of the loop they cost another 60 spilled registers
        const typename A::Mod m = A::mod(base, p);
        const Modulus md = base.mod[p];
        const u64 q = A::q(m);
        u64 x[1][16], y[2][16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int v = enc_byte(du, r);
            x[0][r] = v < 0 ? q - 1 : (u64)v;
        }
        A::fwd(x, base, p, m, lds, tid);                         // slot r of this thread: NTT-form word r * TP + tid
        const u64 *p0 = pk_ntt + (size_t)p * N + tid, *p1 = pk_ntt + ((size_t)k + p) * N + tid;
#pragma unroll
        for (int r = 0; r < 16; r++) {                           // both key products per slot; x dies here: the two inverse transforms run as a
            y[0][r] = A::mul(x[0][r], p0[r * TP], m, md);        // PAIR (one set of twiddle loads, the register budget of the pair kernels)
            y[1][r] = A::mul(x[0][r], p1[r * TP], m, md);
        }
        A::inv(y, base, p, m, lds, tid);                         // after a forward transform: no barrier needed (ntt_core.h, CONTRACT)
        u64 *c0 = ct + ((e * 2) * k + p) * N + tid, *c1 = ct + ((e * 2 + 1) * k + p) * N + tid;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int nv = enc_byte(d1, r);
            u64 v = addmod(y[0][r], nv < 0 ? q - (u64)(-nv) : (u64)nv, q);
            if (plain) {
                const u64 mm = plain[e * N + r * TP + tid];
                if (mm) {
                    u64 lift = mul_barrett(Lf.delta[p], mm % q, md);
                    if (mm >= Lf.threshold) lift = addmod(lift, Lf.increment[p], q);
                    v = addmod(v, lift, q);
                }
            }
            c0[r * TP] = v;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int nv = enc_byte(d2, r);
            c1[r * TP] = addmod(y[1][r], nv < 0 ? q - (u64)(-nv) : (u64)nv, q);
        }
        ntt_lds_release();
    }
}

__global__ __launch_bounds__(256) void k_enc_draws(ChaChaKey key, u64 first_index, u64 count, signed char *__restrict__ out, u32 n) {
    const u64 per = n / 8, total = count * 3 * per;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        const u64 b = g % (3 * per), e = g / (3 * per);               // b = block counter: role = b / per
        u64 r[8];
        chacha20_block(key, b, first_index + e, r);
        const bool tern = b < per;
        signed char *dst = out + e * 3 * n + b * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = (signed char)(tern ? draw_ternary(r[i]) : draw_noise(r[i]));
    }
}

// FractionalEncoder::encode (fhe_frac_encode, fhe_hip.hip) per value: coefficient d < int_coeffs = bit d of |whole| (negated for a negative
// value), coefficient n - i = bit i of the binary expansion of |frac| with flipped sign.  Bit i of |frac| is the low bit of
// trunc(|frac| 2^i): the scaling is exact, and beyond 2^53 the product is an even integer (the host loop's doubling and
// subtracting reaches the same digits: every step of it is exact)
__global__ __launch_bounds__(256) void k_frac_encode(const double *__restrict__ values, u64 count, int int_coeffs, int frac_coeffs, u64 t, u64 *__restrict__ out,
                                                     u32 n) {
    for (u64 e = blockIdx.y; e < count; e += gridDim.y) {
        const double value = values[e];
        const long long whole = (long long)value;
        const double frac = fabs(value - (double)whole);
        const u64 mag = whole < 0 ? (u64)(-whole) : (u64)whole;
        const bool negative = value < 0;
        for (u32 c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
            u64 coef = 0;
            if (c < (u32)int_coeffs && c < 64 && ((mag >> c) & 1)) coef = whole < 0 ? t - 1 : 1;
            const u32 i = n - c;                                       // digit of weight 2^-i
            if (frac != 0.0 && i >= 1 && i <= (u32)frac_coeffs && i <= 1074) {
                const double x = ldexp(frac, (int)i);                  // < 2^1074: finite
                const bool bit = x < 9007199254740992.0 ? (((u64)x) & 1) != 0 : false;
                if (bit) coef = negative ? 1 : t - 1;
            }
            out[e * n + c] = coef;
        }
    }
}


// ---- decryption (the clients' half: homo/client_jpeg.cpp:266-280, homo/client_resize.cpp:190-210) ----------------------------------
// phase = sum_j c_j s^j by Horner's rule per NTT slot: acc = c_{size-1}; acc = acc s + c_j
__global__ __launch_bounds__(256) void k_dec_horner(const u64 *__restrict__ ct_ntt, const u64 *__restrict__ sk_ntt, u64 *__restrict__ acc_out,
                                                    const Modulus *__restrict__ mods, u32 k, u32 n, u32 size, u64 count) {
    const u64 rows = count * k;
    for (u64 rp = blockIdx.y; rp < rows; rp += gridDim.y) {
        const u64 e = rp / k;
        const u32 p = (u32)(rp % k);
        const Modulus m = mods[p];
        const u64 *s = sk_ntt + (u64)p * n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const u64 sv = s[i];
            u64 acc = ct_ntt[((e * size + (size - 1)) * k + p) * n + i];
            for (int j = (int)size - 2; j >= 0; --j) acc = addmod(mul_barrett(acc, sv, m), ct_ntt[((e * size + j) * k + p) * n + i], m.q);
            acc_out[rp * n + i] = acc;
        }
    }
}

// m = floor((t x + floor(q/2)) / q) mod t for x = the CRT value of the phase residues, EXACTLY, without composing x:
//   y_i = phase_i (q/q_i)^-1 mod q_i,   x = sum_i y_i (q/q_i) - v q,   t y_i = a_i q_i + r_i
//   => t x + floor(q/2) = (sum_i a_i - t v) q + (sum_i r_i (q/q_i) + floor(q/2)),  so  m = (sum_i a_i + j) mod t  with
//   j = floor((sum_i r_i (q/q_i) + floor(q/2)) / q) in [0, k]  -- k multi-word comparisons against the multiples of q --
//   and the invariant noise |t x - (quotient) q| = |sum_i r_i (q/q_i) - j q| (its bit length goes to noise_bits by atomicMax).
// a_i comes from an exact division: t y_i - r_i is a multiple of the odd q_i and a_i < t < 2^64, so a_i = low64(t y_i - r_i) q_i^-1 mod 2^64.
#define DEC_K FHE_MAX_K
#define DEC_L (FHE_MAX_K + 1)
struct DecConsts {
    u64 t;
    u64 inv_punct[DEC_K], t_mod_q[DEC_K], qinv64[DEC_K];
    u64 punct[DEC_K][DEC_L];          // q / q_i
    u64 jq[DEC_K + 1][DEC_L];         // j q, j = 0 .. k
    u64 qhalf[DEC_L];                 // floor(q / 2)
};
template <int L>
__device__ __forceinline__ bool big_ge(const u64 (&a)[L], const u64 *b) {
    bool ge = true;                                                    // from the least significant word up: the last difference decides
#pragma unroll
    for (int l = 0; l < L; ++l)
        if (a[l] != b[l]) ge = a[l] > b[l];
    return ge;
}
// K = number of primes (compile-time: every multi-word value lives in registers), L = K + 1 words.  One atomicMax per WAVE: a wave's
// 64 consecutive coefficients belong to one ciphertext (n is a multiple of 64), the lanes' maxima are folded with shuffles first
// (8,192 atomics on one address per ciphertext made the first version of this kernel 25 x slower than it is now).
template <int K>
__global__ __launch_bounds__(256) void k_dec_round(const u64 *__restrict__ phase, u64 *__restrict__ plain, u32 *__restrict__ noise_bits,
                                                   const Modulus *__restrict__ mods, u32 n, u64 count, const DecConsts C) {
    constexpr int L = K + 1;
    const u64 total = count * n;
    for (u64 g0 = (u64)blockIdx.x * blockDim.x; g0 < total; g0 += (u64)gridDim.x * blockDim.x) {
        const u64 g = g0 + threadIdx.x;                               // total is a multiple of 256: no partial blocks
        const u64 e = g / n;
        const u32 c = (u32)(g % n);
        u64 N[L];
#pragma unroll
        for (int l = 0; l < L; ++l) N[l] = 0;
        u64 A = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const Modulus md = mods[i];
            const u64 y = mul_barrett(phase[(e * K + i) * n + c], C.inv_punct[i], md);
            const u64 r = mul_barrett(C.t_mod_q[i], y, md);
            const u64 a = (C.t * y - r) * C.qinv64[i];
            A += a;
            if (A >= C.t) A -= C.t;
            u64 carry = 0;
#pragma unroll
            for (int l = 0; l < L; ++l) {                              // N += r * punct_i
                const u64 lo = r * C.punct[i][l], hi = __umul64hi(r, C.punct[i][l]);
                const u64 s1 = N[l] + lo, c1 = s1 < lo;
                const u64 s2 = s1 + carry, c2 = s2 < carry;
                N[l] = s2;
                carry = hi + c1 + c2;
            }
        }
        u64 M[L];                                                      // N + floor(q/2)
        u64 cy = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const u64 s1 = N[l] + C.qhalf[l], c1 = s1 < N[l];
            const u64 s2 = s1 + cy, c2 = s2 < cy;
            M[l] = s2;
            cy = c1 + c2;
        }
        u32 j = 0;
#pragma unroll
        for (int m = 1; m <= K; ++m) j += big_ge<L>(M, C.jq[m]) ? 1 : 0;
        u64 v = A + j;
        while (v >= C.t) v -= C.t;
        plain[g] = v;
        if (noise_bits) {
            u64 J[L];                                                  // j q, selected without indexing the argument block by a lane value
#pragma unroll
            for (int l = 0; l < L; ++l) J[l] = 0;
#pragma unroll
            for (int m = 1; m <= K; ++m)
                if (j == (u32)m) {
#pragma unroll
                    for (int l = 0; l < L; ++l) J[l] = C.jq[m][l];
                }
            const bool pos = big_ge<L>(N, J);
            u64 borrow = 0;
            int bits = 0;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const u64 hi_ = pos ? N[l] : J[l], lo_ = pos ? J[l] : N[l];
                const u64 d1 = hi_ - lo_, b1 = hi_ < lo_;
                const u64 d2 = d1 - borrow, b2 = d1 < borrow;
                borrow = b1 + b2;
                if (d2) bits = 64 * l + (64 - __clzll((long long)d2));
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
            if ((threadIdx.x & 63) == 0) atomicMax(noise_bits + e, (u32)bits);
        }
    }
}

ChaChaKey load_key(const uint8_t key[32]) {
    ChaChaKey k;
    for (int i = 0; i < 8; ++i) k.w[i] = (u32)key[4 * i] | ((u32)key[4 * i + 1] << 8) | ((u32)key[4 * i + 2] << 16) | ((u32)key[4 * i + 3] << 24);
    return k;
}
unsigned blocks_for(u64 threads) {
    const u64 b = (threads + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}
}  // namespace

extern "C" void fhe_noise_cdt(uint64_t out[FHE_NOISE_CDT_LEN]) {
    for (int i = 0; i < FHE_NOISE_CDT_LEN; ++i) out[i] = kNoiseCdtHost[i];
}

extern "C" int fhe_frac_encode_batch(const fhe_ctx *c, const double *values, uint64_t count, int int_coeffs, int frac_coeffs, uint64_t *d_plain, fhe_stream s) {
    if (!c || (!values && count) || (!d_plain && count)) return fail(FHE_ERR_PARAM, "null argument");
    if (int_coeffs < 0 || frac_coeffs < 0 || (uint32_t)(int_coeffs + frac_coeffs) > c->n) return fail(FHE_ERR_PARAM, "encoder coefficient counts do not fit the polynomial");
    for (u64 i = 0; i < count; ++i) {                                  // the refusals of fhe_frac_encode, before anything is launched
        const double v = values[i];
        if (!std::isfinite(v) || std::fabs(v) >= 9.0e18) return fail(FHE_ERR_PARAM, "value %llu out of range", (unsigned long long)i);
        const long long whole = (long long)v;
        const u64 mag = whole < 0 ? (u64)(-whole) : (u64)whole;
        if (int_coeffs < 64 && (mag >> int_coeffs)) return fail(FHE_ERR_PARAM, "integer part of value %llu needs more than %d coefficients", (unsigned long long)i, int_coeffs);
    }
    hipStream_t st = (hipStream_t)s;
    const u64 per_slot = FHE_STAGE_SLOT_BYTES / sizeof(double);
    for (u64 done = 0; done < count; done += per_slot) {
        const u64 part = count - done < per_slot ? count - done : per_slot;
        FheStage sg;
        int rc = fhe_stage_acquire(values + done, part * sizeof(double), st, &sg);
        if (rc) return rc;
        dim3 grid((c->n + 255) / 256, (unsigned)(part < 32768 ? part : 32768));
        k_frac_encode<<<grid, 256, 0, st>>>((const double *)sg.dev, part, int_coeffs, frac_coeffs, c->t, (u64 *)d_plain + done * c->n, c->n);
        const hipError_t le = hipGetLastError();
        rc = fhe_stage_release(sg, st);
        if (le != hipSuccess) return fail(FHE_ERR_HIP, "kernel launch: %s", hipGetErrorString(le));
        if (rc) return rc;
    }
    return FHE_OK;
}

extern "C" size_t fhe_encrypt_scratch_bytes(const fhe_ctx *c, uint64_t count) { return c ? (size_t)count * c->k * c->n * sizeof(u64) : 0; }

extern "C" int fhe_encrypt_batch(const fhe_ctx *c, const uint64_t *d_pk_ntt, const uint64_t *d_plain, uint64_t count, const uint8_t key[32], uint64_t first_index,
                                 uint64_t *d_out, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!c || !d_pk_ntt || !key || (!d_out && count)) return fail(FHE_ERR_PARAM, "null argument");
    if (!count) return FHE_OK;
    if (!scratch || scratch_bytes < fhe_encrypt_scratch_bytes(c, count)) return fail(FHE_ERR_PARAM, "scratch too small: need fhe_encrypt_scratch_bytes()");
    if (first_index + count < first_index) return fail(FHE_ERR_PARAM, "encryption index wraps: a (key, index) pair would repeat");
    if (c->n % 8) return fail(FHE_ERR_PARAM, "n must be a multiple of 8");
    hipStream_t st = (hipStream_t)s;
    const ChaChaKey k = load_key(key);
    EncLift lift;
    lift.threshold = c->upper_half_threshold;
    for (u32 i = 0; i < FHE_MAX_K; ++i) { lift.delta[i] = i < c->k ? c->delta_mod[i] : 0; lift.increment[i] = i < c->k ? c->upper_half_increment[i] : 0; }
    // one launch (k_enc_fused) where the base has register transforms this file instantiates: pseudo-Mersenne bases, and any other base of
    // primes up to 58 bits on the lazy Shoup passes; FHE_ENC_UNFUSED=1 (and 59..61-bit primes) keeps the five launches below -- same bits
    bool shoup_ok = c->max_prime_bits <= 58;                      // EncShoup: lazy passes (<= 58 bits) and the float quotient estimate of canon_below_64q (>= 2^33)
    for (u32 i = 0; i < c->k; ++i) shoup_ok = shoup_ok && (c->qb.primes[i] >> 33);
    if (!c->opt.enc_unfused && count <= 0x7fffffffULL && (shoup_ok || (c->qb.pm_class && !c->opt.ntt_nopm))) {
        const RnsBase base = c->qb.dev();
        // OCC = waves per SIMD asked of the register allocator: 2 (256 VGPRs, no scratch to speak of, one workgroup per CU at n = 8192; the default:
        // 0.67 against 0.77 us per ciphertext at P8192, 0.38 against 0.50 at P4096) or 4 (128 VGPRs, 300-500 bytes of scratch per lane, two workgroups per
        // CU; FHE_ENC_OCC=4) -- profiles/EXPERIMENTS.md section 13
#define GO_ENC(AA, OCC) DISPATCH_L(c->logn, (k_enc_fused<L, AA<L>, OCC><<<(unsigned)count, NttShape<L>::TP, 0, st>>>(k, first_index, (u64 *)d_out, (const u64 *)d_plain, (const u64 *)d_pk_ntt, base, lift)))
#define GO_ENC2(AA) do { if (c->opt.enc_occ4) { GO_ENC(AA, 4); } else { GO_ENC(AA, 2); } } while (0)
        if (c->qb.pm_class == 1 && !c->opt.ntt_nopm) { GO_ENC2(EncPmA); }
        else if (c->qb.pm_class == 2 && !c->opt.ntt_nopm) { GO_ENC2(EncPmB); }
        else { GO_ENC2(EncShoup); }
#undef GO_ENC2
#undef GO_ENC
        KERNEL_CHECK();
        return FHE_OK;
    }
    u64 *u = (u64 *)scratch;
    k_enc_sample_u<<<blocks_for(count * (c->n / 8)), 256, 0, st>>>(k, first_index, count, u, c->qb.d_mod, c->k, c->n);
    KERNEL_CHECK();
    int rc = fhe_ntt_forward(c, (const uint64_t *)u, (uint64_t *)u, count, s);
    if (rc) return rc;
    {
        const u64 rows = count * c->k;
        dim3 grid((c->n + 255) / 256, (unsigned)(rows < 32768 ? rows : 32768));
        k_enc_pk_mul<<<grid, 256, 0, st>>>(u, (const u64 *)d_pk_ntt, (u64 *)d_out, c->qb.d_mod, c->k, c->n, count);
        KERNEL_CHECK();
    }
    if ((rc = fhe_ntt_inverse(c, d_out, d_out, count * 2, s))) return rc;
    k_enc_finish<<<blocks_for(count * 2 * (c->n / 8)), 256, 0, st>>>(k, first_index, count, (u64 *)d_out, (const u64 *)d_plain, c->qb.d_mod, c->k, c->n, lift);
    KERNEL_CHECK();
    return FHE_OK;
}

extern "C" int fhe_encrypt_draws(const fhe_ctx *c, const uint8_t key[32], uint64_t first_index, uint64_t count, int8_t *d_draws, fhe_stream s) {
    if (!c || !key || (!d_draws && count)) return fail(FHE_ERR_PARAM, "null argument");
    if (!count) return FHE_OK;
    if (c->n % 8) return fail(FHE_ERR_PARAM, "n must be a multiple of 8");
    k_enc_draws<<<blocks_for(count * 3 * (c->n / 8)), 256, 0, (hipStream_t)s>>>(load_key(key), first_index, count, (signed char *)d_draws, c->n);
    KERNEL_CHECK();
    return FHE_OK;
}

// ---- decryption ---------------------------------------------------------------------------------------------------------------
extern "C" uint32_t fhe_ctx_modulus_bits(const fhe_ctx *c) {
    if (!c) return 0;
    hostmath::BigUInt Q(1);
    for (u32 i = 0; i < c->k; ++i) Q.mul_small(c->qb.primes[i]);
    return (uint32_t)Q.bits();
}
extern "C" size_t fhe_decrypt_scratch_bytes(const fhe_ctx *c, uint32_t size, uint64_t count) {
    return c ? (size_t)count * ((size_t)size + 1) * c->k * c->n * sizeof(u64) : 0;
}
extern "C" int fhe_decrypt_batch(const fhe_ctx *c, const uint64_t *d_sk_ntt, const uint64_t *d_ct, uint32_t size, uint64_t count, uint64_t *d_plain,
                                 uint32_t *d_noise_bits, void *scratch, size_t scratch_bytes, fhe_stream s) {
    using namespace hostmath;
    if (!c || !d_sk_ntt || (!d_ct && count) || (!d_plain && count)) return fail(FHE_ERR_PARAM, "null argument");
    if (size < 2) return fail(FHE_ERR_PARAM, "a ciphertext has at least two polynomials");
    if (!count) return FHE_OK;
    if (!scratch || scratch_bytes < fhe_decrypt_scratch_bytes(c, size, count)) return fail(FHE_ERR_PARAM, "scratch too small: need fhe_decrypt_scratch_bytes()");
    if (c->t >> 60) return fail(FHE_ERR_PARAM, "plain modulus too large for the exact rounding kernel");
    // everything the last kernel (k_dec_round: full 256-thread workgroups, full-wave shuffle folds) needs, before anything is enqueued
    if (c->n % 256) return fail(FHE_ERR_PARAM, "n must be a multiple of 256");
    if (c->k < 1 || c->k > 8) return fail(FHE_ERR_PARAM, "unsupported number of primes");
    hipStream_t st = (hipStream_t)s;
    const u32 k = c->k, n = c->n;
    DecConsts C;
    std::memset(&C, 0, sizeof C);
    C.t = c->t;
    BigUInt Q(1, DEC_L + 1);
    for (u32 i = 0; i < k; ++i) Q.mul_small(c->qb.primes[i]);
    for (u32 i = 0; i < k; ++i) {
        const u64 qi = c->qb.primes[i];
        BigUInt P(1, DEC_L + 1);
        for (u32 j = 0; j < k; ++j)
            if (j != i) P.mul_small(c->qb.primes[j]);
        for (u32 l = 0; l < DEC_L; ++l) C.punct[i][l] = P.w[l];
        C.inv_punct[i] = invmod(P.mod_small(qi), qi);
        C.t_mod_q[i] = c->t % qi;
        u64 inv = qi;                                               // Newton: q^-1 mod 2^64 for odd q (5 steps double the correct low bits from 3)
        for (int it = 0; it < 6; ++it) inv *= 2 - qi * inv;
        C.qinv64[i] = inv;
    }
    BigUInt J(0, DEC_L + 1);
    for (u32 j = 0; j <= k; ++j) {
        for (u32 l = 0; l < DEC_L; ++l) C.jq[j][l] = J.w[l];
        J.add(Q);
    }
    BigUInt H = Q;
    H.shr1();
    for (u32 l = 0; l < DEC_L; ++l) C.qhalf[l] = H.w[l];
    u64 *ntt = (u64 *)scratch, *acc = ntt + (size_t)count * size * k * n;
    int rc = fhe_ntt_forward(c, d_ct, (uint64_t *)ntt, count * size, s);
    if (rc) return rc;
    {
        const u64 rows = count * k;
        dim3 grid((n + 255) / 256, (unsigned)(rows < 32768 ? rows : 32768));
        k_dec_horner<<<grid, 256, 0, st>>>(ntt, (const u64 *)d_sk_ntt, acc, c->qb.d_mod, k, n, size, count);
        KERNEL_CHECK();
    }
    if ((rc = fhe_ntt_inverse(c, (const uint64_t *)acc, (uint64_t *)acc, count, s))) return rc;
    if (d_noise_bits) HIP_TRY(hipMemsetAsync(d_noise_bits, 0, count * sizeof(u32), st));
    const unsigned blocks = blocks_for(count * n);
    switch (k) {
#define GO(KK) case KK: k_dec_round<KK><<<blocks, 256, 0, st>>>(acc, (u64 *)d_plain, d_noise_bits, c->qb.d_mod, n, count, C); break;
        GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
#undef GO
        default: return fail(FHE_ERR_PARAM, "unsupported number of primes");
    }
    KERNEL_CHECK();
    return FHE_OK;
}
