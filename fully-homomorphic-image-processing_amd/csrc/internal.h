// internal.h -- shared declarations of the library's translation units (not part of the ABI).
#pragma once
#include "../../include/fhe_hip.h"

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "modarith.h"
#include "ntt_core.h"

int fhe_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail fhe_fail
#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(FHE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
#define KERNEL_CHECK()                                                                            \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(FHE_ERR_HIP, "kernel launch: %s", hipGetErrorString(e_)); \
    } while (0)

struct BaseTables {
    std::vector<u64> primes;
    ulonglong2 *d_tw = nullptr, *d_itw = nullptr;
    Modulus *d_mod = nullptr;
    std::vector<Modulus> h_mod;
    // exact-FP64 companion tables (only when every prime is below 2^48): centred twiddles as doubles
    double *d_tw_f64 = nullptr, *d_itw_f64 = nullptr;   // [count][n]
    // pseudo-Mersenne tables (ntt_core.h): (w, w 2^31 mod q) pairs; pm_class 0 = the base does not qualify
    ulonglong2 *d_tw_pm = nullptr, *d_itw_pm = nullptr;
    ulonglong2 *d_tw_pm3 = nullptr, *d_itw_pm3 = nullptr;   // the same pairs ordered for passes of three stages (dct_u64.hip), class 1 bases only
    PmMod *d_pm = nullptr;
    int pm_class = 0;
    RnsBase dev() const { return RnsBase{d_tw, d_itw, d_mod, (u32)primes.size(), d_tw_pm, d_itw_pm, d_pm}; }
};

// Experiment switches.  Read from the environment ONCE, in fhe_ctx_create, and fixed for the life of the context: no
// launch path calls getenv, two contexts of one process may differ, and a concurrent setenv cannot race a launch.
// Every default is the measured-best path; the alternatives stay for A/B measurements and for the parity tests that
// run the fallback kernels (tests/test_gpu_parity.py creates a second context with the variable set).
struct FheOptions {
    bool force_u64 = false;          // FHE_DCT_FORCE_U64=1: the u64 kernels where the exact-FP64 ones would run (NTT, multiply_plain, DCT, rgb_to_ycc)
    bool dct_pipeline = false;       // FHE_DCT_PIPELINE=1: column kernel of wave w on a second stream beside the row kernel of wave w + 1
    u64 dct_wave_blocks = 256;       // FHE_DCT_WAVE_BLOCKS: blocks per wave of the fused DCT pair (size of the intermediate)
    int dct_le = 3;                  // FHE_DCT_LE=4: 16 coefficients per thread in the fused FP64 pair where 8 is the default
    bool dct_pack = true;            // FHE_DCT_PACK=0: FP64 intermediate instead of the packed one (primes <= 37 bits)
    bool dct_ldsc = true;            // FHE_DCT_LDSC=0: row-kernel constants through registers instead of LDS
    bool dct_u64_fused = true;       // FHE_DCT_U64_FUSED=0: three-launch general path instead of the fused u64 pair
    u32 dct_one_launch = 0;          // FHE_DCT_ONE_LAUNCH=D (experiment): rows and columns of the FP64 pair in one launch, columns D units behind the rows
    bool ntt_nolazy = false;         // FHE_NTT_NOLAZY=1: Harvey butterflies with conditional subtractions everywhere
    bool ntt_single = false;         // FHE_NTT_SINGLE=1: one polynomial per workgroup at n >= 8192 as well
    bool ntt_nopm = false;           // FHE_NTT_NOPM=1: Shoup butterflies where the pseudo-Mersenne ones would run
    bool behz_aux61 = false;         // FHE_BEHZ_AUX61=1: 61-bit auxiliary base (SEAL 2.3's size) where 58 bits suffice
    bool behz_chunk3 = false;        // FHE_BEHZ_CHUNK3=1: base conversions reduce every three terms (the 61-bit schedule)
    bool behz_tensor_canon = false;  // FHE_BEHZ_TENSOR_CANON=1: tensor step with canonical Barrett products and modular additions
    bool behz_tensor_single = false; // FHE_BEHZ_TENSOR_SINGLE=1: tensor + inverse transform one polynomial per workgroup
    bool behz_square_full = false;   // FHE_BEHZ_SQUARE_FULL=1: squares form a_i a_j and a_j a_i separately (the general tensor kernel)
    bool plain_sum_unfused = false;  // FHE_PLAIN_SUM_UNFUSED=1: the Taylor / harmonic sums as separate multiply_plain calls and additions (before k_mulplain_sum_pm)
    bool cubic_unfused = false;      // FHE_CUBIC_UNFUSED=1: Cubic's three products as three complete fhe_multiply calls + k_cubic_combine_g (before round 4's fused tail)
    bool enc_unfused = false;        // FHE_ENC_UNFUSED=1: fhe_encrypt_batch as the five launches of round 5 (k_enc_sample_u, transforms, k_enc_pk_mul, k_enc_finish) instead of k_enc_fused
    bool enc_occ4 = false;           // FHE_ENC_OCC=4: k_enc_fused built for four waves per SIMD (128 VGPRs, scratch) instead of two
    bool relin_steps = false;        // FHE_RELIN_STEPS=1: fhe_relinearize_n runs its key switches one after the other (round 6: the default folds them into one pass, behz.hip relin_pm)
    bool relin_fused = false;        // FHE_RELIN_FUSED=1 (experiment, round 5): key-switch accumulation inside the inverse-transform kernel (k_relin_accum_inv_add_pm: one launch and
                                     // 4 MB of traffic per relinearisation less, same bits, 1-3 % SLOWER at dbc = 30 -- the digits are read twice and the kernels are issue-bound; profiles/EXPERIMENTS.md)
    bool behz_fused_prepare = false; // FHE_BEHZ_FUSED_PREPARE=1: base extension fused into the forward transforms (k_behz_prepare_pm: 25 % less HBM traffic per
                                     // product, 5 % slower -- the y_i are recomputed per auxiliary prime and the kernels are issue-bound; profiles/EXPERIMENTS.md)
};

struct fhe_ctx {
    u32 n = 0, logn = 0, k = 0;
    FheOptions opt;
    u64 t = 0;
    int device = 0;
    int max_prime_bits = 0;
    BaseTables qb;     // q-base
    // plaintext lifting (SEAL 2.3 multiply_plain / preencrypt semantics)
    u64 upper_half_threshold = 0;
    u64 plain_upper_half_increment[FHE_MAX_K] = {0};   // (q - t) mod q_i
    u64 delta_mod[FHE_MAX_K] = {0};                    // floor(q/t) mod q_i
    u64 upper_half_increment[FHE_MAX_K] = {0};         // (q mod t) mod q_i
    // ct x ct tables (behz.hip): auxiliary base, its twiddles, base-conversion constants.  Built by the FIRST entry point that
    // needs them (fhe_behz_ensure, std::call_once): a context that only runs the linear circuits (DCT, colour conversion,
    // add / multiply_plain) never searches for auxiliary primes and cannot fail on them.  After the once-flag has fired the
    // pointer never changes again, so the context is as immutable for its users as before.
    struct BehzTables *behz = nullptr;
    mutable std::once_flag behz_once;
    int behz_rc = 0;
    std::string behz_err;
    // second stream + events for overlapping the column kernel of one wave of blocks with the row
    // kernel of the next (fhe_dct8x8_quant); created with the context
    u32 *d_arrived = nullptr;        // one-launch experiment: [0] = error flag, [1 ...] per-unit arrival counters
    u64 arrived_cap = 0;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_cols[2] = {nullptr, nullptr};
    // rgb_to_ycc_fhe constants (nine encoded factors + Delta*encode(128)), built on first use of each
    // (int_coeffs, frac_coeffs) pair under rgb_mutex and kept until the context is destroyed (entries are
    // never freed or moved while the context lives, so a pointer handed to an in-flight launch stays valid):
    // the reference re-encodes them on every call
    struct RgbConsts {
        int int_coeffs = -1, frac_coeffs = -1;
        ulonglong2 *d_c = nullptr;      // [9][k][n] Shoup pairs, u64 kernels' slot order
        double *d_c_f64 = nullptr;      // [9][k][n] centred doubles, fused kernels' slot order (or null)
        u64 *d_off = nullptr;           // [k][off_len]
        u32 off_len = 0;
    };
    mutable std::vector<RgbConsts *> rgb;
    mutable std::mutex rgb_mutex;
};

#define DCT_NCONST 76
struct fhe_dct_plan {
    ulonglong2 *d_consts = nullptr;   // [DCT_NCONST][k][n] Shoup pairs, slot order
    double *d_consts_f64 = nullptr;   // [DCT_NCONST][k][n] centred doubles (FP64 path), or null
    ulonglong2 *d_consts_le3 = nullptr;   // [DCT_NCONST][k][n] Shoup pairs in the fused u64 kernels' order (dct_u64.hip), or null
    u32 k = 0, n = 0;
    bool has_quant = false;
};

#define DISPATCH_L(logn, ...)                                                    \
    switch (logn) {                                                              \
        case 10: { constexpr int L = 10; __VA_ARGS__; } break;                   \
        case 11: { constexpr int L = 11; __VA_ARGS__; } break;                   \
        case 12: { constexpr int L = 12; __VA_ARGS__; } break;                   \
        case 13: { constexpr int L = 13; __VA_ARGS__; } break;                   \
        case 14: { constexpr int L = 14; __VA_ARGS__; } break;                   \
        default: return fail(FHE_ERR_PARAM, "unsupported log2(n)=%u", logn);     \
    }

// which ciphertext of a batch pair / output `c` refers to: an explicit index array, the periodic map (off + c / div) % cnt, or c itself
// (the circuits' operands are taps into resident pixels, one offset per output column, one sine polynomial per harmonic, ...)
struct CMap {
    const u32 *idx;
    u64 div, cnt, off;
    __device__ __forceinline__ u64 operator()(u64 c) const { return idx ? idx[c] : (cnt ? ((off + c / div) % cnt) : c); }
};

// terms of fhe_multiply_plain_sum (fhe_hip.hip): src_i [count][size_i][k][n] -- OVERWRITTEN (left in NTT form) --, plain_i in NTT form
// (fhe_plain_prepare)
#define FHE_PLAIN_SUM_MAX_TERMS 16
struct PlainSumTerms {
    u32 count;
    u32 size[FHE_PLAIN_SUM_MAX_TERMS];
    u64 *src[FHE_PLAIN_SUM_MAX_TERMS];
    const ulonglong2 *plain[FHE_PLAIN_SUM_MAX_TERMS];
};
bool fhe_multiply_plain_sum_supported(const fhe_ctx *c);
int fhe_multiply_plain_sum(const fhe_ctx *c, const PlainSumTerms &T, const u64 *addend, CMap amap, u32 addend_size, u64 *out, u32 out_size, u64 count,
                           hipStream_t st);

// cross-TU internals
// small host -> device hand-overs through the per-device ring of page-locked slots (fhe_hip.hip): acquire copies `bytes` (at most one
// slot) and enqueues the upload on `st`; the caller launches the kernel that reads `dev` on the same stream, then calls release
#define FHE_STAGE_SLOT_BYTES ((size_t)256 << 10)
struct FheStage { void *dev; int slot; void *ring; };
int fhe_stage_acquire(const void *host, size_t bytes, hipStream_t st, FheStage *out);
int fhe_stage_release(const FheStage &sg, hipStream_t st);
int fhe_ntt_launch(bool inverse, const fhe_ctx *c, const BaseTables &B, const u64 *in, u64 *out, u64 n_res_polys, hipStream_t st);
int fhe_build_base(BaseTables &B, const std::vector<u64> &primes, u32 n, u32 logn, bool want_f64);
void fhe_free_base(BaseTables &B);
int fhe_behz_build(fhe_ctx *c);   // called once per context, through fhe_behz_ensure
int fhe_behz_ensure(const fhe_ctx *c);
// Cubic's three products with their tail fused into the floor / back conversion (behz.hip: k_behz_floor3_combine_pm).
// fhe_behz_tensor_shared = fhe_multiply_prepared_shared up to and including the inverse transforms: D_q [count][so][k][n] and
// D_b [count][so][k+1][n] are left in `d` (fhe_behz_d_words); `scratch` holds the prepared form of `a` (fhe_multiply_operand_words).
bool fhe_behz_floor3_supported(const fhe_ctx *c);
size_t fhe_behz_d_words(const fhe_ctx *c, u32 so, u64 count);
int fhe_behz_tensor_shared(const fhe_ctx *c, const u64 *a, u32 sa, const u64 *bp, u32 sb, u64 b_count, u64 b_div, u64 b_first, u64 *d, u64 count,
                           u64 *scratch, hipStream_t st);
int fhe_behz_floor3_combine(const fhe_ctx *c, const u64 *da, const u64 *db, const u64 *dc, u32 size_ab, u32 size_c, const u64 *B, CMap mB, u32 size_b,
                            u64 *out, CMap mo, u64 count, hipStream_t st);   // thread-safe: builds the ct x ct tables on first use; later calls cost one atomic load
void fhe_behz_free(fhe_ctx *c);
// fused FP64 DCT path (dct_fused.hip)
bool fhe_dct_f64_supported(const fhe_ctx *c);
// which: bit 0 = row kernel, bit 1 = column kernel
int fhe_dct_f64_launch(const fhe_ctx *c, const fhe_dct_plan *plan, const u64 *in, u64 *out, u64 n_blocks, double *mid, hipStream_t st, int which = 3);
int fhe_dct_f64_make_consts(const fhe_ctx *c, fhe_dct_plan *plan, hipStream_t st);
// fused u64 (Shoup) DCT path for primes of 48..57 bits (dct_u64.hip)
bool fhe_dct_u64_supported(const fhe_ctx *c);
int fhe_dct_u64_make_consts(const fhe_ctx *c, fhe_dct_plan *plan, hipStream_t st);
int fhe_dct_u64_launch(const fhe_ctx *c, const fhe_dct_plan *plan, const u64 *in, u64 *out, u64 n_blocks, u64 *mid, hipStream_t st);
bool fhe_rgb_f64_supported(const fhe_ctx *c);
int fhe_poly_f64_launch(int mode, const fhe_ctx *c, const u64 *in, u64 *out, u64 n_polys, const ulonglong2 *plain, hipStream_t st);
int fhe_rgb_f64_make_consts(const fhe_ctx *c, const ulonglong2 *d_c, double **out, hipStream_t st);
int fhe_rgb_f64_launch(const fhe_ctx *c, u64 *r, u64 *g, u64 *b, u64 count, const double *consts, const u64 *yoff, u32 yoff_len, hipStream_t st, u32 group = 0, u64 gstride = 0);
