// fhe_hip.hip -- libfhe_hip.so: HIP kernels (gfx950) + the C ABI of include/fhe_hip.h.
//
// Product code.  Never includes, links or calls anything under oracle/.
#include "internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "host_math.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int fhe_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char *fhe_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t fhe_abi_version(void) { return FHE_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
int fhe_build_base(BaseTables &B, const std::vector<u64> &primes, u32 n, u32 logn, bool want_f64) {
    using namespace hostmath;
    const size_t cnt = primes.size();
    B.primes = primes;
    std::vector<ulonglong2> tw(cnt * n), itw(cnt * n);
    B.h_mod.resize(cnt);
    for (size_t i = 0; i < cnt; ++i) {
        const u64 q = primes[i];
        const u64 psi = primitive_2n_root(q, n);
        if (!psi) return fail(FHE_ERR_PARAM, "no primitive 2n-th root modulo %llu", (unsigned long long)q);
        const u64 ipsi = invmod(psi, q), ninv = invmod(n % q, q);
        u64 p = 1, ip = 1;
        for (u32 j = 0; j < n; ++j) {
            const u32 r = bit_reverse(j, (int)logn);
            tw[i * n + r] = make_ulonglong2(p, shoup(p, q));
            itw[i * n + r] = make_ulonglong2(ip, shoup(ip, q));
            p = mulmod(p, psi, q);
            ip = mulmod(ip, ipsi, q);
        }
        // entry 0 is never indexed by a butterfly: it carries n^-1; the last inverse stage
        // (twiddle index 1) is pre-multiplied by n^-1 so the scaling costs nothing extra.
        itw[i * n + 0] = make_ulonglong2(ninv, shoup(ninv, q));
        const u64 w1 = mulmod(itw[i * n + 1].x, ninv, q);
        itw[i * n + 1] = make_ulonglong2(w1, shoup(w1, q));
        Modulus m;
        m.q = q;
        const int b = bit_length(q);
        m.mu = (u64)((((u128)1) << (2 * b)) / q);
        m.s1 = (u32)(b - 1);
        m.s2 = (u32)(b + 1);
        B.h_mod[i] = m;
    }
    HIP_TRY(hipMalloc(&B.d_tw, sizeof(ulonglong2) * cnt * n));
    HIP_TRY(hipMalloc(&B.d_itw, sizeof(ulonglong2) * cnt * n));
    HIP_TRY(hipMalloc(&B.d_mod, sizeof(Modulus) * cnt));
    HIP_TRY(hipMemcpy(B.d_tw, tw.data(), sizeof(ulonglong2) * cnt * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B.d_itw, itw.data(), sizeof(ulonglong2) * cnt * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B.d_mod, B.h_mod.data(), sizeof(Modulus) * cnt, hipMemcpyHostToDevice));
    // pseudo-Mersenne class of the base (ntt_core.h): q = 2^b - delta, product bound 2^b + 2^32 delta, fold bound 2^b + 2^(64-b) delta
    bool a_all = true, b_all = true;
    std::vector<PmMod> pm(cnt);
    for (size_t i = 0; i < cnt; ++i) {
        const u64 q = primes[i];
        const int b = bit_length(q);
        if (b < 34 || b > 58) { a_all = b_all = false; break; }
        const u64 delta = (1ULL << b) - q;
        const u128 prod = ((u128)1 << b) + ((u128)delta << 32), folded = ((u128)1 << b) + ((u128)delta << (64 - b));
        if ((delta >> 31) || folded * 16 > (u128)q * PM_FOLDED) { a_all = b_all = false; break; }
        const u128 vv = ((u128)1 << b) + ((u128)delta << (85 - b));       // mulvv_pm's result bound
        a_all = a_all && b <= 55 && b >= 52 && prod * 16 <= (u128)q * PmA::RQ && vv * 16 <= (u128)q * PmA::RQ;
        b_all = b_all && b >= 52 && prod * 16 <= (u128)q * PmB::RQ && vv * 16 <= (u128)q * PmB::RQ;
        pm[i].q = q;
        pm[i].delta = (u32)delta;
        pm[i].sh = (u32)(b - 32);
        pm[i].mb = (1u << (b - 32)) - 1;
        pm[i].pad = 0;
    }
    const int cls = a_all ? 1 : b_all ? 2 : 0;
    B.pm_class = cls;
    if (cls) {
        std::vector<ulonglong2> twp(cnt * n), itwp(cnt * n);
        for (size_t i = 0; i < cnt; ++i) {
            const u64 q = primes[i];
            for (u32 j = 0; j < n; ++j) {
                const u32 jp = pm_tw_index((int)logn, j);
                twp[i * n + jp] = make_ulonglong2(tw[i * n + j].x, (u64)(((u128)tw[i * n + j].x << 31) % q));
                itwp[i * n + jp] = make_ulonglong2(itw[i * n + j].x, (u64)(((u128)itw[i * n + j].x << 31) % q));
            }
        }
        HIP_TRY(hipMalloc(&B.d_tw_pm, sizeof(ulonglong2) * cnt * n));
        HIP_TRY(hipMalloc(&B.d_itw_pm, sizeof(ulonglong2) * cnt * n));
        HIP_TRY(hipMalloc(&B.d_pm, sizeof(PmMod) * cnt));
        HIP_TRY(hipMemcpy(B.d_tw_pm, twp.data(), sizeof(ulonglong2) * cnt * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(B.d_itw_pm, itwp.data(), sizeof(ulonglong2) * cnt * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(B.d_pm, pm.data(), sizeof(PmMod) * cnt, hipMemcpyHostToDevice));
        if (cls == 1 && logn >= 11 && logn <= 13) {
            // the fused u64 DCT kernels (dct_u64.hip) hold 8 coefficients per thread: passes of three stages, so the [i][th]
            // order inside a stage differs from the one above
            auto p3 = [logn](u32 idx) -> u32 {
                if (idx < 2) return idx;
                int sigma = 0;
                while ((2u << sigma) <= idx) sigma++;
                const int LE = 3, P = sigma / LE, lo = ((int)logn - LE * P - LE) < 0 ? 0 : ((int)logn - LE * P - LE), rb = ((int)logn - 1 - sigma) - lo;
                const u32 off = idx - (1u << sigma), cnt_i = 1u << (LE - 1 - rb), th = off / cnt_i, i = off % cnt_i;
                return (1u << sigma) + i * ((1u << sigma) / cnt_i) + th;
            };
            for (size_t i = 0; i < cnt; ++i) {
                const u64 q = primes[i];
                for (u32 j = 0; j < n; ++j) {
                    const u32 jp = p3(j);
                    twp[i * n + jp] = make_ulonglong2(tw[i * n + j].x, (u64)(((u128)tw[i * n + j].x << 31) % q));
                    itwp[i * n + jp] = make_ulonglong2(itw[i * n + j].x, (u64)(((u128)itw[i * n + j].x << 31) % q));
                }
            }
            HIP_TRY(hipMalloc(&B.d_tw_pm3, sizeof(ulonglong2) * cnt * n));
            HIP_TRY(hipMalloc(&B.d_itw_pm3, sizeof(ulonglong2) * cnt * n));
            HIP_TRY(hipMemcpy(B.d_tw_pm3, twp.data(), sizeof(ulonglong2) * cnt * n, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(B.d_itw_pm3, itwp.data(), sizeof(ulonglong2) * cnt * n, hipMemcpyHostToDevice));
        }
    }
    if (want_f64) {
        // the same twiddles as centred doubles for the exact-FP64 kernels (dct_fused.hip)
        std::vector<double> twd(cnt * n), itwd(cnt * n);
        for (size_t i = 0; i < cnt; ++i) {
            const u64 q = primes[i];
            auto centred = [q](u64 w) { return w > q / 2 ? -(double)(q - w) : (double)w; };
            for (u32 j = 0; j < n; ++j) {
                twd[i * n + j] = centred(tw[i * n + j].x);
                itwd[i * n + j] = centred(itw[i * n + j].x);
            }
        }
        HIP_TRY(hipMalloc(&B.d_tw_f64, sizeof(double) * cnt * n));
        HIP_TRY(hipMalloc(&B.d_itw_f64, sizeof(double) * cnt * n));
        HIP_TRY(hipMemcpy(B.d_tw_f64, twd.data(), sizeof(double) * cnt * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(B.d_itw_f64, itwd.data(), sizeof(double) * cnt * n, hipMemcpyHostToDevice));
    }
    return FHE_OK;
}
void fhe_free_base(BaseTables &B) {
    if (B.d_tw) (void)hipFree(B.d_tw);
    if (B.d_itw) (void)hipFree(B.d_itw);
    if (B.d_mod) (void)hipFree(B.d_mod);
    if (B.d_tw_f64) (void)hipFree(B.d_tw_f64);
    if (B.d_itw_f64) (void)hipFree(B.d_itw_f64);
    if (B.d_tw_pm) (void)hipFree(B.d_tw_pm);
    if (B.d_itw_pm) (void)hipFree(B.d_itw_pm);
    if (B.d_pm) (void)hipFree(B.d_pm);
    if (B.d_tw_pm3) (void)hipFree(B.d_tw_pm3);
    if (B.d_itw_pm3) (void)hipFree(B.d_itw_pm3);
    B.d_tw_pm = B.d_itw_pm = B.d_tw_pm3 = B.d_itw_pm3 = nullptr;
    B.d_pm = nullptr;
    B.pm_class = 0;
    B.d_tw_f64 = B.d_itw_f64 = nullptr;
    B.d_tw = B.d_itw = nullptr;
    B.d_mod = nullptr;
}

// experiment switches: set and not "0"
static bool env_on(const char *name) { const char *e = getenv(name); return e && *e && !(e[0] == '0' && !e[1]); }

extern "C" int fhe_default_coeff_modulus(uint32_t n, int preset, uint64_t *q_out) {
    // preset 0: small (36..44-bit) prime sets, 109/218 bits total (BASELINE.json "3 coeff moduli" at 4096)
    // preset 1: SEAL 2.3.1 coeff_modulus_128 defaults
    static const u64 s3_4096[] = {0xffffee001ULL, 0xffffc4001ULL, 0x1ffffe0001ULL};
    static const u64 s3_8192[] = {0x7fffffd8001ULL, 0x7fffffc8001ULL, 0xfffffffc001ULL, 0xffffff6c001ULL, 0xfffffebc001ULL};
    static const u64 s23_2048[] = {0x3fffffff000001ULL};
    static const u64 s23_4096[] = {0x7fffffff380001ULL, 0x3fffffff000001ULL};
    static const u64 s23_8192[] = {0x7fffffff380001ULL, 0x7ffffffef00001ULL, 0x3fffffff000001ULL, 0x3ffffffef40001ULL};
    // n = 16384 (the last column of the reference's benchmark grid, benchmark/benchmark.py:6): SEAL 2.3.1's 438-bit default, six
    // 55-bit and two 54-bit primes -- the head of its tables of the largest primes = 1 (mod 2^18) of each width (the rule the entries
    // above obey; recollection).  SEAL 3's default there has nine primes, more than FHE_MAX_K: both presets give this one
    static const u64 s23_16384[] = {0x7fffffff380001ULL, 0x7ffffffef00001ULL, 0x7ffffffeac0001ULL, 0x7ffffffe700001ULL, 0x7ffffffe600001ULL, 0x7ffffffe4c0001ULL,
                                    0x3fffffff000001ULL, 0x3ffffffef40001ULL};
    const u64 *src = nullptr;
    int cnt = 0;
#define PICK(a) do { src = a; cnt = (int)(sizeof(a) / sizeof(a[0])); } while (0)
    if (preset == 0) {
        if (n == 4096) PICK(s3_4096);
        else if (n == 8192) PICK(s3_8192);
        else if (n == 2048 || n == 1024) PICK(s23_2048);
        else if (n == 16384) PICK(s23_16384);
    } else if (preset == 1) {
        if (n == 2048 || n == 1024) PICK(s23_2048);
        else if (n == 4096) PICK(s23_4096);
        else if (n == 8192) PICK(s23_8192);
        else if (n == 16384) PICK(s23_16384);
    }
#undef PICK
    if (!src) return fail(FHE_ERR_PARAM, "no default coefficient modulus for n=%u preset=%d", n, preset);
    for (int i = 0; i < cnt; ++i) q_out[i] = src[i];
    return cnt;
}

extern "C" int fhe_ctx_create(uint32_t n, const uint64_t *q, uint32_t k, uint64_t t, int device, fhe_ctx **out) {
    using namespace hostmath;
    if (!out || !q) return fail(FHE_ERR_PARAM, "null argument");
    *out = nullptr;
    if (k == 0 || k > FHE_MAX_K) return fail(FHE_ERR_PARAM, "coeff modulus count %u out of range", k);
    if (n < 1024 || n > 16384 || (n & (n - 1))) return fail(FHE_ERR_PARAM, "poly_modulus_degree %u unsupported", n);
    if (t < 2) return fail(FHE_ERR_PARAM, "plain modulus too small");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FHE_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(FHE_ERR_PARAM, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    std::vector<u64> primes(k);
    for (u32 i = 0; i < k; ++i) primes[i] = q[i];
    for (u32 i = 0; i < k; ++i) {
        if (primes[i] >> 61) return fail(FHE_ERR_PARAM, "modulus %u exceeds 61 bits", i);
        if (!is_prime(primes[i]) || (primes[i] - 1) % (2ULL * n)) return fail(FHE_ERR_PARAM, "modulus %u is not an NTT prime for n=%u", i, n);
        if (t >= primes[i]) return fail(FHE_ERR_PARAM, "plain modulus must be below every coefficient modulus");
        for (u32 j = 0; j < i; ++j)
            if (primes[j] == primes[i]) return fail(FHE_ERR_PARAM, "duplicate modulus");
    }
    fhe_ctx *c = new fhe_ctx();
    {   // the experiment switches, once (internal.h: FheOptions)
        FheOptions &o = c->opt;
        auto off = [](const char *name) { const char *e = getenv(name); return e && e[0] == '0' && !e[1]; };
        o.force_u64 = env_on("FHE_DCT_FORCE_U64");
        o.dct_pipeline = env_on("FHE_DCT_PIPELINE");
        if (const char *e = getenv("FHE_DCT_WAVE_BLOCKS")) { const u64 v = strtoull(e, nullptr, 10); if (v) o.dct_wave_blocks = v; }
        if (const char *e = getenv("FHE_DCT_LE")) o.dct_le = atoi(e) == 4 ? 4 : 3;
        o.dct_pack = !off("FHE_DCT_PACK");
        o.dct_ldsc = !off("FHE_DCT_LDSC");
        o.dct_u64_fused = !off("FHE_DCT_U64_FUSED");
        if (const char *e = getenv("FHE_DCT_ONE_LAUNCH")) o.dct_one_launch = (u32)atoi(e);
        o.relin_fused = env_on("FHE_RELIN_FUSED");
        o.relin_steps = env_on("FHE_RELIN_STEPS");
        o.enc_unfused = env_on("FHE_ENC_UNFUSED");
        { const char *e = std::getenv("FHE_ENC_OCC"); o.enc_occ4 = e && e[0] == '4'; }
        o.ntt_nolazy = env_on("FHE_NTT_NOLAZY");
        o.ntt_single = env_on("FHE_NTT_SINGLE");
        o.ntt_nopm = env_on("FHE_NTT_NOPM");
        o.behz_aux61 = env_on("FHE_BEHZ_AUX61");
        o.behz_chunk3 = env_on("FHE_BEHZ_CHUNK3");
        o.behz_tensor_canon = env_on("FHE_BEHZ_TENSOR_CANON");
        o.behz_tensor_single = env_on("FHE_BEHZ_TENSOR_SINGLE");
        o.behz_fused_prepare = env_on("FHE_BEHZ_FUSED_PREPARE");
        o.cubic_unfused = env_on("FHE_CUBIC_UNFUSED");
        o.plain_sum_unfused = env_on("FHE_PLAIN_SUM_UNFUSED");
        o.behz_square_full = env_on("FHE_BEHZ_SQUARE_FULL");
    }
    c->n = n;
    c->k = k;
    c->t = t;
    c->device = device;
    while ((1u << c->logn) < n) ++c->logn;
    for (u32 i = 0; i < k; ++i) c->max_prime_bits = std::max(c->max_prime_bits, bit_length(primes[i]));
    int rc = fhe_build_base(c->qb, primes, n, c->logn, c->max_prime_bits <= 47);
    if (rc) { delete c; return rc; }
    // plaintext lifting constants
    BigUInt Q(1);
    for (u32 i = 0; i < k; ++i) Q.mul_small(primes[i]);
    const u64 q_mod_t = Q.mod_small(t);
    c->upper_half_threshold = (t + 1) >> 1;
    for (u32 i = 0; i < k; ++i) {
        const u64 qi = primes[i];
        c->plain_upper_half_increment[i] = submod(0, t % qi, qi);
        c->upper_half_increment[i] = q_mod_t % qi;
        c->delta_mod[i] = mulmod(submod(0, q_mod_t % qi, qi), invmod(t % qi, qi), qi);
    }
    // what the linear circuits need is built here; the ct x ct tables (auxiliary base, conversion constants) are built by the
    // first entry point that multiplies ciphertexts (fhe_behz_ensure, call_once), so a DCT-only server neither pays for them
    // nor can fail on the auxiliary-prime search.  FHE_BEHZ_EAGER=1 restores construction at create time.
    if (env_on("FHE_BEHZ_EAGER") && (rc = fhe_behz_ensure(c))) { fhe_ctx_destroy(c); return rc; }
    if (c->opt.dct_one_launch) {
        c->arrived_cap = 1024 * 2 * (u64)k;
        if (hipMalloc((void **)&c->d_arrived, (c->arrived_cap + 1) * sizeof(u32)) != hipSuccess) { fhe_ctx_destroy(c); return fail(FHE_ERR_HIP, "arrival counters"); }
    }
    // the second stream + events of the pipelined DCT mode exist only in contexts created with FHE_DCT_PIPELINE=1: an idle
    // stream still takes a turn in the runtime's round-robin over its four hardware queues, and a host's own copy stream
    // that lands on the compute stream's queue serialises with it (seal/server_jpeg_hip.cpp measured exactly that)
    bool ok = !c->opt.dct_pipeline || hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && ok && c->opt.dct_pipeline; ++i)
        ok = hipEventCreateWithFlags(&c->ev_rows[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_cols[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { fhe_ctx_destroy(c); return fail(FHE_ERR_HIP, "stream/event creation failed"); }
    *out = c;
    return FHE_OK;
}

extern "C" int fhe_ctx_destroy(fhe_ctx *c) {
    if (!c) return FHE_OK;
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->d_arrived) (void)hipFree(c->d_arrived);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_rows[i]) (void)hipEventDestroy(c->ev_rows[i]);
        if (c->ev_cols[i]) (void)hipEventDestroy(c->ev_cols[i]);
    }
    for (fhe_ctx::RgbConsts *r : c->rgb) {
        if (r->d_c) (void)hipFree(r->d_c);
        if (r->d_c_f64) (void)hipFree(r->d_c_f64);
        if (r->d_off) (void)hipFree(r->d_off);
        delete r;
    }
    fhe_behz_free(c);
    fhe_free_base(c->qb);
    delete c;
    return FHE_OK;
}
extern "C" int fhe_ctx_device(const fhe_ctx *c) { return c ? c->device : -1; }
extern "C" int fhe_ctx_bind_thread(const fhe_ctx *c) {
    if (!c) return fail(FHE_ERR_PARAM, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return FHE_OK;
}
extern "C" int fhe_stream_create(fhe_stream *out) {
    if (!out) return fail(FHE_ERR_PARAM, "null argument");
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (fhe_stream)s;
    return FHE_OK;
}
extern "C" int fhe_stream_destroy(fhe_stream s) {
    if (s) HIP_TRY(hipStreamDestroy((hipStream_t)s));
    return FHE_OK;
}
extern "C" uint32_t fhe_ctx_n(const fhe_ctx *c) { return c->n; }
extern "C" uint32_t fhe_ctx_k(const fhe_ctx *c) { return c->k; }
extern "C" uint64_t fhe_ctx_t(const fhe_ctx *c) { return c->t; }
extern "C" uint64_t fhe_ctx_q(const fhe_ctx *c, uint32_t i) { return i < c->k ? c->qb.primes[i] : 0; }

// ------------------------------------------------------------------------------------------------
// memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int fhe_dev_alloc(size_t bytes, void **dptr) {
    if (!dptr) return fail(FHE_ERR_PARAM, "null argument");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e == hipErrorOutOfMemory) return fail(FHE_ERR_NOMEM, "hipMalloc(%zu) out of memory", bytes);
    if (e != hipSuccess) return fail(FHE_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
    return FHE_OK;
}
extern "C" int fhe_dev_free(void *p) { if (p) HIP_TRY(hipFree(p)); return FHE_OK; }
extern "C" int fhe_host_alloc(size_t bytes, void **hptr) {
    if (!hptr) return fail(FHE_ERR_PARAM, "null argument");
    *hptr = nullptr;
    if (!bytes) return FHE_OK;
    HIP_TRY(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return FHE_OK;
}
extern "C" int fhe_host_free(void *p) { if (p) HIP_TRY(hipHostFree(p)); return FHE_OK; }
extern "C" int fhe_upload(void *d, const void *h, size_t bytes, fhe_stream s) {
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return FHE_OK;
}
extern "C" int fhe_download(void *h, const void *d, size_t bytes, fhe_stream s) {
    HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    HIP_TRY(hipStreamSynchronize((hipStream_t)s));
    return FHE_OK;
}
extern "C" int fhe_copy(void *dst, const void *src, size_t bytes, fhe_stream s) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return FHE_OK;
}
extern "C" int fhe_stream_sync(fhe_stream s) { HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return FHE_OK; }

// fhe_gather: `count` scattered device buffers of words_each u64 -> one strided batch.  Up to 256 sources travel in the
// kernel arguments (2 KB): no staging copy, no host synchronisation, one launch.  More sources go through a pointer table in
// device memory that lives exactly as long as the launch needs it (stream-ordered allocation, a staged copy of the host
// array, ONE launch, stream-ordered release) -- round 4 issued one launch per 256 sources, which was more than half of the
// launches of the reference's server_jpeg through the lazy facade (6,912 operands per level: 27 launches per gather).
// This is what turns the facade's one-ciphertext-at-a-time calls into batched launches (seal/seal.h, lazy evaluation).
namespace {
constexpr int GATHER_PTRS = 256;
struct GatherArgs { const ulonglong2 *src[GATHER_PTRS]; };
__global__ __launch_bounds__(256) void k_gather(GatherArgs a, ulonglong2 *__restrict__ dst, u64 pairs_each, u64 dst_stride_pairs) {
    const ulonglong2 *__restrict__ s = a.src[blockIdx.y];
    ulonglong2 *__restrict__ d = dst + (u64)blockIdx.y * dst_stride_pairs;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < pairs_each; i += (u64)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_gather_table(const ulonglong2 *const *__restrict__ table, u64 count, ulonglong2 *__restrict__ dst, u64 pairs_each,
                                                      u64 dst_stride_pairs) {
    for (u64 c = blockIdx.y; c < count; c += gridDim.y) {
        const ulonglong2 *__restrict__ s = table[c];
        ulonglong2 *__restrict__ d = dst + c * dst_stride_pairs;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < pairs_each; i += (u64)gridDim.x * blockDim.x) d[i] = s[i];
    }
}
}  // namespace
extern "C" int fhe_gather(const uint64_t *const *src_host, uint64_t count, uint64_t words_each, uint64_t *dst, uint64_t dst_stride_words, fhe_stream s) {
    if (!count || !words_each) return FHE_OK;
    if (!src_host || !dst) return fail(FHE_ERR_PARAM, "null argument");
    if ((words_each & 1) || (dst_stride_words & 1) || dst_stride_words < words_each || ((uintptr_t)dst & 15))
        return fail(FHE_ERR_PARAM, "gather moves 16-byte units: even word counts and 16-byte aligned buffers");
    for (u64 i = 0; i < count; ++i)
        if (!src_host[i] || ((uintptr_t)src_host[i] & 15)) return fail(FHE_ERR_PARAM, "gather source %llu is null or not 16-byte aligned", (unsigned long long)i);
    const u64 pairs = words_each / 2;
    const unsigned bx = (unsigned)std::min<u64>((pairs + 255) / 256, 64);
    hipStream_t st = (hipStream_t)s;
    if (count <= GATHER_PTRS) {
        GatherArgs a;
        for (unsigned i = 0; i < (unsigned)count; ++i) a.src[i] = (const ulonglong2 *)src_host[i];
        for (unsigned i = (unsigned)count; i < GATHER_PTRS; ++i) a.src[i] = nullptr;
        k_gather<<<dim3(bx, (unsigned)count), 256, 0, st>>>(a, (ulonglong2 *)dst, pairs, dst_stride_words / 2);
        KERNEL_CHECK();
        return FHE_OK;
    }
    // Pointer tables go through the per-device staging ring (fhe_stage_acquire below): the caller's array is pageable (handing it to
    // hipMemcpyAsync would make the call wait for the stream), and a slot is reused only after the gather KERNEL that read its
    // device half has run.  A first version took the device table from hipMallocAsync / hipFreeAsync around every launch: on the
    // default stream the kernel then read zeroed table entries now and then (memory access faults at addresses near 0 in the
    // reference's server_resize through the lazy facade; tests/test_reference_published_resize.py) -- the stream-ordered pool is
    // not used any more.
    const u64 slot_ptrs = FHE_STAGE_SLOT_BYTES / sizeof(void *);
    for (u64 done = 0; done < count; done += slot_ptrs) {
        const u64 part = std::min<u64>(slot_ptrs, count - done);
        FheStage sg;
        int rc = fhe_stage_acquire(src_host + done, part * sizeof(void *), st, &sg);
        if (rc) return rc;
        k_gather_table<<<dim3(bx, (unsigned)part), 256, 0, st>>>((const ulonglong2 *const *)sg.dev, part, (ulonglong2 *)(dst + done * dst_stride_words), pairs,
                                                                dst_stride_words / 2);
        const hipError_t le = hipGetLastError();
        rc = fhe_stage_release(sg, st);
        if (le != hipSuccess) return fail(FHE_ERR_HIP, "kernel launch: %s", hipGetErrorString(le));
        if (rc) return rc;
    }
    return FHE_OK;
}

// Small host -> device hand-overs (pointer tables, batches of doubles): a process-wide ring per device of kSlots page-locked host
// slots, each with its own device slot, allocated once and never freed (static destructors may run after the runtime is gone).
// acquire: copies `bytes` into a free slot, enqueues the upload on `st`, returns the device address; the caller launches the kernel
// that reads it on the SAME stream and calls release, which records the slot's event behind that kernel -- the slot is handed out
// again only after that event has completed.  Thread-safe: a slot between acquire and release is skipped by other threads.
namespace {
struct StageRing {
    enum : int { kSlots = 8 };
    std::mutex mu;
    std::condition_variable cv;
    char *pinned = nullptr, *device = nullptr;
    hipEvent_t ev[kSlots] = {};
    bool used[kSlots] = {}, busy[kSlots] = {};
    int next = 0;
};
StageRing *g_stage_rings[32] = {};
std::mutex g_stage_rings_mu;
}  // namespace
int fhe_stage_acquire(const void *host, size_t bytes, hipStream_t st, FheStage *out) {
    if (!host || !out || !bytes || bytes > FHE_STAGE_SLOT_BYTES) return fail(FHE_ERR_PARAM, "staging: %zu bytes do not fit a slot", bytes);
    int dev = 0;
    // the slot must live on the device the upload and its reader run on: the stream's device (null / legacy stream: the calling
    // thread's current device, on which every launch of this library acts, include/fhe_hip.h)
    if (!st || hipStreamGetDevice(st, &dev) != hipSuccess) HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 32) return fail(FHE_ERR_PARAM, "staging: device %d out of range", dev);
    StageRing *ring;
    {
        std::lock_guard<std::mutex> lk0(g_stage_rings_mu);
        if (!g_stage_rings[dev]) g_stage_rings[dev] = new StageRing();
        ring = g_stage_rings[dev];
    }
    std::unique_lock<std::mutex> lk(ring->mu);
    if (!ring->pinned) {
        hipError_t e = hipHostMalloc((void **)&ring->pinned, (size_t)StageRing::kSlots * FHE_STAGE_SLOT_BYTES, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&ring->device, (size_t)StageRing::kSlots * FHE_STAGE_SLOT_BYTES);
        for (int i = 0; i < StageRing::kSlots && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ring->ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {                                       // give back whatever was obtained: the next call starts from nothing
            for (int i = 0; i < StageRing::kSlots; ++i)
                if (ring->ev[i]) { (void)hipEventDestroy(ring->ev[i]); ring->ev[i] = nullptr; }
            if (ring->device) (void)hipFree(ring->device);
            if (ring->pinned) (void)hipHostFree(ring->pinned);
            ring->pinned = ring->device = nullptr;
            return fail(FHE_ERR_HIP, "staging ring: %s", hipGetErrorString(e));
        }
    }
    int slot = -1;
    for (;;) {
        for (int i = 0; i < StageRing::kSlots; ++i) {
            const int cand = (ring->next + i) % StageRing::kSlots;
            if (!ring->busy[cand]) { slot = cand; break; }
        }
        if (slot >= 0) break;
        ring->cv.wait(lk);
    }
    ring->next = (slot + 1) % StageRing::kSlots;
    ring->busy[slot] = true;
    const bool wait = ring->used[slot];
    lk.unlock();
    auto give_back = [&] { std::lock_guard<std::mutex> g(ring->mu); ring->busy[slot] = false; ring->cv.notify_one(); };
    if (wait && hipEventSynchronize(ring->ev[slot]) != hipSuccess) { give_back(); return fail(FHE_ERR_HIP, "staging ring: event wait failed"); }
    char *hp = ring->pinned + (size_t)slot * FHE_STAGE_SLOT_BYTES, *dp = ring->device + (size_t)slot * FHE_STAGE_SLOT_BYTES;
    memcpy(hp, host, bytes);
    if (hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, st) != hipSuccess) { give_back(); return fail(FHE_ERR_HIP, "staging ring: upload failed"); }
    out->dev = dp;
    out->slot = slot;
    out->ring = ring;
    return FHE_OK;
}
int fhe_stage_release(const FheStage &sg, hipStream_t st) {
    StageRing *ring = (StageRing *)sg.ring;
    const hipError_t e = hipEventRecord(ring->ev[sg.slot], st);
    std::lock_guard<std::mutex> g(ring->mu);
    ring->used[sg.slot] = true;                                      // even after a failed record: the next user then waits on the slot's previous event at worst
    ring->busy[sg.slot] = false;
    ring->cv.notify_one();
    return e == hipSuccess ? FHE_OK : fail(FHE_ERR_HIP, "staging ring: event record failed");
}

// ------------------------------------------------------------------------------------------------
// FractionalEncoder (host)
// ------------------------------------------------------------------------------------------------
extern "C" int fhe_frac_encode(uint32_t n, uint64_t t, double value, int int_coeffs, int frac_coeffs, uint64_t *plain) {
    if (!plain || int_coeffs < 0 || frac_coeffs < 0 || (uint32_t)(int_coeffs + frac_coeffs) > n)
        return fail(FHE_ERR_PARAM, "encoder coefficient counts do not fit the polynomial");
    if (!std::isfinite(value) || std::fabs(value) >= 9.0e18) return fail(FHE_ERR_PARAM, "value out of range");
    memset(plain, 0, sizeof(uint64_t) * n);
    const int64_t whole = (int64_t)value;              // truncation toward zero
    double frac = value - (double)whole;
    uint64_t mag = whole < 0 ? (uint64_t)(-whole) : (uint64_t)whole;
    for (int d = 0; mag; ++d, mag >>= 1) {
        if (!(mag & 1)) continue;
        if (d >= int_coeffs) return fail(FHE_ERR_PARAM, "integer part needs more than %d coefficients", int_coeffs);
        plain[d] = whole < 0 ? t - 1 : 1;
    }
    if (frac != 0.0) {
        const bool negative = value < 0;
        for (int i = 1; i <= frac_coeffs; ++i) {       // digit of weight 2^-i sits at x^(n-i) with flipped sign
            frac *= 2.0;
            const int64_t bit = (int64_t)frac;
            frac -= (double)bit;
            if (bit) plain[n - i] = negative ? 1 : t - 1;
        }
    }
    int len = (int)n;
    while (len > 0 && plain[len - 1] == 0) --len;
    return len;
}

extern "C" double fhe_frac_decode(uint32_t n, uint64_t t, const uint64_t *plain, int int_coeffs, int frac_coeffs) {
    (void)frac_coeffs;   // every coefficient above the integer part is fractional (products push digits down)
    const uint64_t half = (t + 1) >> 1;
    auto centred = [&](uint64_t m) { return m >= half ? -(double)(t - m) : (double)m; };
    double whole = 0.0, frac = 0.0;
    for (int d = int_coeffs - 1; d >= 0; --d) whole = whole * 2.0 + centred(plain[d]);
    for (uint32_t i = (uint32_t)int_coeffs; i < n; ++i) frac = (frac + centred(plain[i])) * 0.5;
    return whole - frac;
}

// ------------------------------------------------------------------------------------------------
// element-wise kernels
// ------------------------------------------------------------------------------------------------
enum { OP_ADD = 0, OP_SUB = 1, OP_NEG = 2 };

// One thread handles two adjacent coefficients (16-byte accesses).  grid.y walks residue
// polynomials so the modulus is uniform per block.
template <int OP>
__global__ __launch_bounds__(256) void k_eltwise(const ulonglong2 *__restrict__ a, const ulonglong2 *__restrict__ b,
                                                 ulonglong2 *__restrict__ out, const Modulus *__restrict__ mods,
                                                 u32 k, u32 half_n, u64 n_res_polys) {
    for (u64 rp = blockIdx.y; rp < n_res_polys; rp += gridDim.y) {
        const u64 q = mods[rp % k].q;
        const u64 base = rp * half_n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < half_n; i += gridDim.x * blockDim.x) {
            ulonglong2 x = a[base + i], r;
            if (OP == OP_NEG) {
                r.x = negmod(x.x, q);
                r.y = negmod(x.y, q);
            } else {
                ulonglong2 y = b[base + i];
                r.x = OP == OP_ADD ? addmod(x.x, y.x, q) : submod(x.x, y.x, q);
                r.y = OP == OP_ADD ? addmod(x.y, y.y, q) : submod(x.y, y.y, q);
            }
            out[base + i] = r;
        }
    }
}

static int launch_eltwise(int op, const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out,
                          uint64_t n_polys, fhe_stream s) {
    if (!c || !a || !out || (op != OP_NEG && !b)) return fail(FHE_ERR_PARAM, "null argument");
    if (n_polys == 0) return FHE_OK;
    const u64 nrp = n_polys * c->k;
    const u32 half_n = c->n / 2;
    dim3 grid((half_n + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
    auto A = (const ulonglong2 *)a;
    auto B = (const ulonglong2 *)b;
    auto O = (ulonglong2 *)out;
    hipStream_t st = (hipStream_t)s;
    if (op == OP_ADD) k_eltwise<OP_ADD><<<grid, 256, 0, st>>>(A, B, O, c->qb.d_mod, c->k, half_n, nrp);
    else if (op == OP_SUB) k_eltwise<OP_SUB><<<grid, 256, 0, st>>>(A, B, O, c->qb.d_mod, c->k, half_n, nrp);
    else k_eltwise<OP_NEG><<<grid, 256, 0, st>>>(A, B, O, c->qb.d_mod, c->k, half_n, nrp);
    KERNEL_CHECK();
    return FHE_OK;
}
extern "C" int fhe_add(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n_polys, fhe_stream s) {
    return launch_eltwise(OP_ADD, c, a, b, out, n_polys, s);
}
extern "C" int fhe_sub(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n_polys, fhe_stream s) {
    return launch_eltwise(OP_SUB, c, a, b, out, n_polys, s);
}
extern "C" int fhe_negate(const fhe_ctx *c, const uint64_t *a, uint64_t *out, uint64_t n_polys, fhe_stream s) {
    return launch_eltwise(OP_NEG, c, a, nullptr, out, n_polys, s);
}

// c_0[i] += / -= v[i] for `count` ciphertexts (add_plain / sub_plain)
__global__ void k_add_plain(u64 *ct, u64 stride_words, u64 count, const u64 *__restrict__ vals, u32 len,
                            const Modulus *__restrict__ mods, u32 k, u32 n, int sign) {
    const u32 prime = blockIdx.y;
    const u64 q = mods[prime].q;
    for (u64 cidx = blockIdx.z; cidx < count; cidx += gridDim.z) {
        u64 *p = ct + cidx * stride_words + (u64)prime * n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
            const u64 v = vals[(u64)prime * len + i];
            p[i] = sign > 0 ? addmod(p[i], v, q) : submod(p[i], v, q);
        }
    }
}

// the same for a plaintext with few non-zero coefficients: the scaled terms travel in the kernel
// arguments, so nothing is staged and the call is fully asynchronous
#define ADD_PLAIN_MAX_TERMS 24
struct PlainTerms {
    u32 count;
    u32 idx[ADD_PLAIN_MAX_TERMS];
    u64 v[ADD_PLAIN_MAX_TERMS][FHE_MAX_K];
};
__global__ void k_add_plain_terms(u64 *ct, u64 stride_words, u64 count, const Modulus *__restrict__ mods, u32 k, u32 n, int sign, const PlainTerms T) {
    const u64 total = count * k * T.count;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u32 t = (u32)(i % T.count);
        const u32 prime = (u32)((i / T.count) % k);
        const u64 cidx = i / ((u64)T.count * k);
        const u64 q = mods[prime].q;
        u64 *p = ct + cidx * stride_words + (u64)prime * n + T.idx[t];
        *p = sign > 0 ? addmod(*p, T.v[t][prime], q) : submod(*p, T.v[t][prime], q);
    }
}

extern "C" int fhe_add_plain(const fhe_ctx *c, uint64_t *ct, uint64_t stride, uint64_t count, const uint64_t *plain,
                             uint32_t len, int sign, fhe_stream s) {
    using namespace hostmath;
    if (!c || !ct || (!plain && len)) return fail(FHE_ERR_PARAM, "null argument");
    if (len > c->n) return fail(FHE_ERR_PARAM, "plaintext longer than the polynomial");
    if (sign != 1 && sign != -1) return fail(FHE_ERR_PARAM, "sign must be +1 or -1");
    if (len == 0 || count == 0) return FHE_OK;
    {
        PlainTerms T;
        T.count = 0;
        bool sparse = true;
        for (u32 j = 0; j < len; ++j) {
            const u64 m = plain[j];
            if (!m) continue;
            if (m >= c->t) return fail(FHE_ERR_PARAM, "plaintext coefficient %u not below the plain modulus", j);
            if (T.count == ADD_PLAIN_MAX_TERMS) { sparse = false; break; }
            T.idx[T.count] = j;
            for (u32 i = 0; i < c->k; ++i) {
                const u64 qi = c->qb.primes[i];
                u64 v = mulmod(c->delta_mod[i], m % qi, qi);
                if (m >= c->upper_half_threshold) v = addmod(v, c->upper_half_increment[i], qi);
                T.v[T.count][i] = v;
            }
            ++T.count;
        }
        if (sparse) {
            if (!T.count) return FHE_OK;
            const u64 total = count * c->k * T.count;
            const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
            k_add_plain_terms<<<blocks, 256, 0, (hipStream_t)s>>>((u64 *)ct, stride, count, c->qb.d_mod, c->k, c->n, sign, T);
            KERNEL_CHECK();
            return FHE_OK;
        }
    }
    std::vector<u64> vals((size_t)c->k * len);
    for (u32 i = 0; i < c->k; ++i) {
        const u64 qi = c->qb.primes[i];
        for (u32 j = 0; j < len; ++j) {
            const u64 m = plain[j];
            if (m >= c->t) return fail(FHE_ERR_PARAM, "plaintext coefficient %u not below the plain modulus", j);
            u64 v = mulmod(c->delta_mod[i], m % qi, qi);
            if (m >= c->upper_half_threshold) v = addmod(v, c->upper_half_increment[i], qi);
            vals[(size_t)i * len + j] = v;
        }
    }
    // Host-sourced operand: staged through a private device buffer with fully synchronous
    // semantics (allocate, blocking copy, launch, wait, free).  Stream-ordered pool allocations
    // combined with pageable async copies were observed to let a later call's copy overtake an
    // earlier call's kernel; this entry point is not on the throughput path.
    hipStream_t st = (hipStream_t)s;
    u64 *d_vals = nullptr;
    HIP_TRY(hipMalloc((void **)&d_vals, vals.size() * sizeof(u64)));
    hipError_t e = hipMemcpy(d_vals, vals.data(), vals.size() * sizeof(u64), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        dim3 grid((len + 255) / 256, c->k, (unsigned)(count < 16384 ? count : 16384));
        k_add_plain<<<grid, 256, 0, st>>>((u64 *)ct, stride, count, d_vals, len, c->qb.d_mod, c->k, c->n, sign);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(d_vals);
    if (e != hipSuccess) return fail(FHE_ERR_HIP, "add_plain: %s", hipGetErrorString(e));
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// NTT kernels: one workgroup per residue polynomial
// ------------------------------------------------------------------------------------------------
template <int L, bool LAZY>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_ntt_fwd(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    const int tid = threadIdx.x;
    const u64 rp = blockIdx.x;
    const u32 prime = (u32)(rp % base.count);
    const u64 q = base.mod[prime].q;
    const NttMod m = ntt_mod(q);
    u64 x[16];
    load_coeff<L>(x, in + rp * NttShape<L>::N, tid);
    ntt_fwd_regs4<L, LAZY>(x, base.tw + (size_t)prime * NttShape<L>::N, m, lds, tid);
    if constexpr (LAZY) {       // every prime <= 58 bits: no conditional subtraction in the butterflies, one product with 1 at the end
        const float cs = canon_scale(q);                          // outputs below (2 + 4 log2 n) q <= 58q: canon_below_64q
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = canon_below_64q(x[r], q, cs);
    } else {                    // below 8q
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = csub(csub(csub(x[r], m.q4), 2 * q), q);
    }
    store_slots<L>(x, out + rp * NttShape<L>::N, tid);
}

// two polynomials of one prime per workgroup (rp and rp + pair_stride): every twiddle serves both
template <int L, bool LAZY>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_ntt_fwd2(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base, u32 pair_stride) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    const int tid = threadIdx.x;
    // blockIdx -> (pair g, prime): residue polynomials (2g) * pair_stride + prime and (2g + 1) * pair_stride + prime
    const u32 prime = blockIdx.x % base.count;
    const u64 g = blockIdx.x / base.count;
    const u64 rp0 = (2 * g) * pair_stride + prime, rp1 = rp0 + pair_stride;
    const u64 q = base.mod[prime].q;
    const NttMod m = ntt_mod(q);
    u64 x[2][16];
    load_coeff<L>(x[0], in + rp0 * NttShape<L>::N, tid);
    load_coeff<L>(x[1], in + rp1 * NttShape<L>::N, tid);
    ntt_fwd_regs4m<L, LAZY, 2>(x, base.tw + (size_t)prime * NttShape<L>::N, m, lds, tid);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if constexpr (LAZY) {
            const float cs = canon_scale(q);
#pragma unroll
            for (int r = 0; r < 16; r++) x[j][r] = canon_below_64q(x[j][r], q, cs);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) x[j][r] = csub(csub(csub(x[j][r], m.q4), 2 * q), q);
        }
    }
    store_slots<L>(x[0], out + rp0 * NttShape<L>::N, tid);
    store_slots<L>(x[1], out + rp1 * NttShape<L>::N, tid);
}

template <int L, bool LAZY>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_ntt_inv(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    const int tid = threadIdx.x;
    const u64 rp = blockIdx.x;
    const u32 prime = (u32)(rp % base.count);
    const u64 q = base.mod[prime].q;
    const NttMod m = ntt_mod(q);
    u64 x[16];
    load_slots<L>(x, in + rp * NttShape<L>::N, tid);
    ntt_inv_regs4<L, LAZY>(x, base.itw + (size_t)prime * NttShape<L>::N, m, lds, tid);       // [0, 4q)
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = csub(csub(x[r], 2 * q), q);
    store_coeff<L>(x, out + rp * NttShape<L>::N, tid);
}

template <int L, bool LAZY>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_ntt_inv2(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base, u32 pair_stride) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    const int tid = threadIdx.x;
    const u32 prime = blockIdx.x % base.count;
    const u64 g = blockIdx.x / base.count;
    const u64 rp0 = (2 * g) * pair_stride + prime, rp1 = rp0 + pair_stride;
    const u64 q = base.mod[prime].q;
    const NttMod m = ntt_mod(q);
    u64 x[2][16];
    load_slots<L>(x[0], in + rp0 * NttShape<L>::N, tid);
    load_slots<L>(x[1], in + rp1 * NttShape<L>::N, tid);
    ntt_inv_regs4m<L, 2, LAZY>(x, base.itw + (size_t)prime * NttShape<L>::N, m, lds, tid);
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[j][r] = csub(csub(x[j][r], 2 * q), q);
    }
    store_coeff<L>(x[0], out + rp0 * NttShape<L>::N, tid);
    store_coeff<L>(x[1], out + rp1 * NttShape<L>::N, tid);
}

// The same transforms on the pseudo-Mersenne passes (ntt_core.h), M polynomials of one prime per workgroup
// (residue polynomials (M g + j) pair_stride + prime); C = PmA / PmB, the class of the base.
template <int L, int M, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_ntt_fwd_pm(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base, u32 pair_stride) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u32 prime = blockIdx.x % base.count;
    const u64 g = blockIdx.x / base.count;
    const PmMod m = base.pm[prime];
    u64 x[M][16];
#pragma unroll
    for (int j = 0; j < M; j++) load_coeff<L>(x[j], in + ((M * g + j) * pair_stride + prime) * N, tid);
    ntt_fwd_regs_pm<L, M, 16, C::LIM, C::CS>(x, base.tw_pm + (size_t)prime * N, m, lds, tid);
#pragma unroll
    for (int j = 0; j < M; j++) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[j][r] = canon_pm(x[j][r], m);
        store_slots<L>(x[j], out + ((M * g + j) * pair_stride + prime) * N, tid);
    }
}
template <int L, int M, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_ntt_inv_pm(const u64 *__restrict__ in, u64 *__restrict__ out, RnsBase base, u32 pair_stride) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u32 prime = blockIdx.x % base.count;
    const u64 g = blockIdx.x / base.count;
    const PmMod m = base.pm[prime];
    u64 x[M][16];
#pragma unroll
    for (int j = 0; j < M; j++) load_slots<L>(x[j], in + ((M * g + j) * pair_stride + prime) * N, tid);
    ntt_inv_regs_pm<L, M, 16, C::XB, C::LIM, C::RQ>(x, base.itw_pm + (size_t)prime * N, m, lds, tid);      // canonical in
#pragma unroll
    for (int j = 0; j < M; j++) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[j][r] = canon_rq_pm<C::RQ>(x[j][r], m);
        store_coeff<L>(x[j], out + ((M * g + j) * pair_stride + prime) * N, tid);
    }
}

// multiply_plain: NTT -> dyadic product with a prepared plaintext (Shoup pairs) -> inverse NTT
template <int L, bool LAZY>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_mulplain(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                               const ulonglong2 *__restrict__ plain, RnsBase base) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid = threadIdx.x;
    const u64 rp = blockIdx.x;
    const u32 prime = (u32)(rp % base.count);
    const u64 q = base.mod[prime].q;
    const NttMod m = ntt_mod(q);
    u64 x[16];
    load_coeff<L>(x, in + rp * N, tid);
    ntt_fwd_regs4<L, LAZY>(x, base.tw + (size_t)prime * N, m, lds, tid);     // LAZY: every prime <= 58 bits; the product takes any operand
    asm volatile("" ::: "memory");       // keep the 16 plaintext pairs (64 VGPRs) from being fetched before the transform
    const ulonglong2 *pl = plain + (size_t)prime * N;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const ulonglong2 w = pl[r * TP + tid];
        x[r] = mul_shoup_lazy4(x[r], w.x, w.y, m.nq, m.zero);
    }
    ntt_inv_regs4<L, LAZY>(x, base.itw + (size_t)prime * N, m, lds, tid);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = csub(csub(x[r], 2 * q), q);
    store_coeff<L>(x, out + rp * N, tid);
}

// multiply_plain on the pseudo-Mersenne passes, M polynomials of one prime per workgroup (they share every twiddle and
// every plaintext slot); the dyadic product takes the plaintext value alone (mulvv_pm), not its Shoup companion
template <int L, int M, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_mulplain_pm(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                                  const ulonglong2 *__restrict__ plain, RnsBase base, u32 pair_stride) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid = threadIdx.x;
    const u32 prime = blockIdx.x % base.count;
    const u64 g = blockIdx.x / base.count;
    const PmMod m = base.pm[prime];
    u64 x[M][16];
#pragma unroll
    for (int j = 0; j < M; j++) load_coeff<L>(x[j], in + ((M * g + j) * pair_stride + prime) * N, tid);
    ntt_fwd_regs_pm<L, M, 16, C::LIM, C::CS>(x, base.tw_pm + (size_t)prime * N, m, lds, tid);
    PM_FENCE();                          // keep the 16 plaintext values from being fetched before the transform
    const ulonglong2 *pl = plain + (size_t)prime * N;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u64 w = pl[r * TP + tid].x;
#pragma unroll
        for (int j = 0; j < M; j++) x[j][r] = mulvv_pm(fold_pm(x[j][r], m), w, m);
    }
    ntt_inv_regs_pm<L, M, C::RQ, C::XB, C::LIM, C::RQ>(x, base.itw_pm + (size_t)prime * N, m, lds, tid);
#pragma unroll
    for (int j = 0; j < M; j++) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[j][r] = canon_rq_pm<C::RQ>(x[j][r], m);
        store_coeff<L>(x[j], out + ((M * g + j) * pair_stride + prime) * N, tid);
    }
}

// sum of plaintext products with ONE inverse transform: out[c][p] = addend[amap(c)][p] + sum over the terms i with p < size_i of
// src_i[c][p] * plain_i.  multiply_plain is linear, so INTT(sum_i NTT(src_i) . plain_i) is the same ring element -- and the same
// canonical residues -- as the sum of the separate multiply_plain results; per output polynomial the terms cost one forward
// transform each and share the inverse one (the Taylor sums of homomorphic_sin / cos: 35 + 11 transforms per ciphertext instead
// of 35 + 35; the harmonic sum of approximated_step: degree + 1 instead of 2 degree).  Two kernels: forward transform + slot
// product IN PLACE over each term (values below RQ, slot order), then sum of the terms' slots + inverse transform + addend.
// (One kernel that keeps the partial sums in registers across the terms' forward transforms was measured first: 223 spilled
// VGPRs, slower than the separate calls.)
template <int L, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_mulplain_fwd_pm(u64 *__restrict__ io, const ulonglong2 *__restrict__ plain, RnsBase base) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid = threadIdx.x;
    const u32 prime = blockIdx.x % base.count;
    const PmMod m = base.pm[prime];
    u64 x[1][16];
    load_coeff<L>(x[0], io + (size_t)blockIdx.x * N, tid);
    ntt_fwd_regs_pm<L, 1, 16, C::LIM, C::CS>(x, base.tw_pm + (size_t)prime * N, m, lds, tid);
    PM_FENCE();
    const ulonglong2 *pl = plain + (size_t)prime * N;
#pragma unroll
    for (int r = 0; r < 16; r++) x[0][r] = mulvv_pm(fold_pm(x[0][r], m), pl[r * TP + tid].x, m);
    store_slots<L>(x[0], io + (size_t)blockIdx.x * N, tid);
}
// the partial sums stay below 8 RQ + 17/16 q < 2^62 (folded every eighth term)
template <int L, typename C>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_sum_inv_pm(const PlainSumTerms T, const u64 *addend, CMap amap, u32 addend_size,
                                                                   u64 *out, u32 out_size, RnsBase base) {      // out may be addend (the harmonic sum accumulates in place)
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u32 prime = blockIdx.x % base.count;
    const u64 cp = blockIdx.x / base.count;
    const u32 poly = (u32)(cp % out_size);
    const u64 ct = cp / out_size;
    const PmMod m = base.pm[prime];
    u64 y[1][16];
#pragma unroll
    for (int r = 0; r < 16; r++) y[0][r] = 0;
    u32 pending = 0;
    for (u32 i = 0; i < T.count; i++) {
        if (poly >= T.size[i]) continue;
        u64 x[16];
        load_slots<L>(x, T.src[i] + ((ct * T.size[i] + poly) * base.count + prime) * N, tid);
        if (++pending == 8) {
            pending = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) y[0][r] = fold_pm(y[0][r], m);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) y[0][r] += x[r];
    }
#pragma unroll
    for (int r = 0; r < 16; r++) y[0][r] = fold_pm(y[0][r], m);
    ntt_inv_regs_pm<L, 1, C::RQ, C::XB, C::LIM, C::RQ>(y, base.itw_pm + (size_t)prime * N, m, lds, tid);
    const u64 *pa = (addend && poly < addend_size) ? addend + ((amap(ct) * addend_size + poly) * base.count + prime) * N : nullptr;
    u64 *po = out + (size_t)blockIdx.x * N;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        u64 v = canon_rq_pm<C::RQ>(y[0][r], m);
        if (pa) v = addmod(v, pa[elem_index<L - 4>(tid, r)], m.q);
        po[elem_index<L - 4>(tid, r)] = v;
    }
}
bool fhe_multiply_plain_sum_supported(const fhe_ctx *c) {
    return c && c->qb.pm_class && !c->opt.ntt_nopm && !(fhe_rgb_f64_supported(c) && !c->opt.force_u64) && !c->opt.plain_sum_unfused;
}
int fhe_multiply_plain_sum(const fhe_ctx *c, const PlainSumTerms &T, const u64 *addend, CMap amap, u32 addend_size, u64 *out, u32 out_size, u64 count,
                           hipStream_t st) {
    if (!fhe_multiply_plain_sum_supported(c)) return fail(FHE_ERR_PARAM, "multiply_plain sum: no pseudo-Mersenne transform for this context");
    if (T.count < 1 || T.count > FHE_PLAIN_SUM_MAX_TERMS || !out || !out_size) return fail(FHE_ERR_PARAM, "multiply_plain sum: bad arguments");
    const u64 nrp = count * out_size * c->k;
    if (!nrp) return FHE_OK;
    if (nrp > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    const RnsBase base = c->qb.dev();
    for (u32 i = 0; i < T.count; i++) {
        if (!T.src[i] || !T.plain[i] || T.size[i] > out_size) return fail(FHE_ERR_PARAM, "multiply_plain sum: bad term %u", i);
        const u64 trp = count * T.size[i] * c->k;
        if (!trp) continue;
#define GO_PM(CC) DISPATCH_L(c->logn, (k_mulplain_fwd_pm<L, CC><<<(unsigned)trp, NttShape<L>::TP, 0, st>>>(T.src[i], T.plain[i], base)))
        if (c->qb.pm_class == 1) { GO_PM(PmA); }
        else { GO_PM(PmB); }
#undef GO_PM
    }
#define GO_PM(CC) DISPATCH_L(c->logn, (k_sum_inv_pm<L, CC><<<(unsigned)nrp, NttShape<L>::TP, 0, st>>>(T, addend, amap, addend_size, out, out_size, base)))
    if (c->qb.pm_class == 1) { GO_PM(PmA); }
    else { GO_PM(PmB); }
#undef GO_PM
    KERNEL_CHECK();
    return FHE_OK;
}

int fhe_ntt_launch(bool inverse, const fhe_ctx *c, const BaseTables &B, const u64 *in, u64 *out, u64 n_res_polys, hipStream_t st) {
    if (n_res_polys == 0) return FHE_OK;
    if (n_res_polys > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    const RnsBase base = B.dev();
    bool lazy = !c->opt.ntt_nolazy;        // forward transform without conditional subtractions, inverse with static range tracking: every prime of the base <= 58 bits
    // n >= 8192: two polynomials of one prime per workgroup, every twiddle pair fetched once for both (P8192 forward +7 %,
    // inverse +19 %; at n = 4096 the pair kernels spill and are slower, so single polynomials stay there)
    const bool pair = c->logn >= 13 && (n_res_polys / base.count) % 2 == 0 && !c->opt.ntt_single;
    for (u64 p : B.primes) lazy = lazy && (p >> 58) == 0 && (p >> 33) != 0;      // canon_below_64q needs q >= 2^33
    if (B.pm_class && !c->opt.ntt_nopm) {
        // pseudo-Mersenne passes, ONE polynomial per workgroup at every n: with the twiddles of a stage laid out for
        // coalesced loads the pair kernel's shared twiddle fetch buys nothing, and its 128 VGPRs spill (P8192 forward
        // 0.59 against 0.62 ms per 2048 ciphertexts, inverse 0.66 against 0.72)
        const unsigned grid = (unsigned)n_res_polys;
#define GO_PM(CC)                                                                                                                  \
    DISPATCH_L(c->logn, {                                                                                                          \
        if (inverse) k_ntt_inv_pm<L, 1, CC><<<grid, NttShape<L>::TP, 0, st>>>(in, out, base, base.count);                           \
        else k_ntt_fwd_pm<L, 1, CC><<<grid, NttShape<L>::TP, 0, st>>>(in, out, base, base.count);                                   \
    })
        if (B.pm_class == 1) { GO_PM(PmA); }
        else { GO_PM(PmB); }
#undef GO_PM
        KERNEL_CHECK();
        return FHE_OK;
    }
    DISPATCH_L(c->logn, {
        if (inverse && pair && lazy) k_ntt_inv2<L, true><<<(unsigned)(n_res_polys / 2), NttShape<L>::TP, 0, st>>>(in, out, base, base.count);
        else if (inverse && pair) k_ntt_inv2<L, false><<<(unsigned)(n_res_polys / 2), NttShape<L>::TP, 0, st>>>(in, out, base, base.count);
        else if (inverse && lazy) k_ntt_inv<L, true><<<(unsigned)n_res_polys, NttShape<L>::TP, 0, st>>>(in, out, base);
        else if (inverse) k_ntt_inv<L, false><<<(unsigned)n_res_polys, NttShape<L>::TP, 0, st>>>(in, out, base);
        else if (lazy && pair) k_ntt_fwd2<L, true><<<(unsigned)(n_res_polys / 2), NttShape<L>::TP, 0, st>>>(in, out, base, base.count);
        else if (lazy) k_ntt_fwd<L, true><<<(unsigned)n_res_polys, NttShape<L>::TP, 0, st>>>(in, out, base);
        else k_ntt_fwd<L, false><<<(unsigned)n_res_polys, NttShape<L>::TP, 0, st>>>(in, out, base);
    });
    KERNEL_CHECK();
    return FHE_OK;
}

extern "C" int fhe_ntt_forward(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n_polys, fhe_stream s) {
    if (!c || !in || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (n_polys && fhe_rgb_f64_supported(c) && !c->opt.force_u64) return fhe_poly_f64_launch(0, c, (const u64 *)in, (u64 *)out, n_polys, nullptr, (hipStream_t)s);
    return fhe_ntt_launch(false, c, c->qb, (const u64 *)in, (u64 *)out, n_polys * c->k, (hipStream_t)s);
}
extern "C" int fhe_ntt_inverse(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n_polys, fhe_stream s) {
    if (!c || !in || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (n_polys && fhe_rgb_f64_supported(c) && !c->opt.force_u64) return fhe_poly_f64_launch(1, c, (const u64 *)in, (u64 *)out, n_polys, nullptr, (hipStream_t)s);
    return fhe_ntt_launch(true, c, c->qb, (const u64 *)in, (u64 *)out, n_polys * c->k, (hipStream_t)s);
}

extern "C" int fhe_multiply_plain(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n_polys,
                                  const uint64_t *d_plain_ntt, fhe_stream s) {
    if (!c || !in || !out || !d_plain_ntt) return fail(FHE_ERR_PARAM, "null argument");
    const u64 nrp = n_polys * c->k;
    if (nrp == 0) return FHE_OK;
    if (nrp > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    if (fhe_rgb_f64_supported(c) && !c->opt.force_u64)
        return fhe_poly_f64_launch(2, c, (const u64 *)in, (u64 *)out, n_polys, (const ulonglong2 *)d_plain_ntt, (hipStream_t)s);
    const RnsBase base = c->qb.dev();
    hipStream_t st = (hipStream_t)s;
    if (c->qb.pm_class && !c->opt.ntt_nopm) {
#define GO_PM(CC) DISPATCH_L(c->logn, (k_mulplain_pm<L, 1, CC><<<(unsigned)nrp, NttShape<L>::TP, 0, st>>>((const u64 *)in, (u64 *)out, (const ulonglong2 *)d_plain_ntt, base, base.count)))
        if (c->qb.pm_class == 1) { GO_PM(PmA); }
        else { GO_PM(PmB); }
#undef GO_PM
    } else if (c->max_prime_bits <= 58) {
        DISPATCH_L(c->logn, (k_mulplain<L, true><<<(unsigned)nrp, NttShape<L>::TP, 0, st>>>((const u64 *)in, (u64 *)out, (const ulonglong2 *)d_plain_ntt, base)));
    } else {
        DISPATCH_L(c->logn, (k_mulplain<L, false><<<(unsigned)nrp, NttShape<L>::TP, 0, st>>>((const u64 *)in, (u64 *)out, (const ulonglong2 *)d_plain_ntt, base)));
    }
    KERNEL_CHECK();
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// multiply_plain by a sparse plaintext: sum of signed rotations, no transform
// ------------------------------------------------------------------------------------------------
struct SparseTerms {
    u32 count;
    u32 exp[FHE_SPARSE_MAX_TERMS];
    u64 w[FHE_SPARSE_MAX_TERMS][FHE_MAX_K], ws[FHE_SPARSE_MAX_TERMS][FHE_MAX_K];   // lifted coefficient mod q_i, Shoup companion
};
template <int L>
__global__ __launch_bounds__(256) void k_mulplain_sparse(const u64 *__restrict__ in, u64 *__restrict__ out, const Modulus *__restrict__ mods,
                                                         u32 k, const SparseTerms T) {
    constexpr int N = 1 << L;
    __shared__ u64 buf[N];                       // staged so that out may alias in
    const u64 rp = blockIdx.x;
    const u32 prime = (u32)(rp % k);
    const u64 q = mods[prime].q;
    for (int j = threadIdx.x; j < N; j += 256) buf[j] = in[rp * N + j];
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += 256) {
        u64 acc = 0;
        for (u32 t = 0; t < T.count; t++) {
            int idx = j - (int)T.exp[t];
            const bool wrap = idx < 0;            // x^n = -1
            idx += wrap ? N : 0;
            const u64 prod = mul_shoup(buf[idx], T.w[t][prime], T.ws[t][prime], q);
            acc = wrap ? submod(acc, prod, q) : addmod(acc, prod, q);
        }
        out[rp * N + j] = acc;
    }
}

extern "C" int fhe_multiply_plain_sparse(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n_polys,
                                         const uint64_t *plain_host, uint32_t plain_len, fhe_stream s) {
    using namespace hostmath;
    if (!c || !in || !out || !plain_host) return fail(FHE_ERR_PARAM, "null argument");
    if (c->logn > 13) return fail(FHE_ERR_PARAM, "sparse multiply_plain supports n <= 8192");
    SparseTerms T;
    T.count = 0;
    for (u32 j = 0; j < plain_len && j < c->n; ++j) {
        const u64 m = plain_host[j];
        if (!m) continue;
        if (m >= c->t) return fail(FHE_ERR_PARAM, "plaintext coefficient not below the plain modulus");
        if (T.count == FHE_SPARSE_MAX_TERMS) return fail(FHE_ERR_PARAM, "plaintext has more than %d non-zero coefficients", FHE_SPARSE_MAX_TERMS);
        T.exp[T.count] = j;
        for (u32 i = 0; i < c->k; ++i) {
            const u64 qi = c->qb.primes[i];
            const u64 w = m >= c->upper_half_threshold ? (m + c->plain_upper_half_increment[i]) % qi : m % qi;
            T.w[T.count][i] = w;
            T.ws[T.count][i] = shoup(w, qi);
        }
        ++T.count;
    }
    if (!T.count) return fail(FHE_ERR_PARAM, "plain cannot be zero");
    const u64 nrp = n_polys * c->k;
    if (nrp == 0) return FHE_OK;
    if (nrp > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    hipStream_t st = (hipStream_t)s;
    switch (c->logn) {
        case 10: k_mulplain_sparse<10><<<(unsigned)nrp, 256, 0, st>>>((const u64 *)in, (u64 *)out, c->qb.d_mod, c->k, T); break;
        case 11: k_mulplain_sparse<11><<<(unsigned)nrp, 256, 0, st>>>((const u64 *)in, (u64 *)out, c->qb.d_mod, c->k, T); break;
        case 12: k_mulplain_sparse<12><<<(unsigned)nrp, 256, 0, st>>>((const u64 *)in, (u64 *)out, c->qb.d_mod, c->k, T); break;
        default: k_mulplain_sparse<13><<<(unsigned)nrp, 256, 0, st>>>((const u64 *)in, (u64 *)out, c->qb.d_mod, c->k, T); break;
    }
    KERNEL_CHECK();
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// Cubic's linear parts (homo/fhe_resize.h:150-172, 181-188) as single passes
// ------------------------------------------------------------------------------------------------
// x^e * P at coefficient j of a negacyclic polynomial: +-P[j - e]; returns the value to ADD
__device__ __forceinline__ u64 rot_term(const u64 *__restrict__ p, int j, int e, int n, u64 q) {
    const int idx = j - e;
    if (idx >= 0) return p[idx];
    const u64 v = p[idx + n];
    return v ? q - v : 0;
}
__global__ __launch_bounds__(256) void k_cubic_coeffs(const u64 *__restrict__ A, const u64 *__restrict__ B, const u64 *__restrict__ C,
                                                      const u64 *__restrict__ D, u64 *__restrict__ a, u64 *__restrict__ b, u64 *__restrict__ c,
                                                      const Modulus *__restrict__ mods, u32 k, u32 n) {
    const u64 rp = blockIdx.x;
    const u64 q = mods[rp % k].q;
    const u64 *pa = A + rp * n, *pb = B + rp * n, *pc = C + rp * n, *pd = D + rp * n;
    for (int j = threadIdx.x; j < (int)n; j += 256) {
        const u64 Aj = pa[j], Bj = pb[j], Cj = pc[j], Dj = pd[j];
        // a = B (x+1) - A - C (x+1) + D
        u64 va = addmod(Bj, rot_term(pb, j, 1, n, q), q);
        va = submod(va, Aj, q);
        va = submod(va, addmod(Cj, rot_term(pc, j, 1, n, q), q), q);
        va = addmod(va, Dj, q);
        // b = A x - B (x^2+1) + C x^2 - D
        u64 vb = rot_term(pa, j, 1, n, q);
        vb = submod(vb, addmod(Bj, rot_term(pb, j, 2, n, q), q), q);
        vb = addmod(vb, rot_term(pc, j, 2, n, q), q);
        vb = submod(vb, Dj, q);
        a[rp * n + j] = va;
        b[rp * n + j] = vb;
        c[rp * n + j] = submod(Cj, Aj, q);
    }
}
// out = (a + b + c) * (-x^(n-1)) + B:  -x^(n-1) = x^(-1), so coefficient j takes S[j+1], the last one -S[0]
__global__ __launch_bounds__(256) void k_cubic_combine(const u64 *__restrict__ a, const u64 *__restrict__ b, const u64 *__restrict__ c,
                                                       const u64 *__restrict__ B, u64 *__restrict__ out, const Modulus *__restrict__ mods,
                                                       u32 k, u32 n, u32 size_abc, u32 size_b) {
    const u64 rp = blockIdx.x;                       // (ct * size_abc + poly) * k + prime
    const u32 prime = (u32)(rp % k);
    const u64 cp = rp / k;
    const u32 poly = (u32)(cp % size_abc);
    const u64 ct = cp / size_abc;
    const u64 q = mods[prime].q;
    const u64 *pa = a + rp * n, *pb = b + rp * n, *pc = c + rp * n;
    const u64 *pB = poly < size_b ? B + ((ct * size_b + poly) * k + prime) * n : nullptr;
    for (int j = threadIdx.x; j < (int)n; j += 256) {
        const int src = j + 1 < (int)n ? j + 1 : 0;
        u64 s = addmod(addmod(pa[src], pb[src], q), pc[src], q);
        if (j + 1 == (int)n) s = s ? q - s : 0;
        out[rp * n + j] = pB ? addmod(s, pB[j], q) : s;
    }
}

extern "C" int fhe_cubic_coeffs(const fhe_ctx *c, const uint64_t *A, const uint64_t *B, const uint64_t *C, const uint64_t *D,
                                uint64_t *a, uint64_t *b, uint64_t *cc, uint32_t size, uint64_t count, fhe_stream s) {
    if (!c || !A || !B || !C || !D || !a || !b || !cc) return fail(FHE_ERR_PARAM, "null argument");
    const u64 nrp = count * size * c->k;
    if (!nrp) return FHE_OK;
    if (nrp > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    k_cubic_coeffs<<<(unsigned)nrp, 256, 0, (hipStream_t)s>>>((const u64 *)A, (const u64 *)B, (const u64 *)C, (const u64 *)D, (u64 *)a, (u64 *)b,
                                                               (u64 *)cc, c->qb.d_mod, c->k, c->n);
    KERNEL_CHECK();
    return FHE_OK;
}
extern "C" int fhe_cubic_combine(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, const uint64_t *cc, uint32_t size_abc,
                                 const uint64_t *B, uint32_t size_b, uint64_t *out, uint64_t count, fhe_stream s) {
    if (!c || !a || !b || !cc || !B || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (size_b > size_abc) return fail(FHE_ERR_PARAM, "B is larger than the products");
    const u64 nrp = count * size_abc * c->k;
    if (!nrp) return FHE_OK;
    if (nrp > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    k_cubic_combine<<<(unsigned)nrp, 256, 0, (hipStream_t)s>>>((const u64 *)a, (const u64 *)b, (const u64 *)cc, (const u64 *)B, (u64 *)out,
                                                                c->qb.d_mod, c->k, c->n, size_abc, size_b);
    KERNEL_CHECK();
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// plaintext preparation
// ------------------------------------------------------------------------------------------------
// in: [k][n] NTT-form values; out: [k][n] (value, Shoup companion) pairs.  in may alias the first
// half of out only if processed back to front -- callers pass a separate buffer.
__global__ void k_make_shoup(const u64 *__restrict__ in, ulonglong2 *__restrict__ out, const Modulus *__restrict__ mods, u32 n, u32 total) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u64 q = mods[i / n].q;
    const u64 w = in[i];
    out[i] = make_ulonglong2(w, shoup_companion(w, q));
}

extern "C" size_t fhe_plain_ntt_words(const fhe_ctx *c) { return c ? (size_t)2 * c->k * c->n : 0; }

static int lift_plain_host(const fhe_ctx *c, const uint64_t *plain, uint32_t len, std::vector<u64> &out) {
    if (len > c->n) return fail(FHE_ERR_PARAM, "plaintext longer than the polynomial");
    out.assign((size_t)c->k * c->n, 0);
    for (u32 j = 0; j < len; ++j) {
        const u64 m = plain[j];
        if (m >= c->t) return fail(FHE_ERR_PARAM, "plaintext coefficient %u not below the plain modulus", j);
        for (u32 i = 0; i < c->k; ++i) {
            const u64 qi = c->qb.primes[i];
            out[(size_t)i * c->n + j] = m >= c->upper_half_threshold ? (m + c->plain_upper_half_increment[i]) % qi : m % qi;
        }
    }
    return FHE_OK;
}

extern "C" int fhe_plain_prepare(const fhe_ctx *c, const uint64_t *plain, uint32_t len, uint64_t *d_out, fhe_stream s) {
    if (!c || (!plain && len) || !d_out) return fail(FHE_ERR_PARAM, "null argument");
    std::vector<u64> lifted;
    int rc = lift_plain_host(c, plain, len, lifted);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)s;
    const size_t words = (size_t)c->k * c->n;
    u64 *tmp = nullptr;                       // same synchronous staging discipline as fhe_add_plain
    HIP_TRY(hipMalloc((void **)&tmp, 2 * words * sizeof(u64)));
    hipError_t e = hipMemcpy(tmp, lifted.data(), words * sizeof(u64), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = fhe_ntt_launch(false, c, c->qb, tmp, tmp + words, c->k, st);
        if (rc == FHE_OK) {
            k_make_shoup<<<(unsigned)((words + 255) / 256), 256, 0, st>>>(tmp + words, (ulonglong2 *)d_out, c->qb.d_mod, c->n, (u32)words);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return fail(FHE_ERR_HIP, "plain_prepare: %s", hipGetErrorString(e));
    return FHE_OK;
}

__global__ void k_plain_ntt_mul(const ulonglong2 *__restrict__ a, const ulonglong2 *__restrict__ b, ulonglong2 *__restrict__ out,
                                const Modulus *__restrict__ mods, u32 n, u32 total) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Modulus m = mods[i / n];
    const u64 w = mul_barrett(a[i].x, b[i].x, m);
    out[i] = make_ulonglong2(w, shoup_companion(w, m.q));
}
extern "C" int fhe_plain_ntt_mul(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, fhe_stream s) {
    if (!c || !a || !b || !out) return fail(FHE_ERR_PARAM, "null argument");
    const u32 total = c->k * c->n;
    k_plain_ntt_mul<<<(total + 255) / 256, 256, 0, (hipStream_t)s>>>((const ulonglong2 *)a, (const ulonglong2 *)b, (ulonglong2 *)out, c->qb.d_mod, c->n, total);
    KERNEL_CHECK();
    return FHE_OK;
}

// dyadic product of two NTT-form operands
__global__ __launch_bounds__(256) void k_dyadic(const u64 *__restrict__ a, const u64 *__restrict__ b, u64 *__restrict__ out,
                                                const Modulus *__restrict__ mods, u32 k, u32 n, u64 n_res_polys) {
    for (u64 rp = blockIdx.y; rp < n_res_polys; rp += gridDim.y) {
        const Modulus m = mods[rp % k];
        const u64 base = rp * n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
            out[base + i] = mul_barrett(a[base + i], b[base + i], m);
    }
}
extern "C" int fhe_dyadic_multiply(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n_polys, fhe_stream s) {
    if (!c || !a || !b || !out) return fail(FHE_ERR_PARAM, "null argument");
    const u64 nrp = n_polys * c->k;
    if (!nrp) return FHE_OK;
    dim3 grid((c->n + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
    k_dyadic<<<grid, 256, 0, (hipStream_t)s>>>((const u64 *)a, (const u64 *)b, (u64 *)out, c->qb.d_mod, c->k, c->n, nrp);
    KERNEL_CHECK();
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// DCT + quantisation block circuit
// ------------------------------------------------------------------------------------------------
// Constant table: [cid][prime][n] Shoup pairs in slot order.
//   cid 0..11  : the twelve LL&M constants of homo/fhe_image.h:221-236, in order of first use
//   cid 12..75 : per-output scale = encode(0.125) [* encode(1/quant[i])], i = row-major output index
static const double kDctConst[12] = {0.541196100, 0.765366865, -1.847759065, 1.175875602, 0.298631336, 2.053119869,
                                     3.072711026, 1.501321110, -0.899976223, -2.562915447, -1.961570560, -0.390180644};
// One 1-D LL&M pass on eight fully reduced residues (same dataflow as homo/fhe_image.h:207-242).
// C(cid) yields the Shoup pair of constant cid at this thread's slot.
template <typename CF>
__device__ __forceinline__ void dct_line_u64(u64 &d0, u64 &d1, u64 &d2, u64 &d3, u64 &d4, u64 &d5, u64 &d6, u64 &d7, const u64 q, CF C) {
    auto MUL = [&](u64 x, int cid) { const ulonglong2 w = C(cid); return mul_shoup(x, w.x, w.y, q); };
    u64 tmp0 = addmod(d0, d7, q), tmp7 = submod(d0, d7, q);
    u64 tmp1 = addmod(d1, d6, q), tmp6 = submod(d1, d6, q);
    u64 tmp2 = addmod(d2, d5, q), tmp5 = submod(d2, d5, q);
    u64 tmp3 = addmod(d3, d4, q), tmp4 = submod(d3, d4, q);
    const u64 tmp10 = addmod(tmp0, tmp3, q), tmp13 = submod(tmp0, tmp3, q);
    const u64 tmp11 = addmod(tmp1, tmp2, q), tmp12 = submod(tmp1, tmp2, q);
    d0 = addmod(tmp10, tmp11, q);
    d4 = submod(tmp10, tmp11, q);
    u64 z1 = MUL(addmod(tmp12, tmp13, q), 0);
    d2 = addmod(z1, MUL(tmp13, 1), q);
    d6 = addmod(z1, MUL(tmp12, 2), q);
    z1 = addmod(tmp4, tmp7, q);
    u64 z2 = addmod(tmp5, tmp6, q), z3 = addmod(tmp4, tmp6, q), z4 = addmod(tmp5, tmp7, q);
    const u64 z5 = MUL(addmod(z3, z4, q), 3);
    tmp4 = MUL(tmp4, 4);
    tmp5 = MUL(tmp5, 5);
    tmp6 = MUL(tmp6, 6);
    tmp7 = MUL(tmp7, 7);
    z1 = MUL(z1, 8);
    z2 = MUL(z2, 9);
    z3 = addmod(MUL(z3, 10), z5, q);
    z4 = addmod(MUL(z4, 11), z5, q);
    d7 = addmod(addmod(tmp4, z1, q), z3, q);
    d5 = addmod(addmod(tmp5, z2, q), z4, q);
    d3 = addmod(addmod(tmp6, z2, q), z3, q);
    d1 = addmod(addmod(tmp7, z1, q), z4, q);
}

// V1 slot kernel: one thread owns one NTT slot of one (block, poly, prime) unit: 64 values in,
// row pass, column pass, per-output scale, 64 values out (in place).
__global__ __launch_bounds__(256) void k_dct_slots(u64 *__restrict__ data, const ulonglong2 *__restrict__ consts,
                                                   const Modulus *__restrict__ mods, u32 k, u32 n) {
    const u32 unit = blockIdx.y;             // (block * 2 + poly) * k + prime
    const u32 prime = unit % k;
    const u32 bp = unit / k;                 // block * 2 + poly
    const u32 blk = bp >> 1, poly = bp & 1;
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    const u64 q = mods[prime].q;
    const size_t ct_stride = (size_t)2 * k * n;
    u64 *p = data + (size_t)blk * 64 * ct_stride + ((size_t)poly * k + prime) * n + slot;
    const ulonglong2 *cp = consts + (size_t)prime * n + slot;
    const size_t cstride = (size_t)k * n;
    auto C = [&](int cid) { return cp[(size_t)cid * cstride]; };
    u64 v[64];
#pragma unroll
    for (int i = 0; i < 64; i++) v[i] = p[(size_t)i * ct_stride];
#pragma unroll
    for (int r = 0; r < 8; r++)
        dct_line_u64(v[8 * r], v[8 * r + 1], v[8 * r + 2], v[8 * r + 3], v[8 * r + 4], v[8 * r + 5], v[8 * r + 6], v[8 * r + 7], q, C);
#pragma unroll
    for (int col = 0; col < 8; col++)
        dct_line_u64(v[col], v[col + 8], v[col + 16], v[col + 24], v[col + 32], v[col + 40], v[col + 48], v[col + 56], q, C);
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const ulonglong2 w = C(12 + i);
        p[(size_t)i * ct_stride] = mul_shoup(v[i], w.x, w.y, q);
    }
}

// Pseudo-Mersenne bases (every prime <= 55 bits, class PmA of ntt_core.h) take the slot step as TWO launches of 8 values
// per thread -- the rows of a block, then its columns with the per-output scale -- on lazy arithmetic: sums and differences
// stay unreduced (a difference gets a power-of-two multiple of q above its subtrahend's bound added), every product is
// mulvv_pm(fold_pm(x), c) -- any 64-bit x in, below 6q out --, row outputs are stored as they are (below 24 q) and only the
// 64 final values are brought to canonical form.  The one-launch kernel above holds 64 values per thread (256 VGPRs, one wave
// per SIMD) and moves 1.5 TB/s; its lazy form spills (202 VGPRs + 528 B).  I = bound of a line's inputs in units of q; the
// largest value met is below 256 q < 2^63.
__host__ __device__ constexpr int dct_pow2_at_least(int v) { int p = 1; while (p < v) p <<= 1; return p; }
template <int I, typename CF>
__device__ __forceinline__ void dct_line_pm(u64 (&d)[8], const PmMod &m, CF C) {
    constexpr int IP = dct_pow2_at_least(I), S1 = dct_pow2_at_least(2 * I), S2 = dct_pow2_at_least(4 * I);
    static_assert(4 * I + S2 <= 256 && 2 * (2 * I + S1) <= 256 && 4 * (I + IP) <= 256, "a sum would pass 256 q");
    const u64 oi = m.q * IP, o1 = m.q * S1, o2 = m.q * S2;
    auto MUL = [&](u64 x, int cid) { return mulvv_pm(fold_pm(x, m), C(cid).x, m); };
    u64 tmp0 = d[0] + d[7], tmp7 = d[0] - d[7] + oi;
    u64 tmp1 = d[1] + d[6], tmp6 = d[1] - d[6] + oi;
    u64 tmp2 = d[2] + d[5], tmp5 = d[2] - d[5] + oi;
    u64 tmp3 = d[3] + d[4], tmp4 = d[3] - d[4] + oi;
    const u64 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3 + o1;
    const u64 tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2 + o1;
    d[0] = tmp10 + tmp11;
    d[4] = tmp10 - tmp11 + o2;
    u64 z1 = MUL(tmp12 + tmp13, 0);
    d[2] = z1 + MUL(tmp13, 1);
    d[6] = z1 + MUL(tmp12, 2);
    z1 = tmp4 + tmp7;
    u64 z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    const u64 z5 = MUL(z3 + z4, 3);
    tmp4 = MUL(tmp4, 4);
    tmp5 = MUL(tmp5, 5);
    tmp6 = MUL(tmp6, 6);
    tmp7 = MUL(tmp7, 7);
    z1 = MUL(z1, 8);
    z2 = MUL(z2, 9);
    z3 = MUL(z3, 10) + z5;
    z4 = MUL(z4, 11) + z5;
    d[7] = tmp4 + z1 + z3;
    d[5] = tmp5 + z2 + z4;
    d[3] = tmp6 + z2 + z3;
    d[1] = tmp7 + z1 + z4;
}
// COLS = false: line `l` = row l of the block (ciphertexts 8 l .. 8 l + 7); true: column l (ciphertexts l, l + 8, ...) + scale
template <bool COLS>
__global__ __launch_bounds__(256) void k_dct_lines_pm(u64 *__restrict__ data, const ulonglong2 *__restrict__ consts,
                                                      const PmMod *__restrict__ pm, u32 k, u32 n) {
    const u32 line = blockIdx.y & 7, unit = blockIdx.y >> 3;      // unit = (block * 2 + poly) * k + prime
    const u32 prime = unit % k;
    const u32 bp = unit / k;
    const u32 blk = bp >> 1, poly = bp & 1;
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    const PmMod m = pm[prime];
    const size_t ct_stride = (size_t)2 * k * n, step = COLS ? 8 * ct_stride : ct_stride;
    u64 *p = data + ((size_t)blk * 64 + (COLS ? line : 8 * line)) * ct_stride + ((size_t)poly * k + prime) * n + slot;
    const ulonglong2 *cp = consts + (size_t)prime * n + slot;
    const size_t cstride = (size_t)k * n;
    auto C = [&](int cid) { return cp[(size_t)cid * cstride]; };
    u64 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = p[(size_t)i * step];
    if constexpr (!COLS) {
        dct_line_pm<1>(v, m, C);
#pragma unroll
        for (int i = 0; i < 8; i++) p[(size_t)i * step] = v[i];
    } else {
        dct_line_pm<24>(v, m, C);
#pragma unroll
        for (int i = 0; i < 8; i++) p[(size_t)i * step] = canon_pm(mulvv_pm(fold_pm(v[i], m), C(12 + 8 * i + line).x, m), m);
    }
}

extern "C" int fhe_dct_plan_create(const fhe_ctx *c, const double *quant64, int int_coeffs, int frac_coeffs, fhe_stream s, fhe_dct_plan **out) {
    if (!c || !out) return fail(FHE_ERR_PARAM, "null argument");
    *out = nullptr;
    fhe_dct_plan *p = new fhe_dct_plan();
    p->k = c->k;
    p->n = c->n;
    p->has_quant = quant64 != nullptr;
    const size_t pw = (size_t)c->k * c->n;   // pairs per constant
    int rc = fhe_dev_alloc(sizeof(ulonglong2) * pw * DCT_NCONST, (void **)&p->d_consts);
    if (rc) { delete p; return rc; }
    std::vector<uint64_t> plain(c->n);
    ulonglong2 *d_eighth = nullptr, *d_tmp = nullptr;
    auto cleanup = [&](int code) {
        if (d_eighth) (void)hipFree(d_eighth);
        if (d_tmp) (void)hipFree(d_tmp);
        if (code) { (void)hipFree(p->d_consts); if (p->d_consts_f64) (void)hipFree(p->d_consts_f64); if (p->d_consts_le3) (void)hipFree(p->d_consts_le3); delete p; }
        return code;
    };
    auto prep = [&](double v, ulonglong2 *dst) -> int {
        int len = fhe_frac_encode(c->n, c->t, v, int_coeffs, frac_coeffs, plain.data());
        if (len < 0) return len;
        return fhe_plain_prepare(c, plain.data(), (uint32_t)len, (uint64_t *)dst, s);
    };
    for (int i = 0; i < 12; ++i)
        if ((rc = prep(kDctConst[i], p->d_consts + pw * i))) return cleanup(rc);
    if (!quant64) {
        for (int i = 0; i < 64; ++i)
            if ((rc = prep(0.125, p->d_consts + pw * (12 + i)))) return cleanup(rc);
    } else {
        if ((rc = fhe_dev_alloc(sizeof(ulonglong2) * pw, (void **)&d_eighth))) return cleanup(rc);
        if ((rc = fhe_dev_alloc(sizeof(ulonglong2) * pw, (void **)&d_tmp))) return cleanup(rc);
        if ((rc = prep(0.125, d_eighth))) return cleanup(rc);
        for (int i = 0; i < 64; ++i) {
            if (!(quant64[i] != 0.0)) return cleanup(fail(FHE_ERR_PARAM, "quant[%d] is zero", i));
            if ((rc = prep(1 / quant64[i], d_tmp))) return cleanup(rc);
            if ((rc = fhe_plain_ntt_mul(c, (const uint64_t *)d_eighth, (const uint64_t *)d_tmp, (uint64_t *)(p->d_consts + pw * (12 + i)), s)))
                return cleanup(rc);
        }
    }
    if (fhe_dct_f64_supported(c) && (rc = fhe_dct_f64_make_consts(c, p, (hipStream_t)s))) return cleanup(rc);
    if (!fhe_dct_f64_supported(c) && fhe_dct_u64_supported(c) && (rc = fhe_dct_u64_make_consts(c, p, (hipStream_t)s))) return cleanup(rc);
    if (hipStreamSynchronize((hipStream_t)s) != hipSuccess) return cleanup(fail(FHE_ERR_HIP, "stream sync failed"));
    *out = p;
    return cleanup(FHE_OK);
}
extern "C" int fhe_dct_plan_destroy(fhe_dct_plan *p) {
    if (!p) return FHE_OK;
    if (p->d_consts) (void)hipFree(p->d_consts);
    if (p->d_consts_f64) (void)hipFree(p->d_consts_f64);
    if (p->d_consts_le3) (void)hipFree(p->d_consts_le3);
    delete p;
    return FHE_OK;
}
// The fused path keeps one row-transformed copy of a wave of blocks between its two kernels
// (12 MiB per block at n=4096, k=3).  Measured: 32 blocks 74.4 k blocks/s, 64: 76.7 k, 128: 78.3 k,
// 256: 79.1 k, 512: 79.0 k (launch tails amortise; Infinity Cache residency of the copy does not pay).
static u64 dct_wave_blocks(const fhe_ctx *c) { return c->opt.dct_wave_blocks; }
extern "C" size_t fhe_dct8x8_scratch_bytes(const fhe_ctx *c, uint64_t n_blocks) {
    if (!c || !(fhe_dct_f64_supported(c) || fhe_dct_u64_supported(c))) return 0;
    const u64 wave = dct_wave_blocks(c) < n_blocks ? dct_wave_blocks(c) : n_blocks;
    return (size_t)wave * 64 * 2 * c->k * c->n * sizeof(double);
}

extern "C" int fhe_dct_path(const fhe_ctx *c) {
    if (!c) return fail(FHE_ERR_PARAM, "null argument");
    if (fhe_dct_f64_supported(c) && !c->opt.force_u64) return 1;
    if (fhe_dct_u64_supported(c) && !c->opt.force_u64) return 2;
    return 0;
}

extern "C" int fhe_dct8x8_quant(const fhe_ctx *c, const fhe_dct_plan *plan, const uint64_t *in, uint64_t *out, uint64_t n_blocks,
                                void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!c || !plan || !in || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (plan->k != c->k || plan->n != c->n) return fail(FHE_ERR_PARAM, "plan was built for another context");
    if (n_blocks == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)s;
    if (plan->d_consts_f64 && fhe_dct_f64_supported(c) && !c->opt.force_u64) {
        const size_t per_block = (size_t)64 * 2 * c->k * c->n;
        const u64 fit = scratch ? scratch_bytes / (per_block * sizeof(double)) : 0;
        if (fit == 0) return fail(FHE_ERR_PARAM, "scratch too small: need fhe_dct8x8_scratch_bytes()");
        u64 wave = fit < dct_wave_blocks(c) ? fit : dct_wave_blocks(c);
        // measured: 68.1k blocks/s pipelined vs 71.8k plain at 64-block waves, so this is opt-in
        const bool pipelined = c->opt.dct_pipeline && fit >= 2 && n_blocks > wave / 2 && wave >= 2;
        if (!pipelined) {
            for (u64 b0 = 0; b0 < n_blocks; b0 += wave) {
                const u64 nb = (n_blocks - b0) < wave ? (n_blocks - b0) : wave;
                int rc = fhe_dct_f64_launch(c, plan, (const u64 *)in + b0 * per_block, (u64 *)out + b0 * per_block, nb, (double *)scratch, st);
                if (rc) return rc;
            }
            return FHE_OK;
        }
        // Two half-size intermediates: the column kernel of wave w runs on a second stream while the
        // row kernel of wave w+1 runs on the caller's stream, so workgroups of both kinds share the CUs
        // (row work is FP64-issue heavy, column work waits more on memory) and launch tails overlap.
        const fhe_ctx *mc = c;      // stream and events belong to the context: one pipelined call per context at a time
        wave = wave / 2 < 1 ? 1 : wave / 2;
        double *midbuf[2] = {(double *)scratch, (double *)scratch + wave * per_block};
        u64 w = 0;
        for (u64 b0 = 0; b0 < n_blocks; b0 += wave, ++w) {
            const u64 nb = (n_blocks - b0) < wave ? (n_blocks - b0) : wave;
            const int s = (int)(w & 1);
            if (w >= 2) HIP_TRY(hipStreamWaitEvent(st, mc->ev_cols[s], 0));            // mid[s] is free again
            int rc = fhe_dct_f64_launch(c, plan, (const u64 *)in + b0 * per_block, (u64 *)out + b0 * per_block, nb, midbuf[s], st, 1);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(mc->ev_rows[s], st));
            HIP_TRY(hipStreamWaitEvent(mc->aux_stream, mc->ev_rows[s], 0));
            rc = fhe_dct_f64_launch(c, plan, (const u64 *)in + b0 * per_block, (u64 *)out + b0 * per_block, nb, midbuf[s], mc->aux_stream, 2);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(mc->ev_cols[s], mc->aux_stream));
        }
        for (int s = 0; s < 2 && (u64)s < w; ++s) HIP_TRY(hipStreamWaitEvent(st, mc->ev_cols[s], 0));   // join: out is complete on `st`
        return FHE_OK;
    }
    // primes of 48..57 bits (SEAL 2.3's own coeff_modulus_128 tables): the same two-launch structure in u64 Shoup arithmetic
    if (plan->d_consts_le3 && fhe_dct_u64_supported(c) && !c->opt.force_u64) {
        const size_t per_block = (size_t)64 * 2 * c->k * c->n;
        const u64 fit = scratch ? scratch_bytes / (per_block * sizeof(u64)) : 0;
        if (fit == 0) return fail(FHE_ERR_PARAM, "scratch too small: need fhe_dct8x8_scratch_bytes()");
        const u64 wave = fit < dct_wave_blocks(c) ? fit : dct_wave_blocks(c);
        for (u64 b0 = 0; b0 < n_blocks; b0 += wave) {
            const u64 nb = (n_blocks - b0) < wave ? (n_blocks - b0) : wave;
            int rc = fhe_dct_u64_launch(c, plan, (const u64 *)in + b0 * per_block, (u64 *)out + b0 * per_block, nb, (u64 *)scratch, st);
            if (rc) return rc;
        }
        return FHE_OK;
    }
    // general path (any prime below 2^61): three launches per chunk, u64 Shoup arithmetic
    const u64 polys_per_block = 64 * 2;   // RNS polynomials (of k residues) per block
    const u64 max_blocks = 4096;
    for (u64 b0 = 0; b0 < n_blocks; b0 += max_blocks) {
        const u64 nb = (n_blocks - b0) < max_blocks ? (n_blocks - b0) : max_blocks;
        const size_t off = (size_t)b0 * polys_per_block * c->k * c->n;
        int rc = fhe_ntt_launch(false, c, c->qb, (const u64 *)in + off, (u64 *)out + off, nb * polys_per_block * c->k, st);
        if (rc) return rc;
        dim3 grid(c->n / 256, (unsigned)(nb * 2 * c->k));
        if (c->qb.pm_class == 1 && !c->opt.ntt_nopm) {
            dim3 grid8(c->n / 256, (unsigned)(nb * 2 * c->k * 8));
            k_dct_lines_pm<false><<<grid8, 256, 0, st>>>((u64 *)out + off, plan->d_consts, c->qb.d_pm, c->k, c->n);
            k_dct_lines_pm<true><<<grid8, 256, 0, st>>>((u64 *)out + off, plan->d_consts, c->qb.d_pm, c->k, c->n);
        } else {
            k_dct_slots<<<grid, 256, 0, st>>>((u64 *)out + off, plan->d_consts, c->qb.d_mod, c->k, c->n);
        }
        KERNEL_CHECK();
        rc = fhe_ntt_launch(true, c, c->qb, (const u64 *)out + off, (u64 *)out + off, nb * polys_per_block * c->k, st);
        if (rc) return rc;
    }
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// rgb_to_ycc_fhe: 9 multiply_plain + adds (homo/fhe_image.h:310-325), fused per residue polynomial
// ------------------------------------------------------------------------------------------------
// consts: [9][k][n] Shoup pairs (0.299, 0.587, 0.114, -0.168736, 0.331264, 0.5, 0.5, 0.418688, 0.081312);
// y_off: [k][len] = Delta * encode(128.0) lifted, subtracted from poly 0 of Y.
template <int L>
__global__ __launch_bounds__(NttShape<L>::TP) void k_rgb2ycc(u64 *__restrict__ R, u64 *__restrict__ G, u64 *__restrict__ Bc,
                                                              const ulonglong2 *__restrict__ consts, const u64 *__restrict__ yoff, u32 yoff_len,
                                                              RnsBase base, u32 group, u64 gstride) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N, TP = NttShape<L>::TP;
    const int tid = threadIdx.x;
    const u64 rp0 = blockIdx.x;                // (pixel * 2 + poly) * k + prime
    const u32 prime = (u32)(rp0 % base.count);
    const u32 poly = (u32)((rp0 / base.count) & 1);
    // pixel of a plane: contiguous (group == 0), or `group` pixels every gstride words (fhe_rgb_to_ycc_blocks); rp is the
    // residue-polynomial index inside the plane in units of N words
    const u64 pix = rp0 / (2 * base.count);
    const u64 rp = group ? ((pix / group) * gstride + (pix % group) * 2 * base.count * N) / N + (u64)poly * base.count + prime : rp0;
    const u64 q = base.mod[prime].q;
    const ulonglong2 *tw = base.tw + (size_t)prime * N, *itw = base.itw + (size_t)prime * N;
    const size_t cstride = (size_t)base.count * N;
    const ulonglong2 *cp = consts + (size_t)prime * N;
    u64 r[16], g[16], b[16];
    load_coeff<L>(r, R + rp * N, tid);
    ntt_fwd_regs<L>(r, tw, q, lds, tid);
    load_coeff<L>(g, G + rp * N, tid);
    ntt_fwd_regs<L>(g, tw, q, lds, tid);
    load_coeff<L>(b, Bc + rp * N, tid);
    ntt_fwd_regs<L>(b, tw, q, lds, tid);
    u64 y[16], u[16], v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int pos = i * TP + tid;
        auto M = [&](u64 x, int cid) { const ulonglong2 w = cp[cid * cstride + pos]; return mul_shoup(x, w.x, w.y, q); };
        y[i] = addmod(addmod(M(r[i], 0), M(g[i], 1), q), M(b[i], 2), q);
        u[i] = addmod(submod(M(r[i], 3), M(g[i], 4), q), M(b[i], 5), q);
        v[i] = submod(submod(M(r[i], 6), M(g[i], 7), q), M(b[i], 8), q);
    }
    ntt_inv_regs<L>(y, itw, q, lds, tid);
    ntt_lds_release();                         // the inverse transform's last transpose reads across waves (ntt_core.h, CONTRACT)
    ntt_inv_regs<L>(u, itw, q, lds, tid);
    ntt_lds_release();
    ntt_inv_regs<L>(v, itw, q, lds, tid);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        y[i] = csub(y[i], q);
        u[i] = csub(u[i], q);
        v[i] = csub(v[i], q);
        if (poly == 0) {
            const int j = elem_index<L - 4>(tid, i);
            if ((u32)j < yoff_len) y[i] = submod(y[i], yoff[(size_t)prime * yoff_len + j], q);
        }
    }
    store_coeff<L>(y, R + rp * N, tid);
    store_coeff<L>(u, G + rp * N, tid);
    store_coeff<L>(v, Bc + rp * N, tid);
}

// Encode, lift and transform the nine factors once per context (synchronous, first call only).
static int rgb_consts(const fhe_ctx *c, int int_coeffs, int frac_coeffs, hipStream_t st, const fhe_ctx::RgbConsts **out) {
    using namespace hostmath;
    std::lock_guard<std::mutex> lock(c->rgb_mutex);
    for (const fhe_ctx::RgbConsts *r : c->rgb)
        if (r->int_coeffs == int_coeffs && r->frac_coeffs == frac_coeffs) { *out = r; return FHE_OK; }
    static const double cc[9] = {0.299, 0.587, 0.114, -0.168736, 0.331264, 0.5, 0.5, 0.418688, 0.081312};
    const size_t pw = (size_t)c->k * c->n;
    std::vector<uint64_t> plain(c->n);
    fhe_ctx::RgbConsts t;
    auto drop = [&](int code) { if (t.d_c) (void)hipFree(t.d_c); if (t.d_c_f64) (void)hipFree(t.d_c_f64); if (t.d_off) (void)hipFree(t.d_off); return code; };
    int rc = fhe_dev_alloc(sizeof(ulonglong2) * pw * 9, (void **)&t.d_c);
    if (rc) return rc;
    for (int i = 0; i < 9; ++i) {
        int len = fhe_frac_encode(c->n, c->t, cc[i], int_coeffs, frac_coeffs, plain.data());
        if (len < 0) return drop(len);
        if ((rc = fhe_plain_prepare(c, plain.data(), (uint32_t)len, (uint64_t *)(t.d_c + pw * i), st))) return drop(rc);
    }
    int len = fhe_frac_encode(c->n, c->t, 128.0, int_coeffs, frac_coeffs, plain.data());
    if (len < 0) return drop(len);
    std::vector<u64> off((size_t)c->k * len);
    for (u32 i = 0; i < c->k; ++i)
        for (int j = 0; j < len; ++j) {
            const u64 qi = c->qb.primes[i], m = plain[j];
            u64 v = mulmod(c->delta_mod[i], m % qi, qi);
            if (m >= c->upper_half_threshold) v = addmod(v, c->upper_half_increment[i], qi);
            off[(size_t)i * len + j] = v;
        }
    if ((rc = fhe_dev_alloc(off.size() * sizeof(u64) + 8, (void **)&t.d_off))) return drop(rc);
    if (hipMemcpy(t.d_off, off.data(), off.size() * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess) return drop(fail(FHE_ERR_HIP, "upload failed"));
    t.off_len = (u32)len;
    if (fhe_rgb_f64_supported(c) && (rc = fhe_rgb_f64_make_consts(c, t.d_c, &t.d_c_f64, st))) return drop(rc);
    if (hipStreamSynchronize(st) != hipSuccess) return drop(fail(FHE_ERR_HIP, "stream sync failed"));
    t.int_coeffs = int_coeffs;
    t.frac_coeffs = frac_coeffs;
    c->rgb.push_back(new fhe_ctx::RgbConsts(t));
    *out = c->rgb.back();
    return FHE_OK;
}

// Pseudo-Mersenne bases (C = the class of the q-base) take THREE launches instead: the one-launch kernel above keeps three
// polynomials per thread (96 data VGPRs beside twiddles and constants: 256 VGPRs, one or two waves per SIMD, and at
// n = 8192 another 640 B of scratch per lane) and runs at 0.08-0.11 of the HBM roofline; the transforms alone run at
// 0.4, so the two extra passes over the planes cost less than that kernel loses.  plane_rp: residue polynomial of a plane
// in units of N words -- contiguous (group == 0), or `group` pixels every gstride words (fhe_rgb_to_ycc_blocks).
__device__ __forceinline__ u64 plane_rp(u64 rp0, u32 count, u32 group, u64 gstride, u32 n) {
    if (!group) return rp0;
    const u64 pix = rp0 / (2 * count), rest = rp0 % (2 * count);
    return ((pix / group) * gstride + (pix % group) * 2 * count * n) / n + rest;
}
constexpr int rgb_sum_bound(int rq) { return rq + 2 * (16 << pm_ceil_log2_q(rq)); }      // bound of the y / u / v sums below
template <int L, typename C, bool INVERSE>
__global__ __launch_bounds__(NttShape<L>::TP, 4) void k_rgb_ntt_pm(u64 *__restrict__ R, u64 *__restrict__ G, u64 *__restrict__ Bc,
                                                                   const u64 *__restrict__ yoff, u32 yoff_len, RnsBase base, u32 group, u64 gstride, u32 nrp) {
    __shared__ u64 lds[NttShape<L>::LDS_WORDS];
    constexpr int N = NttShape<L>::N;
    const int tid = threadIdx.x;
    const u32 plane = blockIdx.x / nrp, rp0 = blockIdx.x % nrp;
    const u32 prime = rp0 % base.count, poly = (rp0 / base.count) & 1;
    u64 *p = (plane == 0 ? R : plane == 1 ? G : Bc) + plane_rp(rp0, base.count, group, gstride, N) * N;
    const PmMod m = base.pm[prime];
    u64 x[1][16];
    if constexpr (!INVERSE) {
        load_coeff<L>(x[0], p, tid);
        ntt_fwd_regs_pm<L, 1, 16, C::LIM, C::CS>(x, base.tw_pm + (size_t)prime * N, m, lds, tid);
        store_slots<L>(x[0], p, tid);          // as they are (below 2^62): the combine step folds what it reads
    } else {
        load_slots<L>(x[0], p, tid);
        ntt_inv_regs_pm<L, 1, rgb_sum_bound(C::RQ), C::XB, C::LIM, C::RQ>(x, base.itw_pm + (size_t)prime * N, m, lds, tid);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            x[0][i] = canon_pm(x[0][i], m);
            if (plane == 0 && poly == 0) {
                const int j = elem_index<L - 4>(tid, i);
                if ((u32)j < yoff_len) x[0][i] = submod(x[0][i], yoff[(size_t)prime * yoff_len + j], m.q);
            }
        }
        store_coeff<L>(x[0], p, tid);
    }
}
// y, u, v from r, g, b slot by slot (NTT form, in place): nine products with the constants' values, sums left unreduced
template <typename C>
__global__ __launch_bounds__(256) void k_rgb_combine_pm(u64 *__restrict__ R, u64 *__restrict__ G, u64 *__restrict__ Bc, const ulonglong2 *__restrict__ consts,
                                                        RnsBase base, u32 n, u32 group, u64 gstride, u32 nrp) {
    const size_t cstride = (size_t)base.count * n;
    for (u32 rp0 = blockIdx.y; rp0 < nrp; rp0 += gridDim.y) {
        const u32 prime = rp0 % base.count;
        const PmMod m = base.pm[prime];
        const u64 off = m.q << pm_ceil_log2_q(C::RQ);         // a multiple of q above any product
        const u64 base_w = plane_rp(rp0, base.count, group, gstride, n) * n;
        for (u32 pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += gridDim.x * blockDim.x) {
            const ulonglong2 *cp = consts + (size_t)prime * n + pos;
            const u64 rr = fold_pm(R[base_w + pos], m), gg = fold_pm(G[base_w + pos], m), bb = fold_pm(Bc[base_w + pos], m);
            auto M = [&](u64 x, int cid) { return mulvv_pm(x, cp[cid * cstride].x, m); };
            R[base_w + pos] = M(rr, 0) + M(gg, 1) + M(bb, 2);
            G[base_w + pos] = M(rr, 3) + M(bb, 5) + (off - M(gg, 4));
            Bc[base_w + pos] = M(rr, 6) + (off - M(gg, 7)) + (off - M(bb, 8));
        }
    }
}
template <typename C>
static int rgb_launch_pm(const fhe_ctx *c, u64 *r, u64 *g, u64 *b, const ulonglong2 *consts, const u64 *yoff, u32 yoff_len, u64 nrp, hipStream_t st, u32 group, u64 gstride) {
    if (3 * nrp > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many pixels for one launch");
    const RnsBase base = c->qb.dev();
    DISPATCH_L(c->logn, (k_rgb_ntt_pm<L, C, false><<<(unsigned)(3 * nrp), NttShape<L>::TP, 0, st>>>(r, g, b, yoff, yoff_len, base, group, gstride, (u32)nrp)));
    k_rgb_combine_pm<C><<<dim3((c->n + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768)), 256, 0, st>>>(r, g, b, consts, base, c->n, group, gstride, (u32)nrp);
    DISPATCH_L(c->logn, (k_rgb_ntt_pm<L, C, true><<<(unsigned)(3 * nrp), NttShape<L>::TP, 0, st>>>(r, g, b, yoff, yoff_len, base, group, gstride, (u32)nrp)));
    KERNEL_CHECK();
    return FHE_OK;
}

static int rgb_launch(const fhe_ctx *c, u64 *r, u64 *g, u64 *b, uint64_t count, int int_coeffs, int frac_coeffs, hipStream_t st, u32 group, u64 gstride) {
    const fhe_ctx::RgbConsts *k9 = nullptr;
    int rc = rgb_consts(c, int_coeffs, frac_coeffs, st, &k9);
    if (rc) return rc;
    if (k9->d_c_f64 && !c->opt.force_u64)
        return fhe_rgb_f64_launch(c, r, g, b, count, k9->d_c_f64, k9->d_off, k9->off_len, st, group, gstride);
    const u64 nrp = count * 2 * c->k;
    if (nrp > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many pixels for one launch");
    const RnsBase base = c->qb.dev();
    if (c->qb.pm_class == 1 && !c->opt.ntt_nopm) return rgb_launch_pm<PmA>(c, r, g, b, k9->d_c, k9->d_off, k9->off_len, nrp, st, group, gstride);
    if (c->qb.pm_class == 2 && !c->opt.ntt_nopm) return rgb_launch_pm<PmB>(c, r, g, b, k9->d_c, k9->d_off, k9->off_len, nrp, st, group, gstride);
    DISPATCH_L(c->logn, (k_rgb2ycc<L><<<(unsigned)nrp, NttShape<L>::TP, 0, st>>>(r, g, b, k9->d_c, k9->d_off, k9->off_len, base, group, gstride)));
    KERNEL_CHECK();
    return FHE_OK;
}
extern "C" int fhe_rgb_to_ycc(const fhe_ctx *c, uint64_t *r, uint64_t *g, uint64_t *b, uint64_t count, int int_coeffs, int frac_coeffs, fhe_stream s) {
    if (!c || !r || !g || !b) return fail(FHE_ERR_PARAM, "null argument");
    if (!count) return FHE_OK;
    return rgb_launch(c, (u64 *)r, (u64 *)g, (u64 *)b, count, int_coeffs, frac_coeffs, (hipStream_t)s, 0, 0);
}
// the same on the layout of the ciphertext streams: blocks [n_blocks][R G B][64][2][k][n], in place -> [n_blocks][Y Cb Cr][64]...
extern "C" int fhe_rgb_to_ycc_blocks(const fhe_ctx *c, uint64_t *blocks, uint64_t n_blocks, int int_coeffs, int frac_coeffs, fhe_stream s) {
    if (!c || !blocks) return fail(FHE_ERR_PARAM, "null argument");
    if (!n_blocks) return FHE_OK;
    const u64 plane = (u64)64 * 2 * c->k * c->n;           // words of one channel of one block
    return rgb_launch(c, (u64 *)blocks, (u64 *)blocks + plane, (u64 *)blocks + 2 * plane, n_blocks * 64, int_coeffs, frac_coeffs, (hipStream_t)s, 64, 3 * plane);
}

// ------------------------------------------------------------------------------------------------
// synthetic inputs, digests
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill_random(u64 *__restrict__ out, const Modulus *__restrict__ mods, u32 k, u32 n,
                                                     u64 n_res_polys, u64 seed, u64 first) {
    for (u64 rp = blockIdx.y; rp < n_res_polys; rp += gridDim.y) {
        const u64 q = mods[rp % k].q;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const u64 idx = rp * n + i;
            out[idx] = splitmix64(seed ^ (first + idx)) % q;
        }
    }
}
extern "C" int fhe_fill_random(const fhe_ctx *c, uint64_t *ct, uint64_t n_polys, uint64_t seed, uint64_t first, fhe_stream s) {
    if (!c || !ct) return fail(FHE_ERR_PARAM, "null argument");
    const u64 nrp = n_polys * c->k;
    if (!nrp) return FHE_OK;
    dim3 grid((c->n + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
    k_fill_random<<<grid, 256, 0, (hipStream_t)s>>>((u64 *)ct, c->qb.d_mod, c->k, c->n, nrp, seed, first);
    KERNEL_CHECK();
    return FHE_OK;
}

__global__ __launch_bounds__(256) void k_digest(const u64 *__restrict__ data, u64 count, u64 index0, u64 *out) {
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (u64)gridDim.x * blockDim.x)
        acc += splitmix64(data[i] ^ splitmix64(index0 + i));
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
extern "C" int fhe_digest(const fhe_ctx *, const uint64_t *data, uint64_t count, uint64_t index0, uint64_t *d_out, fhe_stream s) {
    if (!data || !d_out) return fail(FHE_ERR_PARAM, "null argument");
    hipStream_t st = (hipStream_t)s;
    HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(u64), st));
    if (!count) return FHE_OK;
    u64 blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    k_digest<<<(unsigned)blocks, 256, 0, st>>>((const u64 *)data, count, index0, (u64 *)d_out);
    KERNEL_CHECK();
    return FHE_OK;
}

// residues at or above their modulus, counted: what seal::Ciphertext::load + is_valid_for reject one ciphertext at a time, for a
// whole wave of a stream in one pass (the kernels assume canonical residues: the pseudo-Mersenne products take x < 2^62, the
// FP64 path values below 2^52).  16 bytes per lane, one atomic per wave that saw a bad word.
__global__ __launch_bounds__(256) void k_count_unreduced(const ulonglong2 *__restrict__ data, const Modulus *__restrict__ mods, u32 k, u32 half_n,
                                                         u64 n_res_polys, u64 *out) {
    u64 bad = 0;
    for (u64 rp = blockIdx.y; rp < n_res_polys; rp += gridDim.y) {
        const u64 q = mods[rp % k].q;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < half_n; i += gridDim.x * blockDim.x) {
            const ulonglong2 v = data[rp * half_n + i];
            bad += (v.x >= q) + (v.y >= q);
        }
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, bad);
}
extern "C" int fhe_count_unreduced(const fhe_ctx *c, const uint64_t *ct, uint64_t n_polys, uint64_t *d_count, fhe_stream s) {
    if (!c || !ct || !d_count) return fail(FHE_ERR_PARAM, "null argument");
    const u64 nrp = n_polys * c->k;
    if (!nrp) return FHE_OK;
    dim3 grid((c->n / 2 + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
    k_count_unreduced<<<grid, 256, 0, (hipStream_t)s>>>((const ulonglong2 *)ct, c->qb.d_mod, c->k, c->n / 2, nrp, (u64 *)d_count);
    KERNEL_CHECK();
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// ct x ct and relinearisation live in behz.hip
// ------------------------------------------------------------------------------------------------
