/* TEST INFRASTRUCTURE ONLY -- never loaded, linked or imported by the package or by bench.py's timed path.
 *
 * The entry points of include/fhe_hip.h implemented on the CPU oracle (fhe_oracle.c), with "device"
 * memory = malloc.  Purpose: pin the ORACLE against the reference's own golden outputs without a GPU.
 * oracle/Makefile (target `ref`) links the reference's unmodified homo/client_jpeg.cpp and
 * homo/server_jpeg.cpp against seal/seal.h + this library (oracle/_ref/ref_*_cpu); run on the
 * reference's benchmark image they must print the RMSError values published in the reference's
 * benchmark/results.txt (tests/test_oracle_golden.py, tests/golden/make_golden.py).  The same library
 * lets seal/facade_test.cpp exercise the facade's host logic on a machine without a GPU.
 *
 * The product library (csrc/, libfhe_hip.so) has no CPU fallback and shares no code with this file.
 * Plaintext / key material in "NTT form" uses the oracle's own slot order; it is opaque to callers,
 * exactly like the product's (include/fhe_hip.h, layout note).
 */
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fhe_hip.h"
#include "fhe_oracle.h"

typedef unsigned __int128 u128;

static __thread char g_err[256];
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

struct fhe_ctx {
    fo_ctx *o;
    uint32_t n, k;
    uint64_t t, q[FHE_MAX_K];
};
struct fhe_dct_plan {
    double quant[64];
};

const char *fhe_last_error(void) { return g_err; }
uint32_t fhe_abi_version(void) { return FHE_ABI_VERSION; }

int fhe_default_coeff_modulus(uint32_t n, int preset, uint64_t *q_out) {
    /* the same published SEAL prime tables the product restates */
    static const uint64_t s3_4096[] = {0xffffee001ULL, 0xffffc4001ULL, 0x1ffffe0001ULL};
    static const uint64_t s3_8192[] = {0x7fffffd8001ULL, 0x7fffffc8001ULL, 0xfffffffc001ULL, 0xffffff6c001ULL,
                                       0xfffffebc001ULL};
    static const uint64_t s23_2048[] = {0x3fffffff000001ULL};
    static const uint64_t s23_4096[] = {0x7fffffff380001ULL, 0x3fffffff000001ULL};
    static const uint64_t s23_8192[] = {0x7fffffff380001ULL, 0x7ffffffef00001ULL, 0x3fffffff000001ULL,
                                        0x3ffffffef40001ULL};
    static const uint64_t s23_16384[] = {0x7fffffff380001ULL, 0x7ffffffef00001ULL, 0x7ffffffeac0001ULL, 0x7ffffffe700001ULL,
                                         0x7ffffffe600001ULL, 0x7ffffffe4c0001ULL, 0x3fffffff000001ULL, 0x3ffffffef40001ULL};
    const uint64_t *src = NULL;
    int cnt = 0;
    if (n == 2048 || n == 1024) { src = s23_2048; cnt = 1; }
    else if (n == 16384) { src = s23_16384; cnt = 8; }
    else if (n == 4096) { src = preset ? s23_4096 : s3_4096; cnt = preset ? 2 : 3; }
    else if (n == 8192) { src = preset ? s23_8192 : s3_8192; cnt = preset ? 4 : 5; }
    if (!src || preset < 0 || preset > 1) return fail(FHE_ERR_PARAM, "no default coefficient modulus for n=%u", n);
    memcpy(q_out, src, sizeof(uint64_t) * cnt);
    return cnt;
}

int fhe_ctx_create(uint32_t n, const uint64_t *q, uint32_t k, uint64_t t, int device, fhe_ctx **out) {
    (void)device;
    if (!out || !q || k == 0 || k > FHE_MAX_K) return fail(FHE_ERR_PARAM, "bad argument");
    fo_ctx *o = fo_ctx_create(n, q, k, t);
    if (!o) return fail(FHE_ERR_PARAM, "oracle rejected the parameter set");
    fhe_ctx *c = (fhe_ctx *)calloc(1, sizeof *c);
    c->o = o; c->n = n; c->k = k; c->t = t;
    memcpy(c->q, q, sizeof(uint64_t) * k);
    *out = c;
    return FHE_OK;
}
int fhe_ctx_destroy(fhe_ctx *c) {
    if (c) { fo_ctx_destroy(c->o); free(c); }
    return FHE_OK;
}
uint32_t fhe_ctx_n(const fhe_ctx *c) { return c->n; }
uint32_t fhe_ctx_k(const fhe_ctx *c) { return c->k; }
uint64_t fhe_ctx_t(const fhe_ctx *c) { return c->t; }
uint64_t fhe_ctx_q(const fhe_ctx *c, uint32_t i) { return i < c->k ? c->q[i] : 0; }

int fhe_dev_alloc(size_t bytes, void **p) {
    *p = malloc(bytes ? bytes : 8);
    return *p ? FHE_OK : fail(FHE_ERR_NOMEM, "malloc(%zu)", bytes);
}
int fhe_dev_free(void *p) { free(p); return FHE_OK; }
int fhe_host_alloc(size_t b, void **p) { if (!p) return FHE_ERR_PARAM; *p = b ? malloc(b) : NULL; return (b && !*p) ? FHE_ERR_HIP : FHE_OK; }
int fhe_host_free(void *p) { free(p); return FHE_OK; }
int fhe_upload(void *d, const void *s, size_t b, fhe_stream st) { (void)st; memcpy(d, s, b); return FHE_OK; }
int fhe_download(void *d, const void *s, size_t b, fhe_stream st) { (void)st; memcpy(d, s, b); return FHE_OK; }
int fhe_copy(void *d, const void *s, size_t b, fhe_stream st) { (void)st; memmove(d, s, b); return FHE_OK; }
int fhe_stream_sync(fhe_stream st) { (void)st; return FHE_OK; }
/* every "stream" of the CPU backend is the calling thread: transfers and operations complete before they return */
int fhe_stream_create(fhe_stream *out) { if (!out) return FHE_ERR_PARAM; *out = (fhe_stream)(uintptr_t)1; return FHE_OK; }
int fhe_stream_destroy(fhe_stream st) { (void)st; return FHE_OK; }
int fhe_gather(const uint64_t *const *src, uint64_t count, uint64_t words, uint64_t *dst, uint64_t stride, fhe_stream st) {
    (void)st;
    for (uint64_t i = 0; i < count; i++) memmove(dst + i * stride, src[i], words * 8);
    return FHE_OK;
}

/* encode/decode depend on (n, t) only; keep one small oracle context per pair */
static fo_ctx *codec_ctx(uint32_t n, uint64_t t) {
    static struct { uint32_t n; uint64_t t; fo_ctx *o; } cache[16];
    static int used;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;       /* the facade's thread test encodes from several threads */
    pthread_mutex_lock(&mu);
    fo_ctx *o = NULL;
    for (int i = 0; i < used && !o; i++)
        if (cache[i].n == n && cache[i].t == t) o = cache[i].o;
    if (!o) {
        uint64_t q[FHE_MAX_K];
        if (fhe_default_coeff_modulus(8192, 1, q) > 0) o = fo_ctx_create(n, q, 1, t);      /* 0x7fffffff380001 = 1 mod 2^19 */
        if (o && used < 16) { cache[used].n = n; cache[used].t = t; cache[used].o = o; used++; }
    }
    pthread_mutex_unlock(&mu);
    return o;
}
int fhe_frac_encode(uint32_t n, uint64_t t, double v, int ic, int fc, uint64_t *plain) {
    fo_ctx *o = codec_ctx(n, t);
    if (!o) return fail(FHE_ERR_PARAM, "no codec context for n=%u", n);
    return (int)fo_frac_encode(o, v, ic, fc, plain);
}
double fhe_frac_decode(uint32_t n, uint64_t t, const uint64_t *plain, int ic, int fc) {
    fo_ctx *o = codec_ctx(n, t);
    return o ? fo_frac_decode(o, plain, ic, fc) : 0.0;
}

static size_t pw(const fhe_ctx *c) { return (size_t)c->k * c->n; }

int fhe_add(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t np, fhe_stream s) {
    (void)s;
    for (uint64_t p = 0; p < np; p++)
        for (uint32_t i = 0; i < c->k; i++) {
            const uint64_t q = c->q[i];
            size_t o = p * pw(c) + (size_t)i * c->n;
            for (uint32_t l = 0; l < c->n; l++) { uint64_t v = a[o + l] + b[o + l]; out[o + l] = v >= q ? v - q : v; }
        }
    return FHE_OK;
}
int fhe_sub(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t np, fhe_stream s) {
    (void)s;
    for (uint64_t p = 0; p < np; p++)
        for (uint32_t i = 0; i < c->k; i++) {
            const uint64_t q = c->q[i];
            size_t o = p * pw(c) + (size_t)i * c->n;
            for (uint32_t l = 0; l < c->n; l++) out[o + l] = a[o + l] >= b[o + l] ? a[o + l] - b[o + l] : a[o + l] + q - b[o + l];
        }
    return FHE_OK;
}
int fhe_negate(const fhe_ctx *c, const uint64_t *a, uint64_t *out, uint64_t np, fhe_stream s) {
    (void)s;
    for (uint64_t p = 0; p < np; p++)
        for (uint32_t i = 0; i < c->k; i++) {
            size_t o = p * pw(c) + (size_t)i * c->n;
            for (uint32_t l = 0; l < c->n; l++) out[o + l] = a[o + l] ? c->q[i] - a[o + l] : 0;
        }
    return FHE_OK;
}
int fhe_add_sizes(const fhe_ctx *c, const uint64_t *a, uint32_t sa, const uint64_t *b, uint32_t sb, uint64_t *out, uint64_t count,
                  int subtract, fhe_stream s) {
    (void)s;
    const uint32_t so = sa > sb ? sa : sb;
    uint64_t *tmp = (uint64_t *)malloc((size_t)so * pw(c) * 8);
    if (!tmp) return fail(FHE_ERR_NOMEM, "out of memory");
    for (uint64_t i = 0; i < count; i++) {              /* fo_add / fo_sub grow their first operand in place */
        memset(tmp, 0, (size_t)so * pw(c) * 8);
        memcpy(tmp, a + i * sa * pw(c), (size_t)sa * pw(c) * 8);
        if (subtract) fo_sub(c->o, tmp, sa, b + i * sb * pw(c), sb);
        else fo_add(c->o, tmp, sa, b + i * sb * pw(c), sb);
        memcpy(out + i * so * pw(c), tmp, (size_t)so * pw(c) * 8);
    }
    free(tmp);
    return FHE_OK;
}

int fhe_ntt_forward(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t np, fhe_stream s) {
    (void)s;
    if (in != out) memmove(out, in, np * pw(c) * 8);
    for (uint64_t p = 0; p < np; p++)
        for (uint32_t i = 0; i < c->k; i++) fo_ntt_fwd(c->o, 0, i, out + p * pw(c) + (size_t)i * c->n);
    return FHE_OK;
}
int fhe_ntt_inverse(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t np, fhe_stream s) {
    (void)s;
    if (in != out) memmove(out, in, np * pw(c) * 8);
    for (uint64_t p = 0; p < np; p++)
        for (uint32_t i = 0; i < c->k; i++) fo_ntt_inv(c->o, 0, i, out + p * pw(c) + (size_t)i * c->n);
    return FHE_OK;
}
int fhe_dyadic_multiply(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t np,
                        fhe_stream s) {
    (void)s;
    for (uint64_t p = 0; p < np; p++)
        for (uint32_t i = 0; i < c->k; i++) {
            size_t o = p * pw(c) + (size_t)i * c->n;
            for (uint32_t l = 0; l < c->n; l++) out[o + l] = (uint64_t)(((u128)a[o + l] * b[o + l]) % c->q[i]);
        }
    return FHE_OK;
}

size_t fhe_plain_ntt_words(const fhe_ctx *c) { return pw(c); }
int fhe_plain_prepare(const fhe_ctx *c, const uint64_t *plain, uint32_t len, uint64_t *d, fhe_stream s) {
    fo_plain_lift(c->o, plain, len, d);
    return fhe_ntt_forward(c, d, d, 1, s);
}
int fhe_plain_ntt_mul(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, fhe_stream s) {
    return fhe_dyadic_multiply(c, a, b, out, 1, s);
}
int fhe_multiply_plain(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t np, const uint64_t *d_plain,
                       fhe_stream s) {
    fhe_ntt_forward(c, in, out, np, s);
    for (uint64_t p = 0; p < np; p++) fhe_dyadic_multiply(c, out + p * pw(c), d_plain, out + p * pw(c), 1, s);
    return fhe_ntt_inverse(c, out, out, np, s);
}
/* Cubic's linear parts, composed from the oracle's Evaluator-level calls exactly as homo/fhe_resize.h does */
static void enc_small(const fhe_ctx *c, double v, uint64_t *plain, uint32_t *len) { *len = fo_frac_encode(c->o, v, 100, 100, plain); }
int fhe_cubic_coeffs(const fhe_ctx *c, const uint64_t *A, const uint64_t *B, const uint64_t *C, const uint64_t *D, uint64_t *a,
                     uint64_t *b, uint64_t *cc, uint32_t size, uint64_t count, fhe_stream s) {
    (void)s;
    const size_t w = (size_t)size * pw(c);
    uint64_t *pl = (uint64_t *)calloc(c->n, 8), *t1 = (uint64_t *)malloc(w * 8);
    uint32_t len;
    for (uint64_t i = 0; i < count; i++) {
        const uint64_t *Ai = A + i * w, *Bi = B + i * w, *Ci = C + i * w, *Di = D + i * w;
        uint64_t *ai = a + i * w, *bi = b + i * w, *ci = cc + i * w;
        memcpy(ai, Bi, w * 8); enc_small(c, 3.0, pl, &len); fo_multiply_plain(c->o, ai, size, pl, len);
        fo_sub(c->o, ai, size, Ai, size);
        memcpy(t1, Ci, w * 8); fo_multiply_plain(c->o, t1, size, pl, len); fo_sub(c->o, ai, size, t1, size);
        fo_add(c->o, ai, size, Di, size);
        memcpy(bi, Ai, w * 8); enc_small(c, 2.0, pl, &len); fo_multiply_plain(c->o, bi, size, pl, len);
        memcpy(t1, Bi, w * 8); enc_small(c, 5.0, pl, &len); fo_multiply_plain(c->o, t1, size, pl, len); fo_sub(c->o, bi, size, t1, size);
        memcpy(t1, Ci, w * 8); enc_small(c, 4.0, pl, &len); fo_multiply_plain(c->o, t1, size, pl, len); fo_add(c->o, bi, size, t1, size);
        fo_sub(c->o, bi, size, Di, size);
        memcpy(ci, Ci, w * 8); fo_sub(c->o, ci, size, Ai, size);
    }
    free(pl); free(t1);
    return FHE_OK;
}
int fhe_cubic_combine(const fhe_ctx *c, const uint64_t *a, const uint64_t *b, const uint64_t *cc, uint32_t size_abc, const uint64_t *B,
                      uint32_t size_b, uint64_t *out, uint64_t count, fhe_stream s) {
    (void)s;
    const size_t w = (size_t)size_abc * pw(c), wb = (size_t)size_b * pw(c);
    uint64_t *pl = (uint64_t *)calloc(c->n, 8);
    uint32_t len;
    enc_small(c, 0.5, pl, &len);
    for (uint64_t i = 0; i < count; i++) {
        uint64_t *o = out + i * w;
        memcpy(o, a + i * w, w * 8);
        fo_add(c->o, o, size_abc, b + i * w, size_abc);
        fo_add(c->o, o, size_abc, cc + i * w, size_abc);
        fo_multiply_plain(c->o, o, size_abc, pl, len);
        fo_add(c->o, o, size_abc, B + i * wb, size_b);
    }
    free(pl);
    return FHE_OK;
}
int fhe_multiply_plain_sparse(const fhe_ctx *c, const uint64_t *in, uint64_t *out, uint64_t np, const uint64_t *plain,
                              uint32_t len, fhe_stream s) {
    (void)s;
    if (in != out) memmove(out, in, np * pw(c) * 8);
    fo_multiply_plain(c->o, out, (uint32_t)np, plain, len);
    return FHE_OK;
}
int fhe_add_plain(const fhe_ctx *c, uint64_t *ct, uint64_t stride, uint64_t count, const uint64_t *plain,
                  uint32_t len, int sign, fhe_stream s) {
    (void)s;
    for (uint64_t b = 0; b < count; b++) {
        if (sign >= 0) fo_add_plain(c->o, ct + b * stride, plain, len);
        else fo_sub_plain(c->o, ct + b * stride, plain, len);
    }
    return FHE_OK;
}

/* server-side encryptions (include/fhe_hip.h): the oracle's restatement of the keyed sampler; pk arrives in NTT form */
void fhe_noise_cdt(uint64_t out[FHE_NOISE_CDT_LEN]) { fo_noise_cdt(out); }
int fhe_frac_encode_batch(const fhe_ctx *c, const double *values, uint64_t count, int ic, int fc, uint64_t *d_plain, fhe_stream s) {
    (void)s;
    for (uint64_t i = 0; i < count; i++) {
        memset(d_plain + i * c->n, 0, (size_t)c->n * 8);
        int len = fhe_frac_encode(c->n, c->t, values[i], ic, fc, d_plain + i * c->n);
        if (len < 0) return len;
    }
    return FHE_OK;
}
size_t fhe_encrypt_scratch_bytes(const fhe_ctx *c, uint64_t count) { return (size_t)count * pw(c) * 8; }
int fhe_encrypt_batch(const fhe_ctx *c, const uint64_t *pk_ntt, const uint64_t *d_plain, uint64_t count, const uint8_t key[32], uint64_t first,
                      uint64_t *out, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!c || !pk_ntt || !key || (!out && count)) return fail(FHE_ERR_PARAM, "null argument");
    if (count && (!scratch || scratch_bytes < fhe_encrypt_scratch_bytes(c, count))) return fail(FHE_ERR_PARAM, "scratch too small: need fhe_encrypt_scratch_bytes()");
    uint64_t *pk = (uint64_t *)malloc(2 * pw(c) * 8);
    if (!pk) return fail(FHE_ERR_HIP, "out of memory");
    fhe_ntt_inverse(c, pk_ntt, pk, 2, s);
    for (uint64_t i = 0; i < count; i++) {
        uint32_t len = 0;
        if (d_plain) { len = c->n; while (len && !d_plain[i * c->n + len - 1]) len--; }
        fo_encrypt_keyed(c->o, pk, d_plain ? d_plain + i * c->n : NULL, len, key, first + i, out + i * 2 * pw(c));
    }
    free(pk);
    return FHE_OK;
}
int fhe_encrypt_draws(const fhe_ctx *c, const uint8_t key[32], uint64_t first, uint64_t count, int8_t *d_draws, fhe_stream s) {
    (void)s;
    for (uint64_t i = 0; i < count; i++) fo_encrypt_draws(c->n, key, first + i, d_draws + i * 3 * c->n);
    return FHE_OK;
}

/* decryption in batches (include/fhe_hip.h): the oracle's big-integer decryption per ciphertext; the key arrives in NTT form */
uint32_t fhe_ctx_modulus_bits(const fhe_ctx *c) {
    int nb = 0, mb = 0;
    uint64_t *z = (uint64_t *)calloc(3 * pw(c) + c->n, 8);
    if (!z) return 0;
    fo_decrypt_noise_bits(c->o, z, z + pw(c), 2, z + 3 * pw(c), &nb, &mb);
    free(z);
    return (uint32_t)mb;
}
size_t fhe_decrypt_scratch_bytes(const fhe_ctx *c, uint32_t size, uint64_t count) { return (size_t)count * (size + 1) * pw(c) * 8; }
int fhe_decrypt_batch(const fhe_ctx *c, const uint64_t *sk_ntt, const uint64_t *ct, uint32_t size, uint64_t count, uint64_t *plain, uint32_t *noise_bits,
                      void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!c || !sk_ntt || (!ct && count) || (!plain && count)) return fail(FHE_ERR_PARAM, "null argument");
    if (size < 2) return fail(FHE_ERR_PARAM, "a ciphertext has at least two polynomials");
    if (count && (!scratch || scratch_bytes < fhe_decrypt_scratch_bytes(c, size, count))) return fail(FHE_ERR_PARAM, "scratch too small: need fhe_decrypt_scratch_bytes()");
    uint64_t *sk = (uint64_t *)malloc(pw(c) * 8);
    if (!sk) return fail(FHE_ERR_HIP, "out of memory");
    fhe_ntt_inverse(c, sk_ntt, sk, 1, s);
    for (uint64_t i = 0; i < count; i++) {
        int nb = 0;
        fo_decrypt_noise_bits(c->o, sk, ct + i * size * pw(c), size, plain + i * c->n, &nb, NULL);
        if (noise_bits) noise_bits[i] = (uint32_t)nb;
    }
    free(sk);
    return FHE_OK;
}

size_t fhe_multiply_scratch_bytes(const fhe_ctx *c, uint32_t sa, uint32_t sb, uint64_t count) {
    (void)c; (void)sa; (void)sb; (void)count;
    return 8;
}
int fhe_multiply(const fhe_ctx *c, const uint64_t *a, uint32_t sa, const uint64_t *b, uint32_t sb, uint64_t *out,
                 uint64_t count, void *scr, size_t sbytes, fhe_stream s) {
    (void)scr; (void)sbytes; (void)s;
    for (uint64_t i = 0; i < count; i++)
        fo_multiply(c->o, a + i * sa * pw(c), sa, b + i * sb * pw(c), sb, out + i * (sa + sb - 1) * pw(c));
    return FHE_OK;
}
int fhe_square(const fhe_ctx *c, const uint64_t *a, uint32_t sa, uint64_t *out, uint64_t count, void *scr,
               size_t sbytes, fhe_stream s) {
    (void)scr; (void)sbytes; (void)s;
    for (uint64_t i = 0; i < count; i++) fo_square(c->o, a + i * sa * pw(c), sa, out + i * (2 * sa - 1) * pw(c));
    return FHE_OK;
}
uint32_t fhe_evk_digits(const fhe_ctx *c, uint32_t dbc) { return fo_evk_digits(c->o, dbc); }
size_t fhe_relinearize_scratch_bytes(const fhe_ctx *c, uint32_t dbc, uint64_t count) {
    (void)c; (void)dbc; (void)count;
    return 8;
}
int fhe_relinearize(const fhe_ctx *c, uint64_t *ct3, uint64_t stride, uint64_t count, const uint64_t *evk,
                    uint32_t dbc, void *scr, size_t sbytes, fhe_stream s) {
    (void)scr; (void)sbytes; (void)s;
    for (uint64_t i = 0; i < count; i++) fo_relinearize3(c->o, ct3 + i * stride, evk, dbc);
    return FHE_OK;
}
int fhe_relinearize_to(const fhe_ctx *c, const uint64_t *ct3, uint64_t stride, uint64_t *out2, uint64_t out_stride,
                       uint64_t count, const uint64_t *evk, uint32_t dbc, void *scr, size_t sbytes, fhe_stream s) {
    (void)scr; (void)sbytes; (void)s;
    uint64_t *tmp = (uint64_t *)malloc(3 * pw(c) * 8);
    if (!tmp) return fail(FHE_ERR_NOMEM, "out of memory");
    for (uint64_t i = 0; i < count; i++) {
        memcpy(tmp, ct3 + i * stride, 3 * pw(c) * 8);
        fo_relinearize3(c->o, tmp, evk, dbc);
        memcpy(out2 + i * out_stride, tmp, 2 * pw(c) * 8);
    }
    free(tmp);
    return FHE_OK;
}

int fhe_relinearize_poly(const fhe_ctx *c, const uint64_t *ct, uint64_t stride, uint32_t src_poly, uint64_t *out2, uint64_t out_stride,
                         uint64_t count, const uint64_t *evk, uint32_t dbc, void *scr, size_t sbytes, fhe_stream s) {
    (void)scr; (void)sbytes; (void)s;
    if (src_poly < 2 || src_poly >= FHE_MAX_POLYS) return fail(FHE_ERR_PARAM, "key-switch source polynomial out of range");
    uint64_t *tmp = (uint64_t *)malloc((src_poly + 1) * pw(c) * 8);
    if (!tmp) return fail(FHE_ERR_NOMEM, "out of memory");
    for (uint64_t i = 0; i < count; i++) {
        memcpy(tmp, ct + i * stride, (src_poly + 1) * pw(c) * 8);
        fo_relinearize_poly(c->o, tmp, src_poly, evk, dbc);
        memcpy(out2 + i * out_stride, tmp, 2 * pw(c) * 8);
    }
    free(tmp);
    return FHE_OK;
}
size_t fhe_relinearize_n_scratch_bytes(const fhe_ctx *c, uint32_t size, uint32_t dbc, uint64_t count) {
    (void)c; (void)size; (void)dbc; (void)count;
    return 8;
}
size_t fhe_evk_words(const fhe_ctx *c, uint32_t dbc) { return (size_t)c->k * fo_evk_digits(c->o, dbc) * 2 * pw(c); }
int fhe_relinearize_n(const fhe_ctx *c, uint64_t *ct, uint32_t size, uint64_t stride, uint64_t *out2, uint64_t out_stride, uint64_t count,
                      const uint64_t *evk, uint32_t dbc, void *scr, size_t sbytes, fhe_stream s) {
    if (size < 3 || size > FHE_MAX_POLYS) return fail(FHE_ERR_PARAM, "relinearize: size out of range");
    const size_t ew = fhe_evk_words(c, dbc);
    for (uint32_t p = size - 1; p >= 3; --p) {
        int rc = fhe_relinearize_poly(c, ct, stride, p, ct, stride, count, evk + (size_t)(p - 2) * ew, dbc, scr, sbytes, s);
        if (rc) return rc;
    }
    return fhe_relinearize_poly(c, ct, stride, 2, out2, out_stride, count, evk, dbc, scr, sbytes, s);
}

int fhe_dct_plan_create(const fhe_ctx *c, const double *quant64, int ic, int fc, fhe_stream s, fhe_dct_plan **out) {
    (void)c; (void)s;
    if (ic != 100 || fc != 100) return fail(FHE_ERR_PARAM, "the oracle fixes 100/100 coefficients");
    fhe_dct_plan *p = (fhe_dct_plan *)calloc(1, sizeof *p);
    memcpy(p->quant, quant64, sizeof p->quant);
    *out = p;
    return FHE_OK;
}
int fhe_dct_plan_destroy(fhe_dct_plan *p) { free(p); return FHE_OK; }
size_t fhe_dct8x8_scratch_bytes(const fhe_ctx *c, uint64_t nb) { (void)c; (void)nb; return 8; }
int fhe_dct8x8_quant(const fhe_ctx *c, const fhe_dct_plan *plan, const uint64_t *in, uint64_t *out, uint64_t nb,
                     void *scr, size_t sbytes, fhe_stream s) {
    (void)scr; (void)sbytes; (void)s;
    const size_t bw = 64 * 2 * pw(c);
    if (in != out) memmove(out, in, nb * bw * 8);
    for (uint64_t b = 0; b < nb; b++) {
        fo_encrypted_dct(c->o, out + b * bw);
        fo_quantize(c->o, out + b * bw, plan->quant);
    }
    return FHE_OK;
}
int fhe_rgb_to_ycc(const fhe_ctx *c, uint64_t *r, uint64_t *g, uint64_t *b, uint64_t count, int ic, int fc,
                   fhe_stream s) {
    (void)s;
    if (ic != 100 || fc != 100) return fail(FHE_ERR_PARAM, "the oracle fixes 100/100 coefficients");
    for (uint64_t i = 0; i < count; i++) fo_rgb_to_ycc(c->o, r + i * 2 * pw(c), g + i * 2 * pw(c), b + i * 2 * pw(c));
    return FHE_OK;
}
int fhe_fill_random(const fhe_ctx *c, uint64_t *ct, uint64_t np, uint64_t seed, uint64_t first, fhe_stream s) {
    (void)s;
    fo_fill_random_ct(c->o, ct, np, seed, first);
    return FHE_OK;
}
int fhe_digest(const fhe_ctx *c, const uint64_t *data, uint64_t count, uint64_t index0, uint64_t *d_out,
               fhe_stream s) {
    (void)c; (void)s; (void)index0;
    *d_out = fo_digest(data, count);          /* oracle digest; not the product's position-salted one */
    return FHE_OK;
}
