/*
 * fhe_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See fhe_oracle.h for scope, citations and the PARITY STATUS statement.
 *
 * Deliberately simple: schoolbook-order loops, unsigned __int128 products with
 * a hardware remainder, one operation at a time, plaintexts re-lifted and
 * re-transformed on every multiply_plain exactly as the reference's call
 * pattern makes SEAL do (homo/fhe_image.h:221 `multiply_plain(x, encoder.encode(c))`).
 * One concession to speed, because this file is also bench.py's cpu_baseline ("port"): the
 * butterflies of the two transforms multiply by their table constants the way SEAL's own
 * transforms do (a precomputed quotient per twiddle, "Shoup"/Harvey style, App. A.3 of SURVEY.md)
 * instead of a 128-by-64-bit hardware division per butterfly; the value is the same canonical
 * residue (mulmod_const below is checked against mulmod in tests/test_oracle_properties.py).
 */
#include "fhe_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;

#define MAX_LIMBS (FO_MAX_K + 2)

/* ------------------------------------------------------------------------- */
/* scalar modular helpers                                                     */
/* ------------------------------------------------------------------------- */
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }
static inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
/* a * w mod q for a constant w with wq = floor(w 2^64 / q): q < 2^63, any a < 2^64 */
static inline u64 const_quotient(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
static inline u64 mulmod_const(u64 a, u64 w, u64 wq, u64 q) {
    const u64 hi = (u64)(((u128)a * wq) >> 64);
    const u64 r = a * w - hi * q;              /* exact value in [0, 2q), taken modulo 2^64 */
    return r >= q ? r - q : r;
}
uint64_t fo_mulmod_const_check(uint64_t a, uint64_t w, uint64_t q) { return mulmod_const(a, w, const_quotient(w, q), q); }
static u64 powmod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = mulmod(r, a, q);
        a = mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}
static u64 invmod_prime(u64 a, u64 q) { return powmod(a % q, q - 2, q); }

static int is_prime_u64(u64 n) {
    if (n < 2) return 0;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (unsigned i = 0; i < 12; i++) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    u64 d = n - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    for (unsigned i = 0; i < 12; i++) { /* deterministic for 64-bit with these bases */
        u64 x = powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) {
            x = mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

static u32 bitrev(u32 x, u32 bits) {
    u32 r = 0;
    for (u32 i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

uint64_t fo_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* ------------------------------------------------------------------------- */
/* per-prime NTT tables                                                       */
/* ------------------------------------------------------------------------- */
typedef struct {
    u64 q;
    u64 *psi_br;  /* psi^bitrev(i) */
    u64 *ipsi_br; /* psi^-bitrev(i) */
    u64 *psi_brq, *ipsi_brq; /* floor(w 2^64 / q) of the two tables */
    u64 ninv, ninvq;
} ntt_tab;

static int ntt_tab_init(ntt_tab *T, u64 q, u32 n, u32 logn) {
    T->q = q;
    if (!is_prime_u64(q) || (q - 1) % (2ULL * n) != 0) return -1;
    u64 psi = 0;
    for (u64 g = 2; g < 1000; g++) {
        u64 cand = powmod(g, (q - 1) / (2ULL * n), q);
        if (powmod(cand, n, q) == q - 1) { psi = cand; break; }
    }
    if (!psi) return -1;
    u64 ipsi = invmod_prime(psi, q);
    T->psi_br = (u64 *)malloc(sizeof(u64) * n);
    T->ipsi_br = (u64 *)malloc(sizeof(u64) * n);
    T->psi_brq = (u64 *)malloc(sizeof(u64) * n);
    T->ipsi_brq = (u64 *)malloc(sizeof(u64) * n);
    u64 p = 1, ip = 1;
    for (u32 i = 0; i < n; i++) {
        u32 r = bitrev(i, logn);
        T->psi_br[r] = p;
        T->ipsi_br[r] = ip;
        p = mulmod(p, psi, q);
        ip = mulmod(ip, ipsi, q);
    }
    for (u32 i = 0; i < n; i++) {
        T->psi_brq[i] = const_quotient(T->psi_br[i], q);
        T->ipsi_brq[i] = const_quotient(T->ipsi_br[i], q);
    }
    T->ninv = invmod_prime(n, q);
    T->ninvq = const_quotient(T->ninv, q);
    return 0;
}
static void ntt_tab_free(ntt_tab *T) { free(T->psi_br); free(T->ipsi_br); free(T->psi_brq); free(T->ipsi_brq); }

/* Cooley-Tukey, natural in -> bit-reversed out (merged psi twiddles) */
static void ntt_fwd(const ntt_tab *T, u32 n, u64 *a) {
    u64 q = T->q;
    u32 t = n;
    for (u32 m = 1; m < n; m <<= 1) {
        t >>= 1;
        for (u32 i = 0; i < m; i++) {
            u64 W = T->psi_br[m + i], Wq = T->psi_brq[m + i];
            u32 j1 = 2 * i * t;
            for (u32 j = j1; j < j1 + t; j++) {
                u64 U = a[j], V = mulmod_const(a[j + t], W, Wq, q);
                a[j] = addmod(U, V, q);
                a[j + t] = submod(U, V, q);
            }
        }
    }
}
/* Gentleman-Sande, bit-reversed in -> natural out, scaled by n^-1 */
static void ntt_inv(const ntt_tab *T, u32 n, u64 *a) {
    u64 q = T->q;
    u32 t = 1;
    for (u32 m = n; m > 1; m >>= 1) {
        u32 h = m >> 1, j1 = 0;
        for (u32 i = 0; i < h; i++) {
            u64 W = T->ipsi_br[h + i], Wq = T->ipsi_brq[h + i];
            for (u32 j = j1; j < j1 + t; j++) {
                u64 U = a[j], V = a[j + t];
                a[j] = addmod(U, V, q);
                a[j + t] = mulmod_const(submod(U, V, q), W, Wq, q);
            }
            j1 += 2 * t;
        }
        t <<= 1;
    }
    for (u32 j = 0; j < n; j++) a[j] = mulmod_const(a[j], T->ninv, T->ninvq, q);
}

/* ------------------------------------------------------------------------- */
/* tiny fixed-width big integers (little-endian limbs) for CRT / rounding     */
/* ------------------------------------------------------------------------- */
typedef struct { u64 w[MAX_LIMBS]; } big;
static void big_zero(big *a) { memset(a, 0, sizeof(*a)); }
static void big_from_u64(big *a, u64 v) { big_zero(a); a->w[0] = v; }
static int big_cmp(const big *a, const big *b) {
    for (int i = MAX_LIMBS - 1; i >= 0; i--) {
        if (a->w[i] != b->w[i]) return a->w[i] > b->w[i] ? 1 : -1;
    }
    return 0;
}
static void big_add(big *a, const big *b) {
    u128 c = 0;
    for (int i = 0; i < MAX_LIMBS; i++) { c += (u128)a->w[i] + b->w[i]; a->w[i] = (u64)c; c >>= 64; }
}
static void big_sub(big *a, const big *b) { /* a >= b */
    u64 br = 0;
    for (int i = 0; i < MAX_LIMBS; i++) {
        u128 d = (u128)a->w[i] - b->w[i] - br;
        a->w[i] = (u64)d;
        br = (u64)(d >> 64) & 1;
    }
}
static void big_mul_small(big *r, const big *a, u64 s) {
    u128 c = 0;
    for (int i = 0; i < MAX_LIMBS; i++) { c += (u128)a->w[i] * s; r->w[i] = (u64)c; c >>= 64; }
}
static u64 big_mod_small(const big *a, u64 m) {
    u128 r = 0;
    for (int i = MAX_LIMBS - 1; i >= 0; i--) r = ((r << 64) | a->w[i]) % m;
    return (u64)r;
}
static int big_bits(const big *a) {
    for (int i = MAX_LIMBS - 1; i >= 0; i--)
        if (a->w[i]) return 64 * i + (64 - __builtin_clzll(a->w[i]));
    return 0;
}
static void big_shr1(big *a) {
    for (int i = 0; i < MAX_LIMBS; i++) {
        u64 hi = (i + 1 < MAX_LIMBS) ? a->w[i + 1] : 0;
        a->w[i] = (a->w[i] >> 1) | (hi << 63);
    }
}

/* ------------------------------------------------------------------------- */
/* context                                                                    */
/* ------------------------------------------------------------------------- */
struct fo_ctx {
    u32 n, logn, k;
    u64 t;
    u64 q[FO_MAX_K];
    ntt_tab qt[FO_MAX_K];
    /* plaintext lifting (App. A.3 of SURVEY.md; SEAL 2.3 Evaluator::multiply_plain / preencrypt) */
    u64 plain_upper_half_threshold;        /* (t+1)/2 */
    u64 plain_upper_half_increment[FO_MAX_K]; /* (q - t) mod q_i  */
    u64 delta_mod[FO_MAX_K];               /* floor(q/t) mod q_i */
    u64 upper_half_increment[FO_MAX_K];    /* (q mod t) mod q_i  */
    /* CRT */
    big qbig, qhalf;
    big punct_big[FO_MAX_K];    /* q/q_i */
    u64 inv_punct[FO_MAX_K];    /* (q/q_i)^-1 mod q_i */
    /* BEHZ */
    u32 nb;                     /* |Bsk| = k+1 ; index k = m_sk */
    u64 bsk[FO_MAX_K + 1];
    ntt_tab bt[FO_MAX_K + 1];
    u64 mtilde;                 /* 2^32 */
    u64 punct_q_mod_bsk[FO_MAX_K][FO_MAX_K + 1]; /* (q/q_i) mod bsk_j */
    u64 punct_q_mod_mtilde[FO_MAX_K];
    u64 neg_inv_q_mod_mtilde;
    u64 q_mod_bsk[FO_MAX_K + 1];
    u64 inv_mtilde_mod_bsk[FO_MAX_K + 1];
    u64 inv_q_mod_bsk[FO_MAX_K + 1];
    u64 inv_punct_B[FO_MAX_K];             /* (B/b_j)^-1 mod b_j */
    u64 punct_B_mod_q[FO_MAX_K][FO_MAX_K]; /* (B/b_j) mod q_i */
    u64 punct_B_mod_msk[FO_MAX_K];
    u64 inv_B_mod_msk;
    u64 B_mod_q[FO_MAX_K];
};

uint32_t fo_ctx_n(const fo_ctx *c) { return c->n; }
uint32_t fo_ctx_k(const fo_ctx *c) { return c->k; }
uint64_t fo_ctx_t(const fo_ctx *c) { return c->t; }
uint64_t fo_ctx_q(const fo_ctx *c, uint32_t i) { return c->q[i]; }
uint64_t fo_ctx_aux(const fo_ctx *c, uint32_t i) { return c->bsk[i]; }

fo_ctx *fo_ctx_create(uint32_t n, const uint64_t *q, uint32_t k, uint64_t t) {
    if (k == 0 || k > FO_MAX_K || n < 4 || (n & (n - 1))) return NULL;
    fo_ctx *c = (fo_ctx *)calloc(1, sizeof(fo_ctx));
    c->n = n;
    c->k = k;
    c->t = t;
    c->logn = 0;
    while ((1u << c->logn) < n) c->logn++;
    for (u32 i = 0; i < k; i++) {
        c->q[i] = q[i];
        if (q[i] >> 61) { free(c); return NULL; }
        if (ntt_tab_init(&c->qt[i], q[i], n, c->logn)) { free(c); return NULL; }
        for (u32 j = 0; j < i; j++)
            if (q[j] == q[i]) { free(c); return NULL; }
    }
    /* q as a big integer, punctured products */
    big_from_u64(&c->qbig, 1);
    for (u32 i = 0; i < k; i++) { big tmp; big_mul_small(&tmp, &c->qbig, q[i]); c->qbig = tmp; }
    c->qhalf = c->qbig;
    big_shr1(&c->qhalf);
    for (u32 i = 0; i < k; i++) {
        big_from_u64(&c->punct_big[i], 1);
        for (u32 j = 0; j < k; j++) {
            if (j == i) continue;
            big tmp;
            big_mul_small(&tmp, &c->punct_big[i], q[j]);
            c->punct_big[i] = tmp;
        }
        c->inv_punct[i] = invmod_prime(big_mod_small(&c->punct_big[i], q[i]), q[i]);
    }
    /* plaintext lifting constants */
    c->plain_upper_half_threshold = (t + 1) >> 1;
    u64 q_mod_t = big_mod_small(&c->qbig, t);
    for (u32 i = 0; i < k; i++) {
        c->plain_upper_half_increment[i] = submod(0, t % q[i], q[i]); /* q - t == -t (mod q_i) */
        c->upper_half_increment[i] = q_mod_t % q[i];
    }
    { /* delta = floor(q / t) = (q - (q mod t)) / t ; compute mod q_i as (-(q mod t)) * t^-1 */
        for (u32 i = 0; i < k; i++) {
            u64 tinv = invmod_prime(t % q[i], q[i]);
            c->delta_mod[i] = mulmod(negmod(q_mod_t % q[i], q[i]), tinv, q[i]);
        }
    }
    /* auxiliary base: 61-bit primes = 1 (mod 2^17), descending from 2^61; first is m_sk */
    c->nb = k + 1;
    c->mtilde = 1ULL << 32;
    {
        u64 cand = (1ULL << 61) + 1;
        u64 found[FO_MAX_K + 1];
        u32 nf = 0;
        while (nf < k + 1) {
            cand -= (1ULL << 17);
            if (!is_prime_u64(cand)) continue;
            int clash = 0;
            for (u32 i = 0; i < k; i++) clash |= (cand == q[i]);
            if (!clash) found[nf++] = cand;
        }
        for (u32 j = 0; j < k; j++) c->bsk[j] = found[j + 1];
        c->bsk[k] = found[0]; /* m_sk */
        for (u32 j = 0; j <= k; j++)
            if (ntt_tab_init(&c->bt[j], c->bsk[j], n, c->logn)) { free(c); return NULL; }
    }
    for (u32 i = 0; i < k; i++) {
        for (u32 j = 0; j <= k; j++) c->punct_q_mod_bsk[i][j] = big_mod_small(&c->punct_big[i], c->bsk[j]);
        c->punct_q_mod_mtilde[i] = big_mod_small(&c->punct_big[i], c->mtilde);
    }
    {
        u64 qm = big_mod_small(&c->qbig, c->mtilde); /* odd */
        /* inverse mod 2^32 by Newton iteration */
        u64 x = qm;
        for (int it = 0; it < 6; it++) x = (x * (2 - qm * x)) & (c->mtilde - 1);
        c->neg_inv_q_mod_mtilde = (c->mtilde - x) & (c->mtilde - 1);
    }
    for (u32 j = 0; j <= k; j++) {
        u64 b = c->bsk[j];
        c->q_mod_bsk[j] = big_mod_small(&c->qbig, b);
        c->inv_q_mod_bsk[j] = invmod_prime(c->q_mod_bsk[j], b);
        c->inv_mtilde_mod_bsk[j] = invmod_prime(c->mtilde % b, b);
    }
    { /* B = prod b_j (j<k) */
        big Bbig;
        big_from_u64(&Bbig, 1);
        for (u32 j = 0; j < k; j++) { big tmp; big_mul_small(&tmp, &Bbig, c->bsk[j]); Bbig = tmp; }
        for (u32 j = 0; j < k; j++) {
            big pb;
            big_from_u64(&pb, 1);
            for (u32 l = 0; l < k; l++) {
                if (l == j) continue;
                big tmp;
                big_mul_small(&tmp, &pb, c->bsk[l]);
                pb = tmp;
            }
            c->inv_punct_B[j] = invmod_prime(big_mod_small(&pb, c->bsk[j]), c->bsk[j]);
            for (u32 i = 0; i < k; i++) c->punct_B_mod_q[j][i] = big_mod_small(&pb, q[i]);
            c->punct_B_mod_msk[j] = big_mod_small(&pb, c->bsk[k]);
        }
        c->inv_B_mod_msk = invmod_prime(big_mod_small(&Bbig, c->bsk[k]), c->bsk[k]);
        for (u32 i = 0; i < k; i++) c->B_mod_q[i] = big_mod_small(&Bbig, q[i]);
    }
    return c;
}

void fo_ctx_destroy(fo_ctx *c) {
    if (!c) return;
    for (u32 i = 0; i < c->k; i++) ntt_tab_free(&c->qt[i]);
    for (u32 j = 0; j < c->nb; j++) ntt_tab_free(&c->bt[j]);
    free(c);
}

void fo_fill_random_ct(const fo_ctx *c, uint64_t *ct, uint64_t n_polys, uint64_t seed,
                       uint64_t first_linear_index) {
    u64 idx = first_linear_index;
    for (u64 p = 0; p < n_polys; p++)
        for (u32 i = 0; i < c->k; i++)
            for (u32 j = 0; j < c->n; j++, idx++)
                ct[(p * c->k + i) * c->n + j] = fo_splitmix64(seed ^ idx) % c->q[i];
}

void fo_ntt_fwd(const fo_ctx *c, int base, uint32_t i, uint64_t *a) {
    ntt_fwd(base ? &c->bt[i] : &c->qt[i], c->n, a);
}
void fo_ntt_inv(const fo_ctx *c, int base, uint32_t i, uint64_t *a) {
    ntt_inv(base ? &c->bt[i] : &c->qt[i], c->n, a);
}

/* ------------------------------------------------------------------------- */
/* exact ring ops                                                             */
/* ------------------------------------------------------------------------- */
#define POLY(ct, j, i) ((ct) + ((size_t)(j) * c->k + (i)) * c->n)

uint32_t fo_add(const fo_ctx *c, uint64_t *a, uint32_t sa, const uint64_t *b, uint32_t sb) {
    u32 mn = sa < sb ? sa : sb;
    for (u32 j = 0; j < mn; j++)
        for (u32 i = 0; i < c->k; i++) {
            u64 *x = POLY(a, j, i);
            const u64 *y = POLY(b, j, i);
            for (u32 l = 0; l < c->n; l++) x[l] = addmod(x[l], y[l], c->q[i]);
        }
    if (sb > sa) memcpy(POLY(a, sa, 0), POLY(b, sa, 0), sizeof(u64) * (size_t)(sb - sa) * c->k * c->n);
    return sa > sb ? sa : sb;
}
uint32_t fo_sub(const fo_ctx *c, uint64_t *a, uint32_t sa, const uint64_t *b, uint32_t sb) {
    u32 mn = sa < sb ? sa : sb;
    for (u32 j = 0; j < mn; j++)
        for (u32 i = 0; i < c->k; i++) {
            u64 *x = POLY(a, j, i);
            const u64 *y = POLY(b, j, i);
            for (u32 l = 0; l < c->n; l++) x[l] = submod(x[l], y[l], c->q[i]);
        }
    for (u32 j = sa; j < sb; j++)
        for (u32 i = 0; i < c->k; i++) {
            u64 *x = POLY(a, j, i);
            const u64 *y = POLY(b, j, i);
            for (u32 l = 0; l < c->n; l++) x[l] = negmod(y[l], c->q[i]);
        }
    return sa > sb ? sa : sb;
}
void fo_negate(const fo_ctx *c, uint64_t *a, uint32_t size) {
    for (u32 j = 0; j < size; j++)
        for (u32 i = 0; i < c->k; i++) {
            u64 *x = POLY(a, j, i);
            for (u32 l = 0; l < c->n; l++) x[l] = negmod(x[l], c->q[i]);
        }
}

/* Delta*m' mod q_i, m' = centred lift of m (SEAL 2.3 `preencrypt`) */
static inline u64 scaled_plain_coeff(const fo_ctx *c, u64 m, u32 i) {
    u64 v = mulmod(c->delta_mod[i], m % c->q[i], c->q[i]);
    if (m >= c->plain_upper_half_threshold) v = addmod(v, c->upper_half_increment[i], c->q[i]);
    return v;
}
void fo_add_plain(const fo_ctx *c, uint64_t *a, const uint64_t *plain, uint32_t len) {
    for (u32 i = 0; i < c->k; i++) {
        u64 *x = POLY(a, 0, i);
        for (u32 l = 0; l < len && l < c->n; l++) x[l] = addmod(x[l], scaled_plain_coeff(c, plain[l], i), c->q[i]);
    }
}
void fo_sub_plain(const fo_ctx *c, uint64_t *a, const uint64_t *plain, uint32_t len) {
    for (u32 i = 0; i < c->k; i++) {
        u64 *x = POLY(a, 0, i);
        for (u32 l = 0; l < len && l < c->n; l++) x[l] = submod(x[l], scaled_plain_coeff(c, plain[l], i), c->q[i]);
    }
}

void fo_plain_lift(const fo_ctx *c, const uint64_t *plain, uint32_t len, uint64_t *out) {
    memset(out, 0, sizeof(u64) * (size_t)c->k * c->n);
    for (u32 i = 0; i < c->k; i++)
        for (u32 l = 0; l < len && l < c->n; l++) {
            u64 m = plain[l];
            out[(size_t)i * c->n + l] =
                m >= c->plain_upper_half_threshold ? (m + c->plain_upper_half_increment[i]) % c->q[i] : m % c->q[i];
        }
}

void fo_multiply_plain(const fo_ctx *c, uint64_t *a, uint32_t size, const uint64_t *plain, uint32_t len) {
    u64 *P = (u64 *)malloc(sizeof(u64) * (size_t)c->k * c->n);
    fo_plain_lift(c, plain, len, P);
    for (u32 i = 0; i < c->k; i++) ntt_fwd(&c->qt[i], c->n, P + (size_t)i * c->n);
    for (u32 j = 0; j < size; j++)
        for (u32 i = 0; i < c->k; i++) {
            u64 *x = POLY(a, j, i);
            const u64 *p = P + (size_t)i * c->n;
            ntt_fwd(&c->qt[i], c->n, x);
            for (u32 l = 0; l < c->n; l++) x[l] = mulmod(x[l], p[l], c->q[i]);
            ntt_inv(&c->qt[i], c->n, x);
        }
    free(P);
}

/* ------------------------------------------------------------------------- */
/* BEHZ multiply                                          */
/* ------------------------------------------------------------------------- */
/* step 0+1: input poly (q-base, [k][n]) -> Bsk residues of c' = (x + q r)/mtilde, [k+1][n] */
static void behz_to_bsk(const fo_ctx *c, const u64 *in, u64 *out) {
    u32 k = c->k, n = c->n;
    u64 mt_mask = c->mtilde - 1;
    for (u32 l = 0; l < n; l++) {
        u64 y[FO_MAX_K];
        for (u32 i = 0; i < k; i++) {
            u64 v = mulmod(in[(size_t)i * n + l], c->mtilde % c->q[i], c->q[i]);
            y[i] = mulmod(v, c->inv_punct[i], c->q[i]);
        }
        /* FastBConv to m_tilde */
        u64 xm = 0;
        for (u32 i = 0; i < k; i++) xm = (xm + (y[i] & mt_mask) * c->punct_q_mod_mtilde[i]) & mt_mask;
        u64 r = (xm * c->neg_inv_q_mod_mtilde) & mt_mask; /* r = -x q^-1 mod mtilde, in [0, mtilde) */
        for (u32 j = 0; j <= k; j++) {
            u64 b = c->bsk[j];
            u64 xb = 0;
            for (u32 i = 0; i < k; i++) xb = addmod(xb, mulmod(y[i] % b, c->punct_q_mod_bsk[i][j], b), b);
            /* centred remainder: r >= mtilde/2 represents r - mtilde */
            u64 rb = (r >= (c->mtilde >> 1)) ? (r + b - c->mtilde) : r;
            u64 v = addmod(xb, mulmod(c->q_mod_bsk[j], rb % b, b), b);
            out[(size_t)j * n + l] = mulmod(v, c->inv_mtilde_mod_bsk[j], b);
        }
    }
}

/* steps 3+4: (t*D in q-base [k][n], t*D in Bsk [k+1][n]) -> result in q-base [k][n] */
static void behz_floor_and_back(const fo_ctx *c, const u64 *dq, u64 *dbsk, u64 *out) {
    u32 k = c->k, n = c->n;
    u64 msk = c->bsk[k];
    for (u32 l = 0; l < n; l++) {
        u64 y[FO_MAX_K];
        for (u32 i = 0; i < k; i++) y[i] = mulmod(dq[(size_t)i * n + l], c->inv_punct[i], c->q[i]);
        u64 f[FO_MAX_K + 1];
        for (u32 j = 0; j <= k; j++) { /* fast floor */
            u64 b = c->bsk[j];
            u64 conv = 0;
            for (u32 i = 0; i < k; i++) conv = addmod(conv, mulmod(y[i] % b, c->punct_q_mod_bsk[i][j], b), b);
            f[j] = mulmod(submod(dbsk[(size_t)j * n + l], conv, b), c->inv_q_mod_bsk[j], b);
        }
        /* Shenoy-Kumaresan: B -> q with m_sk correcting the overflow alpha */
        u64 z[FO_MAX_K];
        for (u32 j = 0; j < k; j++) z[j] = mulmod(f[j], c->inv_punct_B[j], c->bsk[j]);
        u64 conv_sk = 0;
        for (u32 j = 0; j < k; j++) conv_sk = addmod(conv_sk, mulmod(z[j] % msk, c->punct_B_mod_msk[j], msk), msk);
        u64 alpha = mulmod(submod(conv_sk, f[k], msk), c->inv_B_mod_msk, msk);
        int alpha_neg = alpha > (msk >> 1);
        for (u32 i = 0; i < k; i++) {
            u64 qi = c->q[i];
            u64 conv = 0;
            for (u32 j = 0; j < k; j++) conv = addmod(conv, mulmod(z[j] % qi, c->punct_B_mod_q[j][i], qi), qi);
            u64 corr = alpha_neg ? mulmod((msk - alpha) % qi, c->B_mod_q[i], qi)
                                 : negmod(mulmod(alpha % qi, c->B_mod_q[i], qi), qi);
            out[(size_t)i * n + l] = addmod(conv, corr, qi);
        }
    }
}

uint32_t fo_multiply(const fo_ctx *c, const uint64_t *a, uint32_t sa, const uint64_t *b, uint32_t sb,
                     uint64_t *out) {
    u32 k = c->k, n = c->n, nb = c->nb, so = sa + sb - 1;
    size_t pq = (size_t)k * n, pb = (size_t)nb * n;
    u64 *aq = (u64 *)malloc(sizeof(u64) * pq * sa), *bq = (u64 *)malloc(sizeof(u64) * pq * sb);
    u64 *ab = (u64 *)malloc(sizeof(u64) * pb * sa), *bb = (u64 *)malloc(sizeof(u64) * pb * sb);
    u64 *dq = (u64 *)calloc(pq * so, sizeof(u64)), *db = (u64 *)calloc(pb * so, sizeof(u64));
    memcpy(aq, a, sizeof(u64) * pq * sa);
    memcpy(bq, b, sizeof(u64) * pq * sb);
    for (u32 j = 0; j < sa; j++) behz_to_bsk(c, a + pq * j, ab + pb * j);
    for (u32 j = 0; j < sb; j++) behz_to_bsk(c, b + pq * j, bb + pb * j);
    for (u32 j = 0; j < sa; j++) {
        for (u32 i = 0; i < k; i++) ntt_fwd(&c->qt[i], n, aq + pq * j + (size_t)i * n);
        for (u32 i = 0; i < nb; i++) ntt_fwd(&c->bt[i], n, ab + pb * j + (size_t)i * n);
    }
    for (u32 j = 0; j < sb; j++) {
        for (u32 i = 0; i < k; i++) ntt_fwd(&c->qt[i], n, bq + pq * j + (size_t)i * n);
        for (u32 i = 0; i < nb; i++) ntt_fwd(&c->bt[i], n, bb + pb * j + (size_t)i * n);
    }
    for (u32 ja = 0; ja < sa; ja++)
        for (u32 jb = 0; jb < sb; jb++) {
            u32 o = ja + jb;
            for (u32 i = 0; i < k; i++) {
                u64 *d = dq + pq * o + (size_t)i * n;
                const u64 *x = aq + pq * ja + (size_t)i * n, *y = bq + pq * jb + (size_t)i * n;
                for (u32 l = 0; l < n; l++) d[l] = addmod(d[l], mulmod(x[l], y[l], c->q[i]), c->q[i]);
            }
            for (u32 i = 0; i < nb; i++) {
                u64 *d = db + pb * o + (size_t)i * n;
                const u64 *x = ab + pb * ja + (size_t)i * n, *y = bb + pb * jb + (size_t)i * n;
                for (u32 l = 0; l < n; l++) d[l] = addmod(d[l], mulmod(x[l], y[l], c->bsk[i]), c->bsk[i]);
            }
        }
    for (u32 o = 0; o < so; o++) {
        for (u32 i = 0; i < k; i++) {
            u64 *d = dq + pq * o + (size_t)i * n;
            ntt_inv(&c->qt[i], n, d);
            for (u32 l = 0; l < n; l++) d[l] = mulmod(d[l], c->t % c->q[i], c->q[i]);
        }
        for (u32 i = 0; i < nb; i++) {
            u64 *d = db + pb * o + (size_t)i * n;
            ntt_inv(&c->bt[i], n, d);
            for (u32 l = 0; l < n; l++) d[l] = mulmod(d[l], c->t % c->bsk[i], c->bsk[i]);
        }
        behz_floor_and_back(c, dq + pq * o, db + pb * o, out + pq * o);
    }
    free(aq); free(bq); free(ab); free(bb); free(dq); free(db);
    return so;
}

uint32_t fo_square(const fo_ctx *c, const uint64_t *a, uint32_t sa, uint64_t *out) {
    /* SEAL special-cases size 2 as (c0^2, 2 c0 c1, c1^2); as ring elements that is the
     * same tensor the generic product forms, so the result is identical. */
    return fo_multiply(c, a, sa, a, sa, out);
}

/* ------------------------------------------------------------------------- */
/* FractionalEncoder, base 2                              */
/* ------------------------------------------------------------------------- */
uint32_t fo_frac_encode(const fo_ctx *c, double v, int int_coeffs, int frac_coeffs, uint64_t *plain) {
    u32 n = c->n;
    u64 t = c->t;
    memset(plain, 0, sizeof(u64) * n);
    int64_t ip = (int64_t)v; /* truncation toward zero */
    double f = v - (double)ip;
    int neg = v < 0;
    u64 mag = ip < 0 ? (u64)(-ip) : (u64)ip;
    for (int d = 0; mag && d < int_coeffs; d++, mag >>= 1)
        if (mag & 1) plain[d] = (ip < 0) ? t - 1 : 1;
    if (f != 0.0) {
        for (int i = 1; i <= frac_coeffs; i++) {
            f *= 2.0;
            int64_t b = (int64_t)f;
            f -= (double)b;
            if (b && (u32)i > n) return 0xffffffffu; /* the digit does not fit the ring (test-sized n): caller raises */
            if (b) plain[n - i] = neg ? 1 : t - 1; /* -x^(n-i) == x^(-i) */
        }
    }
    u32 len = n;
    while (len > 0 && plain[len - 1] == 0) len--;
    return len;
}

double fo_frac_decode(const fo_ctx *c, const uint64_t *plain, int int_coeffs, int frac_coeffs) {
    (void)frac_coeffs; /* everything above the integer part is read as fraction */
    u32 n = c->n;
    u64 t = c->t, thr = (t + 1) >> 1;
    double ip = 0.0;
    for (int d = int_coeffs - 1; d >= 0; d--) {
        double cv = plain[d] >= thr ? -(double)(t - plain[d]) : (double)plain[d];
        ip = ip * 2.0 + cv;
    }
    double fr = 0.0;
    for (u32 idx = (u32)int_coeffs; idx < n; idx++) { /* Horner from the lowest fractional digit */
        double cv = plain[idx] >= thr ? -(double)(t - plain[idx]) : (double)plain[idx];
        fr = (fr + cv) / 2.0;
    }
    return ip - fr;
}

/* ------------------------------------------------------------------------- */
/* keys, encrypt, decrypt (test scaffolding; textbook BFV) */
/* ------------------------------------------------------------------------- */
typedef struct { u64 s; } rng;
static u64 rng_next(rng *r) { r->s += 0x9E3779B97F4A7C15ULL; return fo_splitmix64(r->s); }
static double rng_unit(rng *r) { return ((rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static int64_t rng_noise(rng *r) { /* clipped normal, sigma 3.19, |e| <= 6 sigma */
    for (;;) {
        double u1 = rng_unit(r), u2 = rng_unit(r);
        double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * 3.19;
        if (fabs(g) <= 19.14) return (int64_t)llround(g);
    }
}
static void sample_ternary(const fo_ctx *c, rng *r, u64 *out /* [k][n] */) {
    for (u32 l = 0; l < c->n; l++) {
        u64 v = rng_next(r) % 3; /* 0,1,2 -> 0,1,-1 */
        for (u32 i = 0; i < c->k; i++) out[(size_t)i * c->n + l] = v == 2 ? c->q[i] - 1 : v;
    }
}
static void sample_noise(const fo_ctx *c, rng *r, u64 *out) {
    for (u32 l = 0; l < c->n; l++) {
        int64_t e = rng_noise(r);
        for (u32 i = 0; i < c->k; i++) out[(size_t)i * c->n + l] = e < 0 ? c->q[i] - (u64)(-e) : (u64)e;
    }
}
static void sample_uniform(const fo_ctx *c, rng *r, u64 *out) {
    for (u32 i = 0; i < c->k; i++)
        for (u32 l = 0; l < c->n; l++) out[(size_t)i * c->n + l] = rng_next(r) % c->q[i];
}
/* out = a * b in R_q (all [k][n], coefficient form) */
static void ring_mul(const fo_ctx *c, const u64 *a, const u64 *b, u64 *out) {
    u32 n = c->n;
    u64 *x = (u64 *)malloc(sizeof(u64) * n), *y = (u64 *)malloc(sizeof(u64) * n);
    for (u32 i = 0; i < c->k; i++) {
        memcpy(x, a + (size_t)i * n, sizeof(u64) * n);
        memcpy(y, b + (size_t)i * n, sizeof(u64) * n);
        ntt_fwd(&c->qt[i], n, x);
        ntt_fwd(&c->qt[i], n, y);
        for (u32 l = 0; l < n; l++) x[l] = mulmod(x[l], y[l], c->q[i]);
        ntt_inv(&c->qt[i], n, x);
        memcpy(out + (size_t)i * n, x, sizeof(u64) * n);
    }
    free(x); free(y);
}

void fo_keygen(const fo_ctx *c, uint64_t seed, uint64_t *sk, uint64_t *pk) {
    rng r = {seed};
    size_t pq = (size_t)c->k * c->n;
    u64 *e = (u64 *)malloc(sizeof(u64) * pq);
    sample_ternary(c, &r, sk);
    sample_uniform(c, &r, pk + pq); /* a */
    sample_noise(c, &r, e);
    ring_mul(c, pk + pq, sk, pk); /* a*s */
    for (u32 i = 0; i < c->k; i++)
        for (u32 l = 0; l < c->n; l++) {
            size_t x = (size_t)i * c->n + l;
            pk[x] = negmod(addmod(pk[x], e[x], c->q[i]), c->q[i]); /* -(a s + e) */
        }
    free(e);
}

void fo_encrypt(const fo_ctx *c, const uint64_t *pk, const uint64_t *plain, uint32_t len, uint64_t seed,
                uint64_t *ct) {
    rng r = {seed ^ 0xC1F3E57A11ULL};
    size_t pq = (size_t)c->k * c->n;
    u64 *u = (u64 *)malloc(sizeof(u64) * pq), *e = (u64 *)malloc(sizeof(u64) * pq);
    sample_ternary(c, &r, u);
    for (u32 j = 0; j < 2; j++) {
        ring_mul(c, pk + pq * j, u, ct + pq * j);
        sample_noise(c, &r, e);
        for (u32 i = 0; i < c->k; i++)
            for (u32 l = 0; l < c->n; l++) {
                size_t x = (size_t)i * c->n + l;
                ct[pq * j + x] = addmod(ct[pq * j + x], e[x], c->q[i]);
            }
    }
    fo_add_plain(c, ct, plain, len);
    free(u); free(e);
}

/* ---- the keyed sampler of include/fhe_hip.h ("server-side encryptions"), restated -------------------------------------
 * ChaCha20 in D. J. Bernstein's original layout (256-bit key, 64-bit block counter, 64-bit nonce), the published algorithm:
 * state = "expand 32-byte k" | key | counter | nonce as sixteen little-endian words, ten double rounds of the quarter round
 * (a += b; d ^= a; d <<<= 16; c += d; b ^= c; b <<<= 12; a += b; d ^= a; d <<<= 8; c += d; b ^= c; b <<<= 7) on columns then
 * diagonals, output = state + input.  tests/test_encrypt_sampler.py pins it against the published all-zero-key block. */
#define FO_ROTL(x, r) (((x) << (r)) | ((x) >> (32 - (r))))
#define FO_QR(a, b, c, d) a += b; d ^= a; d = FO_ROTL(d, 16); c += d; b ^= c; b = FO_ROTL(b, 12); a += b; d ^= a; d = FO_ROTL(d, 8); c += d; b ^= c; b = FO_ROTL(b, 7);
void fo_chacha20_block(const uint8_t key[32], uint64_t counter, uint64_t nonce, uint8_t out[64]) {
    uint32_t in[16], x[16];
    in[0] = 0x61707865u; in[1] = 0x3320646eu; in[2] = 0x79622d32u; in[3] = 0x6b206574u;
    for (int i = 0; i < 8; i++)
        in[4 + i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = (uint32_t)nonce; in[15] = (uint32_t)(nonce >> 32);
    memcpy(x, in, sizeof x);
    for (int r = 0; r < 10; r++) {
        FO_QR(x[0], x[4], x[8], x[12]) FO_QR(x[1], x[5], x[9], x[13]) FO_QR(x[2], x[6], x[10], x[14]) FO_QR(x[3], x[7], x[11], x[15])
        FO_QR(x[0], x[5], x[10], x[15]) FO_QR(x[1], x[6], x[11], x[12]) FO_QR(x[2], x[7], x[8], x[13]) FO_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) {
        uint32_t v = x[i] + in[i];
        out[4 * i] = (uint8_t)v; out[4 * i + 1] = (uint8_t)(v >> 8); out[4 * i + 2] = (uint8_t)(v >> 16); out[4 * i + 3] = (uint8_t)(v >> 24);
    }
}
/* floor(2^63 P(|e| <= i)) for the rounded normal, sigma 3.19, redrawn beyond 19 (tools/noise_cdt.py) */
static const u64 fo_cdt[19] = {
    0x0ff141e3023416d2ULL, 0x2e4f850f76b8d9a6ULL, 0x488c5acec8fd6db3ULL, 0x5d1ca569fc3e4ccbULL, 0x6bbb5699bdd65b9cULL, 0x75291bf8371e7eccULL,
    0x7aad3cf138611a69ULL, 0x7d9aa4d4ab7c76bdULL, 0x7f0368341f79807cULL, 0x7fa0f21e3a554470ULL, 0x7fdf5971c6494be2ULL, 0x7ff5c5a33f74a4e1ULL,
    0x7ffd148ddcc40605ULL, 0x7fff3db0052c58c3ULL, 0x7fffd206471c7fcfULL, 0x7ffff61ba7b56e58ULL, 0x7ffffe11d76ecb8aULL, 0x7fffffa9c1e61510ULL,
    0x7ffffff3ceaa701fULL};
void fo_noise_cdt(uint64_t out[19]) { memcpy(out, fo_cdt, sizeof fo_cdt); }
/* the draws of encryption number `index` under `key`: draws [3][n] = u (ternary), e1, e2; 64-bit draw d = bytes [8 d, 8 d + 8) of the
 * stream with nonce `index`; d = j, n + j, 2 n + j */
void fo_encrypt_draws(uint32_t n, const uint8_t key[32], uint64_t index, int8_t *draws) {
    uint8_t blk[64];
    for (u64 d = 0; d < (u64)3 * n; d++) {
        if (d % 8 == 0) fo_chacha20_block(key, d / 8, index, blk);
        u64 r = 0;
        for (int b = 7; b >= 0; b--) r = (r << 8) | blk[8 * (d % 8) + b];
        if (d < n) {
            draws[d] = (int8_t)((int)(u64)(((unsigned __int128)r * 3) >> 64) - 1);
        } else {
            u64 x = r >> 1;
            int m = 0;
            for (int i = 0; i < 19; i++) m += x >= fo_cdt[i];
            draws[d] = (int8_t)((r & 1) ? -m : m);
        }
    }
}
/* Enc(m) = (Delta m' + pk0 u + e1, pk1 u + e2) from given draws; pk [2][k][n] in coefficient form */
void fo_encrypt_with_draws(const fo_ctx *c, const uint64_t *pk, const uint64_t *plain, uint32_t len, const int8_t *draws, uint64_t *ct) {
    size_t pq = (size_t)c->k * c->n;
    u64 *u = (u64 *)malloc(sizeof(u64) * pq);
    for (u32 i = 0; i < c->k; i++)
        for (u32 l = 0; l < c->n; l++) u[(size_t)i * c->n + l] = draws[l] < 0 ? c->q[i] - (u64)(-draws[l]) : (u64)draws[l];
    for (u32 j = 0; j < 2; j++) {
        ring_mul(c, pk + pq * j, u, ct + pq * j);
        const int8_t *e = draws + (size_t)(1 + j) * c->n;
        for (u32 i = 0; i < c->k; i++)
            for (u32 l = 0; l < c->n; l++) {
                size_t x = pq * j + (size_t)i * c->n + l;
                ct[x] = addmod(ct[x], e[l] < 0 ? c->q[i] - (u64)(-e[l]) : (u64)e[l], c->q[i]);
            }
    }
    if (len) fo_add_plain(c, ct, plain, len);
    free(u);
}
void fo_encrypt_keyed(const fo_ctx *c, const uint64_t *pk, const uint64_t *plain, uint32_t len, const uint8_t key[32], uint64_t index, uint64_t *ct) {
    int8_t *draws = (int8_t *)malloc((size_t)3 * c->n);
    fo_encrypt_draws(c->n, key, index, draws);
    fo_encrypt_with_draws(c, pk, plain, len, draws, ct);
    free(draws);
}

void fo_decrypt_phase(const fo_ctx *c, const uint64_t *sk, const uint64_t *ct, uint32_t size,
                      uint64_t *phase) {
    u32 n = c->n;
    u64 *s = (u64 *)malloc(sizeof(u64) * n), *acc = (u64 *)malloc(sizeof(u64) * n),
        *x = (u64 *)malloc(sizeof(u64) * n);
    for (u32 i = 0; i < c->k; i++) {
        u64 qi = c->q[i];
        memcpy(s, sk + (size_t)i * n, sizeof(u64) * n);
        ntt_fwd(&c->qt[i], n, s);
        /* Horner in s: acc = c_{size-1}; acc = acc*s + c_j */
        memcpy(acc, POLY(ct, size - 1, i), sizeof(u64) * n);
        ntt_fwd(&c->qt[i], n, acc);
        for (int j = (int)size - 2; j >= 0; j--) {
            memcpy(x, POLY(ct, j, i), sizeof(u64) * n);
            ntt_fwd(&c->qt[i], n, x);
            for (u32 l = 0; l < n; l++) acc[l] = addmod(mulmod(acc[l], s[l], qi), x[l], qi);
        }
        ntt_inv(&c->qt[i], n, acc);
        memcpy(phase + (size_t)i * n, acc, sizeof(u64) * n);
    }
    free(s); free(acc); free(x);
}

int fo_decrypt_noise_bits(const fo_ctx *c, const uint64_t *sk, const uint64_t *ct, uint32_t size, uint64_t *plain, int *noise_bits, int *modulus_bits);
int fo_decrypt(const fo_ctx *c, const uint64_t *sk, const uint64_t *ct, uint32_t size, uint64_t *plain) {
    return fo_decrypt_noise_bits(c, sk, ct, size, plain, NULL, NULL);
}
/* the same with the raw figures: *noise_bits = bit length of the largest |t x - m q|, *modulus_bits = bit length of q */
int fo_decrypt_noise_bits(const fo_ctx *c, const uint64_t *sk, const uint64_t *ct, uint32_t size, uint64_t *plain, int *noise_bits, int *modulus_bits) {
    u32 n = c->n, k = c->k;
    u64 *phase = (u64 *)malloc(sizeof(u64) * (size_t)k * n);
    fo_decrypt_phase(c, sk, ct, size, phase);
    int max_noise_bits = 0;
    for (u32 l = 0; l < n; l++) {
        big x;
        big_zero(&x);
        for (u32 i = 0; i < k; i++) {
            u64 y = mulmod(phase[(size_t)i * n + l], c->inv_punct[i], c->q[i]);
            big term;
            big_mul_small(&term, &c->punct_big[i], y);
            big_add(&x, &term);
        }
        while (big_cmp(&x, &c->qbig) >= 0) big_sub(&x, &c->qbig);
        /* m = floor((t x + floor(q/2)) / q) mod t, by binary search on the quotient */
        big tx;
        big_mul_small(&tx, &x, c->t);
        big num = tx;
        big_add(&num, &c->qhalf);
        u64 lo = 0, hi = c->t; /* quotient in [0, t] */
        while (lo < hi) {
            u64 mid = lo + (hi - lo + 1) / 2;
            big prod;
            big_mul_small(&prod, &c->qbig, mid);
            if (big_cmp(&prod, &num) <= 0) lo = mid; else hi = mid - 1;
        }
        plain[l] = lo % c->t;
        /* invariant noise: |t x - lo q| */
        big prod, diff;
        big_mul_small(&prod, &c->qbig, lo);
        if (big_cmp(&tx, &prod) >= 0) { diff = tx; big_sub(&diff, &prod); }
        else { diff = prod; big_sub(&diff, &tx); }
        int nb_ = big_bits(&diff);
        if (nb_ > max_noise_bits) max_noise_bits = nb_;
    }
    free(phase);
    if (noise_bits) *noise_bits = max_noise_bits;
    if (modulus_bits) *modulus_bits = big_bits(&c->qbig);
    int budget = big_bits(&c->qbig) - max_noise_bits - 1;
    return budget < 0 ? 0 : budget;
}

/* ------------------------------------------------------------------------- */
/* evaluation keys + relinearisation                      */
/* ------------------------------------------------------------------------- */
static u32 bits_u64(u64 v) { return v ? 64 - (u32)__builtin_clzll(v) : 0; }
uint32_t fo_evk_digits(const fo_ctx *c, uint32_t dbc) {
    u32 mx = 0;
    for (u32 i = 0; i < c->k; i++) {
        u32 d = (bits_u64(c->q[i]) + dbc - 1) / dbc;
        if (d > mx) mx = d;
    }
    return mx;
}

void fo_evk_gen(const fo_ctx *c, const uint64_t *sk, uint32_t dbc, uint64_t seed, uint64_t *evk) { fo_evk_gen_pow(c, sk, 2, dbc, seed, evk); }
/* keys for s^power (SEAL 2.3 generate_evaluation_keys(dbc, count, keys) makes them for s^2 .. s^(count+1); the reference only
 * ever asks for count = 1, tests/parameters.cpp:89): the same construction with s^power in the place of s^2; the sampler is
 * seeded by (seed, power) so that the keys of different powers are independent and power = 2 is fo_evk_gen */
void fo_evk_gen_pow(const fo_ctx *c, const uint64_t *sk, uint32_t power, uint32_t dbc, uint64_t seed, uint64_t *evk) {
    u32 n = c->n, k = c->k, nd = fo_evk_digits(c, dbc);
    size_t pq = (size_t)k * n;
    rng r = {(seed ^ 0xE7A1BEEF5ULL) + 0x9E3779B97F4A7C15ULL * (u64)(power - 2)};
    u64 *s2 = (u64 *)malloc(sizeof(u64) * pq), *a = (u64 *)malloc(sizeof(u64) * pq),
        *e = (u64 *)malloc(sizeof(u64) * pq), *as = (u64 *)malloc(sizeof(u64) * pq);
    ring_mul(c, sk, sk, s2);
    for (u32 pw = 2; pw < power; pw++) {                     /* s^power */
        ring_mul(c, s2, sk, as);
        memcpy(s2, as, sizeof(u64) * pq);
    }
    for (u32 i = 0; i < k; i++)
        for (u32 d = 0; d < nd; d++) {
            u64 *k0 = evk + (((size_t)i * nd + d) * 2 + 0) * pq;
            u64 *k1 = evk + (((size_t)i * nd + d) * 2 + 1) * pq;
            sample_uniform(c, &r, a);
            sample_noise(c, &r, e);
            ring_mul(c, a, sk, as);
            for (u32 ii = 0; ii < k; ii++)
                for (u32 l = 0; l < n; l++) {
                    size_t x = (size_t)ii * n + l;
                    k0[x] = negmod(addmod(as[x], e[x], c->q[ii]), c->q[ii]);
                    k1[x] = a[x];
                }
            /* + w^d * s^2 in RNS component i only (CRT idempotent E_i) */
            u64 wd = powmod(2, (u64)dbc * d, c->q[i]);
            for (u32 l = 0; l < n; l++) {
                size_t x = (size_t)i * n + l;
                k0[x] = addmod(k0[x], mulmod(s2[x], wd, c->q[i]), c->q[i]);
            }
            for (u32 ii = 0; ii < k; ii++) {
                ntt_fwd(&c->qt[ii], n, k0 + (size_t)ii * n);
                ntt_fwd(&c->qt[ii], n, k1 + (size_t)ii * n);
            }
        }
    free(s2); free(a); free(e); free(as);
}

void fo_relinearize3(const fo_ctx *c, uint64_t *ct, const uint64_t *evk, uint32_t dbc) { fo_relinearize_poly(c, ct, 2, evk, dbc); }
/* one key-switch step (SEAL 2.3 relinearize_one_step): the LAST polynomial `src_poly` of a ciphertext of src_poly + 1 polynomials is
 * decomposed and folded into c0 / c1 with the keys for s^src_poly; the polynomials in between stay */
void fo_relinearize_poly(const fo_ctx *c, uint64_t *ct, uint32_t src_poly, const uint64_t *evk, uint32_t dbc) {
    u32 n = c->n, k = c->k, nd = fo_evk_digits(c, dbc);
    size_t pq = (size_t)k * n;
    u64 mask = (dbc >= 64) ? ~0ULL : ((1ULL << dbc) - 1);
    u64 *acc0 = (u64 *)calloc(pq, sizeof(u64)), *acc1 = (u64 *)calloc(pq, sizeof(u64));
    u64 *dig = (u64 *)malloc(sizeof(u64) * n);
    for (u32 i = 0; i < k; i++) {
        const u64 *c2 = POLY(ct, src_poly, i);
        for (u32 d = 0; d < nd; d++) {
            const u64 *k0 = evk + (((size_t)i * nd + d) * 2 + 0) * pq;
            const u64 *k1 = evk + (((size_t)i * nd + d) * 2 + 1) * pq;
            for (u32 ii = 0; ii < k; ii++) {
                for (u32 l = 0; l < n; l++) dig[l] = ((c2[l] >> (dbc * d)) & mask) % c->q[ii];
                ntt_fwd(&c->qt[ii], n, dig);
                u64 *a0 = acc0 + (size_t)ii * n, *a1 = acc1 + (size_t)ii * n;
                const u64 *e0 = k0 + (size_t)ii * n, *e1 = k1 + (size_t)ii * n;
                for (u32 l = 0; l < n; l++) {
                    a0[l] = addmod(a0[l], mulmod(dig[l], e0[l], c->q[ii]), c->q[ii]);
                    a1[l] = addmod(a1[l], mulmod(dig[l], e1[l], c->q[ii]), c->q[ii]);
                }
            }
        }
    }
    for (u32 ii = 0; ii < k; ii++) {
        ntt_inv(&c->qt[ii], n, acc0 + (size_t)ii * n);
        ntt_inv(&c->qt[ii], n, acc1 + (size_t)ii * n);
        u64 *x0 = POLY(ct, 0, ii), *x1 = POLY(ct, 1, ii);
        for (u32 l = 0; l < n; l++) {
            x0[l] = addmod(x0[l], acc0[(size_t)ii * n + l], c->q[ii]);
            x1[l] = addmod(x1[l], acc1[(size_t)ii * n + l], c->q[ii]);
        }
    }
    free(acc0); free(acc1); free(dig);
}

/* ------------------------------------------------------------------------- */
/* circuits, one Evaluator call at a time                                      */
/* ------------------------------------------------------------------------- */
#define ENC_INT 100
#define ENC_FRAC 100

static void mp_const(const fo_ctx *c, u64 *ct, u32 size, double v) {
    u64 *p = (u64 *)malloc(sizeof(u64) * c->n);
    u32 len = fo_frac_encode(c, v, ENC_INT, ENC_FRAC, p);
    fo_multiply_plain(c, ct, size, p, len);
    free(p);
}

/* One 1-D pass of homo/fhe_image.h:206-244 (rows, scale == 0) or :246-284 (columns, every
 * output additionally multiplied by encode(0.125)).  d[i] points at the i-th ciphertext of
 * the line; same add/sub/multiply_plain sequence, same constants, same association. */
static void dct_line(const fo_ctx *c, u64 *d[8], int scale) {
    size_t ctw = (size_t)2 * c->k * c->n, bytes = sizeof(u64) * ctw;
    u64 *buf = (u64 *)malloc(bytes * 20);
    u64 *tmp0 = buf, *tmp1 = buf + ctw, *tmp2 = buf + 2 * ctw, *tmp3 = buf + 3 * ctw, *tmp4 = buf + 4 * ctw,
        *tmp5 = buf + 5 * ctw, *tmp6 = buf + 6 * ctw, *tmp7 = buf + 7 * ctw, *tmp10 = buf + 8 * ctw,
        *tmp11 = buf + 9 * ctw, *tmp12 = buf + 10 * ctw, *tmp13 = buf + 11 * ctw, *z1 = buf + 12 * ctw,
        *z2 = buf + 13 * ctw, *z3 = buf + 14 * ctw, *z4 = buf + 15 * ctw, *z5 = buf + 16 * ctw,
        *w = buf + 17 * ctw;
#define CP(dst, src) memcpy(dst, src, bytes)
#define ADD(dst, x, y) do { CP(w, x); fo_add(c, w, 2, y, 2); CP(dst, w); } while (0)
#define SUB(dst, x, y) do { CP(w, x); fo_sub(c, w, 2, y, 2); CP(dst, w); } while (0)
#define MP(dst, x, v) do { CP(w, x); mp_const(c, w, 2, v); CP(dst, w); } while (0)
#define OUT(idx, x) do { if (scale) { CP(w, x); mp_const(c, w, 2, 0.125); CP(d[idx], w); } else CP(d[idx], x); } while (0)
    ADD(tmp0, d[0], d[7]); SUB(tmp7, d[0], d[7]);
    ADD(tmp1, d[1], d[6]); SUB(tmp6, d[1], d[6]);
    ADD(tmp2, d[2], d[5]); SUB(tmp5, d[2], d[5]);
    ADD(tmp3, d[3], d[4]); SUB(tmp4, d[3], d[4]);
    ADD(tmp10, tmp0, tmp3); SUB(tmp13, tmp0, tmp3);
    ADD(tmp11, tmp1, tmp2); SUB(tmp12, tmp1, tmp2);
    ADD(z5, tmp10, tmp11); OUT(0, z5);
    SUB(z5, tmp10, tmp11); OUT(4, z5);
    ADD(z1, tmp12, tmp13); MP(z1, z1, 0.541196100);
    MP(z2, tmp13, 0.765366865); ADD(z2, z1, z2); OUT(2, z2);
    MP(z2, tmp12, -1.847759065); ADD(z2, z1, z2); OUT(6, z2);
    ADD(z1, tmp4, tmp7); ADD(z2, tmp5, tmp6); ADD(z3, tmp4, tmp6); ADD(z4, tmp5, tmp7);
    ADD(z5, z3, z4); MP(z5, z5, 1.175875602);
    MP(tmp4, tmp4, 0.298631336); MP(tmp5, tmp5, 2.053119869);
    MP(tmp6, tmp6, 3.072711026); MP(tmp7, tmp7, 1.501321110);
    MP(z1, z1, -0.899976223); MP(z2, z2, -2.562915447);
    MP(z3, z3, -1.961570560); MP(z4, z4, -0.390180644);
    ADD(z3, z3, z5); ADD(z4, z4, z5);
    ADD(tmp10, tmp4, z1); ADD(tmp10, tmp10, z3); OUT(7, tmp10);
    ADD(tmp10, tmp5, z2); ADD(tmp10, tmp10, z4); OUT(5, tmp10);
    ADD(tmp10, tmp6, z2); ADD(tmp10, tmp10, z3); OUT(3, tmp10);
    ADD(tmp10, tmp7, z1); ADD(tmp10, tmp10, z4); OUT(1, tmp10);
#undef CP
#undef ADD
#undef SUB
#undef MP
#undef OUT
    free(buf);
}

void fo_encrypted_dct(const fo_ctx *c, uint64_t *data) {
    size_t ctw = (size_t)2 * c->k * c->n;
    u64 *d[8];
    for (int r = 0; r < 8; r++) { /* rows: indices 8r .. 8r+7 */
        for (int i = 0; i < 8; i++) d[i] = data + ctw * (size_t)(8 * r + i);
        dct_line(c, d, 0);
    }
    for (int col = 0; col < 8; col++) { /* columns: indices col + 8 i, each output x encode(0.125) */
        for (int i = 0; i < 8; i++) d[i] = data + ctw * (size_t)(col + 8 * i);
        dct_line(c, d, 1);
    }
}

void fo_quantize(const fo_ctx *c, uint64_t *data, const double *quant) {
    size_t ctw = (size_t)2 * c->k * c->n;
    for (int i = 0; i < 64; i++) mp_const(c, data + ctw * (size_t)i, 2, 1 / quant[i]);
}

/* encrypted_dct + quantize_fhe on n_blocks independent blocks, OpenMP over blocks: the "all cores" CPU
 * baseline of bench.py (the reference's loop, homo/server_jpeg.cpp:113, is serial; blocks are independent).
 * Returns the number of threads used. */
int fo_dct_quant_blocks(const fo_ctx *c, uint64_t *data, uint32_t n_blocks, const double *quant) {
    const size_t blk = (size_t)64 * 2 * c->k * c->n;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (uint32_t b = 0; b < n_blocks; b++) {
        fo_encrypted_dct(c, data + blk * b);
        fo_quantize(c, data + blk * b, quant);
    }
    return threads;
}

void fo_rgb_to_ycc(const fo_ctx *c, uint64_t *r, uint64_t *g, uint64_t *b) {
    size_t ctw = (size_t)2 * c->k * c->n, bytes = sizeof(u64) * ctw;
    u64 *buf = (u64 *)malloc(bytes * 6);
    u64 *y = buf, *u = buf + ctw, *v = buf + 2 * ctw, *x1 = buf + 3 * ctw, *x2 = buf + 4 * ctw;
    u64 *p = (u64 *)malloc(sizeof(u64) * c->n);
    /* Y = .299 R + .587 G + .114 B - 128 */
    memcpy(y, r, bytes); mp_const(c, y, 2, 0.299);
    memcpy(x1, g, bytes); mp_const(c, x1, 2, 0.587); fo_add(c, y, 2, x1, 2);
    memcpy(x2, b, bytes); mp_const(c, x2, 2, 0.114); fo_add(c, y, 2, x2, 2);
    u32 len = fo_frac_encode(c, 128.0, ENC_INT, ENC_FRAC, p);
    fo_sub_plain(c, y, p, len);
    /* Cb = -.168736 R - .331264 G + .5 B */
    memcpy(u, r, bytes); mp_const(c, u, 2, -0.168736);
    memcpy(x1, g, bytes); mp_const(c, x1, 2, 0.331264); fo_sub(c, u, 2, x1, 2);
    memcpy(x2, b, bytes); mp_const(c, x2, 2, 0.5); fo_add(c, u, 2, x2, 2);
    /* Cr = .5 R - .418688 G - .081312 B */
    memcpy(v, r, bytes); mp_const(c, v, 2, 0.5);
    memcpy(x1, g, bytes); mp_const(c, x1, 2, 0.418688); fo_sub(c, v, 2, x1, 2);
    memcpy(x2, b, bytes); mp_const(c, x2, 2, 0.081312); fo_sub(c, v, 2, x2, 2);
    memcpy(r, y, bytes); memcpy(g, u, bytes); memcpy(b, v, bytes);
    free(buf); free(p);
}

/* homo/fhe_resize.h:143-189.  Note the reference's t3 = t*t (sic, :175). */
uint32_t fo_cubic(const fo_ctx *c, const uint64_t *A, const uint64_t *B, const uint64_t *C,
                  const uint64_t *D, uint32_t s, const uint64_t *t, uint64_t *result) {
    size_t pq = (size_t)c->k * c->n;
    size_t cap = pq * (s + 2), bytes_s = sizeof(u64) * pq * s;
    u64 *a = (u64 *)calloc(cap, sizeof(u64)), *b = (u64 *)calloc(cap, sizeof(u64)),
        *cc = (u64 *)calloc(cap, sizeof(u64)), *x = (u64 *)calloc(cap, sizeof(u64)),
        *t2 = (u64 *)calloc(pq * 3, sizeof(u64)), *t3 = (u64 *)calloc(pq * 3, sizeof(u64)),
        *prod = (u64 *)calloc(cap, sizeof(u64));
    /* a = 3B - A - 3C + D */
    memcpy(a, B, bytes_s); mp_const(c, a, s, 3); fo_sub(c, a, s, A, s);
    memcpy(x, C, bytes_s); mp_const(c, x, s, 3); fo_sub(c, a, s, x, s);
    fo_add(c, a, s, D, s);
    /* b = 2A - 5B + 4C - D */
    memcpy(b, A, bytes_s); mp_const(c, b, s, 2);
    memcpy(x, B, bytes_s); mp_const(c, x, s, 5); fo_sub(c, b, s, x, s);
    memcpy(x, C, bytes_s); mp_const(c, x, s, 4); fo_add(c, b, s, x, s);
    fo_sub(c, b, s, D, s);
    /* c = C - A ; d = B */
    memcpy(cc, C, bytes_s); fo_sub(c, cc, s, A, s);
    fo_square(c, t, 2, t2);
    fo_multiply(c, t, 2, t, 2, t3);
    u32 sa = fo_multiply(c, a, s, t3, 3, prod); memcpy(a, prod, sizeof(u64) * pq * sa);
    u32 sb = fo_multiply(c, b, s, t2, 3, prod); memcpy(b, prod, sizeof(u64) * pq * sb);
    u32 sc = fo_multiply(c, cc, s, t, 2, prod); memcpy(cc, prod, sizeof(u64) * pq * sc);
    sa = fo_add(c, a, sa, b, sb);
    sa = fo_add(c, a, sa, cc, sc);
    mp_const(c, a, sa, 0.5);
    sa = fo_add(c, a, sa, B, s);
    memcpy(result, a, sizeof(u64) * pq * sa);
    free(a); free(b); free(cc); free(x); free(t2); free(t3); free(prod);
    return sa;
}

/* homo/fhe_resize.h:191-204: (1 - t) A + t B */
uint32_t fo_linear(const fo_ctx *c, const uint64_t *A, const uint64_t *B, uint32_t s, const uint64_t *t,
                   uint64_t *result) {
    size_t pq = (size_t)c->k * c->n;
    u64 *omt = (u64 *)malloc(sizeof(u64) * pq * 2), *p = (u64 *)malloc(sizeof(u64) * c->n);
    u64 *x = (u64 *)calloc(pq * (s + 1), sizeof(u64));
    memcpy(omt, t, sizeof(u64) * pq * 2);
    fo_negate(c, omt, 2);
    u32 len = fo_frac_encode(c, 1.0, ENC_INT, ENC_FRAC, p);
    fo_add_plain(c, omt, p, len);
    u32 s1 = fo_multiply(c, omt, 2, A, s, result);
    u32 s2 = fo_multiply(c, B, s, t, 2, x);
    s1 = fo_add(c, result, s1, x, s2);
    free(omt); free(p); free(x);
    return s1;
}

uint64_t fo_digest(const uint64_t *p, uint64_t count) {
    u64 h = 0x243F6A8885A308D3ULL;
    for (u64 i = 0; i < count; i++) h = fo_splitmix64(h ^ p[i]) + i;
    return h;
}
