// circuits.hip -- the reference's ciphertext x ciphertext circuits on whole batches, behind the C ABI of
// include/fhe_circuits.h: Cubic / Linear / SampleBicubic / SampleLinear / ResizeImage (homo/fhe_resize.h:143-392),
// homomorphic_sin / homomorphic_cos / approximated_step (homo/fhe_decode.h:48-242) and the per-channel driver loop of
// homo/server_decode.cpp:120-137.
//
// Every function here is host orchestration over the library's own primitives (fhe_hip.h) plus four small
// gather / scatter kernels; all temporaries come from a bump allocator over the caller's scratch.  Each circuit is
// written ONCE against `Run`: the `*_scratch_bytes` queries execute the same code with launches suppressed and
// report the allocator's high-water mark, so the figure can never drift from what the circuit needs.
#include "internal.h"
#include "../../include/fhe_circuits.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <set>

#include "host_math.h"

namespace {

// ------------------------------------------------------------------------------------------------
// kernels: index maps, gathers, unequal-size additions, Cubic's linear parts through index arrays
// ------------------------------------------------------------------------------------------------
// CMap (internal.h): which ciphertext of a batch pair / output `c` refers to -- an explicit index array, the periodic map
// (off + c / div) % cnt, or c itself
inline CMap ident() { return CMap{nullptr, 1, 0, 0}; }
inline CMap by_index(const u32 *idx) { return CMap{idx, 1, 0, 0}; }
inline CMap periodic(u64 div, u64 cnt, u64 off = 0) { return CMap{nullptr, div, cnt, off}; }

// out[c][p] = (p < size_a ? a[amap(c)][p] : 0) +- (p < size_b ? b[bmap(c)][p] : 0) for p < size_out: seal::Evaluator's add / sub
// on ciphertexts of unequal sizes (the destination grows, missing polynomials count as zero), copies and zero padding
// (b == nullptr), and broadcasts (periodic maps) in one pass.  out may alias a when the maps are the identity.
template <bool SUB>
__global__ __launch_bounds__(256) void k_add_general(const ulonglong2 *a, u32 size_a, CMap amap, const ulonglong2 *b, u32 size_b, CMap bmap,
                                                     ulonglong2 *out, u32 size_out, const Modulus *__restrict__ mods, u32 k, u32 half_n, u64 n_res_polys) {
    for (u64 rp = blockIdx.y; rp < n_res_polys; rp += gridDim.y) {
        const u32 prime = (u32)(rp % k);
        const u64 cp = rp / k;
        const u32 poly = (u32)(cp % size_out);
        const u64 ct = cp / size_out;
        const u64 q = mods[prime].q;
        const ulonglong2 *pa = (a && poly < size_a) ? a + ((amap(ct) * size_a + poly) * k + prime) * half_n : nullptr;
        const ulonglong2 *pb = (b && poly < size_b) ? b + ((bmap(ct) * size_b + poly) * k + prime) * half_n : nullptr;
        ulonglong2 *po = out + rp * half_n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < half_n; i += gridDim.x * blockDim.x) {
            ulonglong2 x = pa ? pa[i] : make_ulonglong2(0, 0);
            if (pb) {
                const ulonglong2 y = pb[i];
                x.x = SUB ? submod(x.x, y.x, q) : addmod(x.x, y.x, q);
                x.y = SUB ? submod(x.y, y.y, q) : addmod(x.y, y.y, q);
            }
            po[i] = x;
        }
    }
}

// the sum of homomorphic_sin / homomorphic_cos (homo/fhe_decode.h:113-118) in one pass: out[c][p] = zero[zmap(c)][p] (p < 2) +
// sum over the five power terms i with p < size_i of term_i[tmap(c)][p], p < size_out (11; 2 in the relinearised mode) -- modular additions of canonical residues, so
// their order does not show in the result (seven launches of k_add_general before)
struct TaylorTerms { const ulonglong2 *t[5]; u32 size[5]; };
__global__ __launch_bounds__(256) void k_taylor_sum(const ulonglong2 *__restrict__ zero, CMap zmap, TaylorTerms T, CMap tmap, ulonglong2 *__restrict__ out,
                                                    const Modulus *__restrict__ mods, u32 k, u32 half_n, u64 n_res_polys, u32 size_out) {
    for (u64 rp = blockIdx.y; rp < n_res_polys; rp += gridDim.y) {
        const u32 prime = (u32)(rp % k);
        const u64 cp = rp / k;
        const u32 poly = (u32)(cp % size_out);
        const u64 ct = cp / size_out;
        const u64 q = mods[prime].q;
        const ulonglong2 *pz = poly < 2 ? zero + ((zmap(ct) * 2 + poly) * k + prime) * half_n : nullptr;
        const u64 tc = tmap(ct);
        const ulonglong2 *pt[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) pt[i] = poly < T.size[i] ? T.t[i] + ((tc * T.size[i] + poly) * k + prime) * half_n : nullptr;
        ulonglong2 *po = out + rp * half_n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < half_n; i += gridDim.x * blockDim.x) {
            ulonglong2 x = pz ? pz[i] : make_ulonglong2(0, 0);
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (pt[t]) {
                    const ulonglong2 y = pt[t][i];
                    x.x = addmod(x.x, y.x, q);
                    x.y = addmod(x.y, y.y, q);
                }
            po[i] = x;
        }
    }
}

// x^e * P at coefficient j of a negacyclic polynomial: +-P[j - e]; returns the value to ADD
__device__ __forceinline__ u64 rot_term(const u64 *__restrict__ p, int j, int e, int n, u64 q) {
    const int idx = j - e;
    if (idx >= 0) return p[idx];
    const u64 v = p[idx + n];
    return v ? q - v : 0;
}
// Cubic's coefficient ciphertexts (homo/fhe_resize.h:150-172) for the base-2 encoder (encode(3) = x + 1, encode(2) = x,
// encode(5) = x^2 + 1, encode(4) = x^2):  a = 3B - A - 3C + D,  b = 2A - 5B + 4C - D,  c = C - A, the operands taken
// through index maps (the taps of SampleBicubic; no gathered copy of the source pixels is made)
__global__ __launch_bounds__(256) void k_cubic_coeffs_g(const u64 *__restrict__ A, CMap ma, const u64 *__restrict__ B, CMap mb,
                                                        const u64 *__restrict__ C, CMap mc, const u64 *__restrict__ D, CMap md,
                                                        u64 *__restrict__ a, u64 *__restrict__ b, u64 *__restrict__ c,
                                                        const Modulus *__restrict__ mods, u32 k, u32 n, u32 size) {
    const u64 rp = blockIdx.x;                       // (ct * size + poly) * k + prime
    const u32 prime = (u32)(rp % k);
    const u64 cp = rp / k;
    const u32 poly = (u32)(cp % size);
    const u64 ct = cp / size;
    const u64 q = mods[prime].q;
    const u64 in_off = ((u64)poly * k + prime) * n, ct_words = (u64)size * k * n;
    const u64 *pa = A + ma(ct) * ct_words + in_off, *pb = B + mb(ct) * ct_words + in_off;
    const u64 *pc = C + mc(ct) * ct_words + in_off, *pd = D + md(ct) * ct_words + in_off;
    for (int j = threadIdx.x; j < (int)n; j += 256) {
        const u64 Aj = pa[j], Bj = pb[j], Cj = pc[j], Dj = pd[j];
        u64 va = addmod(Bj, rot_term(pb, j, 1, n, q), q);                    // a = B (x+1) - A - C (x+1) + D
        va = submod(va, Aj, q);
        va = submod(va, addmod(Cj, rot_term(pc, j, 1, n, q), q), q);
        va = addmod(va, Dj, q);
        u64 vb = rot_term(pa, j, 1, n, q);                                   // b = A x - B (x^2+1) + C x^2 - D
        vb = submod(vb, addmod(Bj, rot_term(pb, j, 2, n, q), q), q);
        vb = addmod(vb, rot_term(pc, j, 2, n, q), q);
        vb = submod(vb, Dj, q);
        a[rp * n + j] = va;
        b[rp * n + j] = vb;
        c[rp * n + j] = submod(Cj, Aj, q);
    }
}
// Cubic's tail (homo/fhe_resize.h:181-188): out = (a + b + c) * encode(0.5) + B with encode(0.5) = -x^(n-1) = x^(-1):
// coefficient j takes S[j+1], the last one -S[0].  a, b have size_ab polynomials, c has size_c <= size_ab (c * t is one
// polynomial shorter than a * t^2), B has size_b <= size_ab and comes through an index map; the output ciphertext
// number goes through `mo` (row Cubics land in their cache slots).
__global__ __launch_bounds__(256) void k_cubic_combine_g(const u64 *__restrict__ a, const u64 *__restrict__ b, const u64 *__restrict__ c, u32 size_c,
                                                         const u64 *__restrict__ B, CMap mB, u32 size_b, u64 *__restrict__ out, CMap mo,
                                                         const Modulus *__restrict__ mods, u32 k, u32 n, u32 size_ab) {
    const u64 rp = blockIdx.x;                       // (ct * size_ab + poly) * k + prime
    const u32 prime = (u32)(rp % k);
    const u64 cp = rp / k;
    const u32 poly = (u32)(cp % size_ab);
    const u64 ct = cp / size_ab;
    const u64 q = mods[prime].q;
    const u64 *pa = a + rp * n, *pb = b + rp * n;
    const u64 *pc = poly < size_c ? c + ((ct * size_c + poly) * k + prime) * n : nullptr;
    const u64 *pB = poly < size_b ? B + ((mB(ct) * size_b + poly) * k + prime) * n : nullptr;
    u64 *po = out + ((mo(ct) * size_ab + poly) * k + prime) * n;
    for (int j = threadIdx.x; j < (int)n; j += 256) {
        const int src = j + 1 < (int)n ? j + 1 : 0;
        u64 s = addmod(pa[src], pb[src], q);
        if (pc) s = addmod(s, pc[src], q);
        if (j + 1 == (int)n) s = s ? q - s : 0;
        po[j] = pB ? addmod(s, pB[j], q) : s;
    }
}

// c_0 += / -= vals for `count` ciphertexts: add_plain / sub_plain with the scaled plaintext Delta * m' already on the device
__global__ void k_add_plain_dev(u64 *ct, u64 stride_words, u64 count, const u64 *__restrict__ vals, u32 len, const Modulus *__restrict__ mods,
                                u32 n, int sign) {
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


// approximated_step's offset chain (homo/fhe_decode.h:228-229) in one launch: output ciphertext b = offset + the plaintexts
// add_plain had added before step b (c_0 only; `chain` [count][k][len] holds those sums, Delta m' form, reduced mod q_i)
__global__ __launch_bounds__(256) void k_offset_chain(const u64 *__restrict__ offset, const u64 *__restrict__ chain, u32 len, u64 *__restrict__ out,
                                                      const Modulus *__restrict__ mods, u32 k, u32 n, u64 count) {
    const u32 pp = blockIdx.y;                         // poly * k + prime of a size-2 ciphertext
    const u32 prime = pp % k;
    const u64 q = mods[prime].q;
    for (u64 b = blockIdx.z; b < count; b += gridDim.z) {
        const u64 *tab = chain + (b * k + prime) * len;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            u64 v = offset[(u64)pp * n + i];
            if (pp < k && i < len) v = addmod(v, tab[i], q);
            out[(b * 2 * k + pp) * n + i] = v;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// constants of the circuits
// ------------------------------------------------------------------------------------------------
struct CircConst {
    std::vector<u64> plain;     // significant coefficients in [0, t)
    u32 nnz = 0;
    bool sparse = false;        // <= FHE_SPARSE_MAX_TERMS terms: multiply_plain as signed rotations (kernel arguments only)
    u64 *d_ntt = nullptr;       // fhe_plain_prepare form, built on first use
    u64 *d_scaled = nullptr;    // [k][len] Delta * m' for add_plain, built on first use when the plaintext has too many terms for the argument path
};

// out[omap(c)][rp][:] = src[c][rp][:], rp < rp_per_ct residue polynomials of half_n 16-byte pairs
__global__ __launch_bounds__(256) void k_scatter_ct(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ out, CMap omap, u32 rp_per_ct, u32 half_n, u64 nrp) {
    for (u64 rp = blockIdx.y; rp < nrp; rp += gridDim.y) {
        const u64 c = rp / rp_per_ct, r = rp % rp_per_ct;
        const ulonglong2 *s = src + rp * half_n;
        ulonglong2 *d = out + (omap(c) * rp_per_ct + r) * half_n;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < half_n; i += gridDim.x * blockDim.x) d[i] = s[i];
    }
}

struct fhe_circuits {
    const fhe_ctx *c = nullptr;
    int ic = 0, fc = 0;
    bool base2 = false;         // the encoder writes Cubic's constants as the fused passes assume
    const u64 *evk = nullptr;   // relinearised mode (fhe_circuits_create_relin): evaluation keys for s^2 (placement 1: s^2 then s^3), NTT form, caller-owned
    u32 dbc = 0;                // its decomposition bit count; 0 = the reference's mode (no relinearisation)
    u32 placement = FHE_RELIN_EVERY_PRODUCT;      // where the relinearised mode relinearises (include/fhe_circuits.h)
    mutable std::mutex mu;
    mutable std::map<u64, std::unique_ptr<CircConst>> consts;     // keyed by the bits of the double
    // pinned staging ring for index arrays: host memcpy + stream-ordered copy, the host never waits for the device
    // unless all slots are in flight
    static constexpr int kSlots = 8;
    static constexpr size_t kSlotBytes = 256 << 10;
    mutable unsigned char *pinned = nullptr;
    mutable hipEvent_t slot_ev[kSlots] = {};
    mutable bool slot_used[kSlots] = {};
    mutable int next_slot = 0;
};

namespace {

// the ABI speaks uint64_t (unsigned long), the kernels u64 (unsigned long long): same representation
inline const uint64_t *cu(const u64 *p) { return (const uint64_t *)p; }
inline uint64_t *mu(u64 *p) { return (uint64_t *)p; }

const CircConst *get_const(const fhe_circuits *cc, double v) {
    u64 bits;
    memcpy(&bits, &v, 8);
    if (v == 0.0) bits = 0;                                      // -0.0 encodes like 0.0
    std::lock_guard<std::mutex> lk(cc->mu);
    auto it = cc->consts.find(bits);
    if (it != cc->consts.end()) return it->second.get();
    std::unique_ptr<CircConst> k(new CircConst);
    std::vector<u64> buf(cc->c->n);
    const int len = fhe_frac_encode(cc->c->n, cc->c->t, v, cc->ic, cc->fc, mu(buf.data()));
    if (len < 0) return nullptr;
    k->plain.assign(buf.begin(), buf.begin() + len);
    for (u64 m : k->plain) k->nnz += m != 0;
    k->sparse = k->nnz > 0 && k->nnz <= FHE_SPARSE_MAX_TERMS && cc->c->logn <= 13;
    const CircConst *out = k.get();
    cc->consts.emplace(bits, std::move(k));
    return out;
}
int ensure_ntt(const fhe_circuits *cc, const CircConst *kc, hipStream_t st) {
    std::lock_guard<std::mutex> lk(cc->mu);
    if (kc->d_ntt) return FHE_OK;
    u64 *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, fhe_plain_ntt_words(cc->c) * sizeof(u64)));
    const int rc = fhe_plain_prepare(cc->c, cu(kc->plain.data()), (u32)kc->plain.size(), mu(d), st);      // synchronous; first use only
    if (rc) { (void)hipFree(d); return rc; }
    const_cast<CircConst *>(kc)->d_ntt = d;
    return FHE_OK;
}
int ensure_scaled(const fhe_circuits *cc, const CircConst *kc) {
    using namespace hostmath;
    std::lock_guard<std::mutex> lk(cc->mu);
    if (kc->d_scaled) return FHE_OK;
    const fhe_ctx *c = cc->c;
    const size_t len = kc->plain.size();
    std::vector<u64> vals((size_t)c->k * len);
    for (u32 i = 0; i < c->k; ++i) {
        const u64 qi = c->qb.primes[i];
        for (size_t j = 0; j < len; ++j) {
            const u64 m = kc->plain[j];
            u64 v = mulmod(c->delta_mod[i], m % qi, qi);
            if (m >= c->upper_half_threshold) v = addmod(v, c->upper_half_increment[i], qi);
            vals[i * len + j] = v;
        }
    }
    u64 *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, vals.size() * sizeof(u64)));
    const hipError_t e = hipMemcpy(d, vals.data(), vals.size() * sizeof(u64), hipMemcpyHostToDevice);       // blocking; first use only
    if (e != hipSuccess) { (void)hipFree(d); return fail(FHE_ERR_HIP, "constant upload: %s", hipGetErrorString(e)); }
    const_cast<CircConst *>(kc)->d_scaled = d;
    return FHE_OK;
}

// host index array -> device, stream-ordered, through the pinned ring
int stage_u32(const fhe_circuits *cc, const u32 *host, size_t count, u32 *dev, hipStream_t st) {
    std::lock_guard<std::mutex> lk(cc->mu);
    size_t done = 0;
    const size_t per = fhe_circuits::kSlotBytes / sizeof(u32);
    while (done < count) {
        const size_t part = count - done < per ? count - done : per;
        const int s = cc->next_slot;
        cc->next_slot = (s + 1) % fhe_circuits::kSlots;
        if (cc->slot_used[s]) HIP_TRY(hipEventSynchronize(cc->slot_ev[s]));
        unsigned char *p = cc->pinned + (size_t)s * fhe_circuits::kSlotBytes;
        memcpy(p, host + done, part * sizeof(u32));
        HIP_TRY(hipMemcpyAsync(dev + done, p, part * sizeof(u32), hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(cc->slot_ev[s], st));
        cc->slot_used[s] = true;
        done += part;
    }
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// Run: one evaluation (or one dry run) of a circuit
// ------------------------------------------------------------------------------------------------
#define TRY(expr)                    \
    do {                             \
        const int rc_ = (expr);      \
        if (rc_) return rc_;         \
    } while (0)

struct Run {
    const fhe_circuits *cc;
    const fhe_ctx *c;
    hipStream_t st;
    bool dry;
    bool query = false;                    // a *_scratch_bytes query: placeholder scalars, so constants are neither required to fit the encoder nor to be non-zero
    bool relin;                            // the handle relinearises after every multiply / square: every ciphertext has two polynomials
    bool tail;                             // the handle relinearises ONCE at the end of every Cubic / Linear (size 4 / 3 -> 2): inside them the reference's sizes
    bool stail;                            // ... ONCE per output of a sampler / of a stand-alone Cubic or Linear (FHE_RELIN_PER_SAMPLE): the reference's sizes until then
    uintptr_t base = 0;
    size_t cap = 0, top = 0, high = 0;     // bytes
    u32 k, n;
    size_t pw;                             // words of one RNS polynomial

    Run(const fhe_circuits *circ, void *scratch, size_t bytes, fhe_stream s, bool dry_run)
        : cc(circ), c(circ->c), st((hipStream_t)s), dry(dry_run), relin(circ->dbc != 0 && circ->placement == FHE_RELIN_EVERY_PRODUCT),
          tail(circ->dbc != 0 && circ->placement == FHE_RELIN_PER_CUBIC), stail(circ->dbc != 0 && circ->placement == FHE_RELIN_PER_SAMPLE), base((uintptr_t)scratch), cap(bytes), k(circ->c->k), n(circ->c->n),
          pw((size_t)circ->c->k * circ->c->n) {}

    // polynomials of a ciphertext that has `ref` of them in the reference's evaluation
    u32 S(u32 ref) const { return relin && ref > 2 ? 2 : ref; }
    // polynomials of a Cubic's / Linear's / sampler's RESULT (either relinearised mode: 2)
    u32 O(u32 ref) const { return (relin || tail) && ref > 2 ? 2 : ref; }
    // ... of what a circuit hands to its caller (every relinearised placement: 2)
    u32 F(u32 ref) const { return (relin || tail || stail) && ref > 2 ? 2 : ref; }
    // FHE_RELIN_PER_SAMPLE: `produce(dst)` forms a batch of `ref`-polynomial results; they are relinearised into `out` (ref -> 2: all
    // the key switches of one evaluator.relinearize, fhe_relinearize_n).  Other placements: produce(out).
    template <typename P>
    int finish_sample(u32 ref, u64 *out, u64 count, P &&produce) {
        if (!stail || ref <= 2) return produce(out);
        if (ref - 2 > 4) return fail(FHE_ERR_PARAM, "per-sample relinearisation: a result of %u polynomials needs keys beyond s^5", ref);
        const size_t m = mark();
        u64 *raw = alloc(count * ref * pw);
        int rc = produce(raw);
        if (!rc) rc = relin_n(raw, ref, out, count);
        release(m);
        return rc;
    }

    // 256-byte aligned bump allocation; in a dry run only the high-water mark is real
    u64 *alloc(size_t words) {
        const size_t bytes = (words * sizeof(u64) + 255) & ~(size_t)255;
        const size_t at = top;
        top += bytes;
        if (top > high) high = top;
        return (u64 *)(base + at);
    }
    u32 *alloc_u32(size_t count) { return (u32 *)alloc((count + 1) / 2); }
    size_t mark() const { return top; }
    void release(size_t m) { top = m; }

    const CircConst *K(double v) {
        const CircConst *kc = get_const(cc, v);
        if (!kc && query) {                // the arena's high-water mark does not depend on the constants' values
            static const CircConst placeholder = [] { CircConst p; p.nnz = 1; return p; }();
            return &placeholder;
        }
        if (!kc) fail(FHE_ERR_PARAM, "constant %.17g does not fit the encoder", v);
        return kc;
    }

    int stage(const u32 *host, size_t count, u32 *dev) { return dry ? FHE_OK : stage_u32(cc, host, count, dev, st); }
    int add(const u64 *a, const u64 *b, u64 *out, u64 polys) { return dry ? FHE_OK : fhe_add(c, cu(a), cu(b), mu(out), polys, st); }
    int sub(const u64 *a, const u64 *b, u64 *out, u64 polys) { return dry ? FHE_OK : fhe_sub(c, cu(a), cu(b), mu(out), polys, st); }
    int neg(const u64 *a, u64 *out, u64 polys) { return dry ? FHE_OK : fhe_negate(c, cu(a), mu(out), polys, st); }
    int copy(u64 *dst, const u64 *src, u64 polys) {
        if (dry || !polys) return FHE_OK;
        HIP_TRY(hipMemcpyAsync(dst, src, polys * pw * sizeof(u64), hipMemcpyDeviceToDevice, st));
        return FHE_OK;
    }
    // out[c] (size_out) = a[amap(c)] (size_a, zero-padded) +- b[bmap(c)] (size_b, zero-padded); b may be null
    int add_general(bool subtract, const u64 *a, u32 size_a, CMap amap, const u64 *b, u32 size_b, CMap bmap, u64 *out, u32 size_out, u64 count) {
        if (dry || !count) return FHE_OK;
        const u64 nrp = count * size_out * k;
        const u32 half_n = n / 2;
        dim3 grid((half_n + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
        if (subtract) k_add_general<true><<<grid, 256, 0, st>>>((const ulonglong2 *)a, size_a, amap, (const ulonglong2 *)b, size_b, bmap, (ulonglong2 *)out, size_out, c->qb.d_mod, k, half_n, nrp);
        else k_add_general<false><<<grid, 256, 0, st>>>((const ulonglong2 *)a, size_a, amap, (const ulonglong2 *)b, size_b, bmap, (ulonglong2 *)out, size_out, c->qb.d_mod, k, half_n, nrp);
        KERNEL_CHECK();
        return FHE_OK;
    }
    // dst[c] (size_dst) += src[map(c)] (size_src <= size_dst), in place
    int acc(u64 *dst, u32 size_dst, const u64 *src, u32 size_src, CMap map, u64 count) {
        return add_general(false, dst, size_dst, ident(), src, size_src, map, dst, size_dst, count);
    }
    int gather_pad(const u64 *src, u32 size_src, CMap map, u64 *dst, u32 size_dst, u64 count) {
        return add_general(false, src, size_src, map, nullptr, 0, ident(), dst, size_dst, count);
    }
    int mul_plain(const u64 *in, u64 *out, u64 polys, const CircConst *kc) {
        if (!kc) return FHE_ERR_PARAM;
        if (!kc->nnz && !query) return fail(FHE_ERR_PARAM, "plain cannot be zero");
        if (dry || !polys) return FHE_OK;
        if (kc->sparse) return fhe_multiply_plain_sparse(c, cu(in), mu(out), polys, cu(kc->plain.data()), (u32)kc->plain.size(), st);
        TRY(ensure_ntt(cc, kc, st));
        return fhe_multiply_plain(c, cu(in), mu(out), polys, cu(kc->d_ntt), st);
    }
    // out[c][p] = addend[amap(c)][p] (p < addend_size) + sum_i src_i[c][p] * kc_i (p < size_i): fhe_multiply_plain_sum; the src_i are overwritten
    int mul_plain_sum(int terms, u64 *const *src, const u32 *sizes, const CircConst *const *kc, const u64 *addend, CMap amap, u32 addend_size,
                      u64 *out, u32 out_size, u64 count) {
        for (int i = 0; i < terms; ++i) {
            if (!kc[i]) return FHE_ERR_PARAM;
            if (!kc[i]->nnz && !query) return fail(FHE_ERR_PARAM, "plain cannot be zero");
        }
        if (dry || !count) return FHE_OK;
        PlainSumTerms T;
        T.count = (u32)terms;
        for (int i = 0; i < terms; ++i) {
            TRY(ensure_ntt(cc, kc[i], st));
            T.src[i] = src[i]; T.size[i] = sizes[i]; T.plain[i] = (const ulonglong2 *)kc[i]->d_ntt;
        }
        return fhe_multiply_plain_sum(c, T, addend, amap, addend_size, out, out_size, count, st);
    }
    int add_plain(u64 *ct, u32 size, u64 count, const CircConst *kc, int sign = 1) {
        if (!kc) return FHE_ERR_PARAM;
        if (dry || !count || !kc->nnz) return FHE_OK;
        if (kc->nnz <= 24) return fhe_add_plain(c, mu(ct), (u64)size * pw, count, cu(kc->plain.data()), (u32)kc->plain.size(), sign, st);      // kernel-argument path
        TRY(ensure_scaled(cc, kc));
        const u32 len = (u32)kc->plain.size();
        dim3 grid((len + 255) / 256, k, (unsigned)(count < 16384 ? count : 16384));
        k_add_plain_dev<<<grid, 256, 0, st>>>(ct, (u64)size * pw, count, kc->d_scaled, len, c->qb.d_mod, n, sign);
        KERNEL_CHECK();
        return FHE_OK;
    }
    // ct x ct products; the BEHZ scratch comes from the arena and is released again
    u64 *prepare_alloc(u32 size, u64 count) { return alloc(fhe_multiply_operand_words(c, size, count)); }
    int prepare(const u64 *a, u32 size, u64 count, u64 *prepared) { return dry ? FHE_OK : fhe_multiply_prepare(c, cu(a), size, count, mu(prepared), st); }
    // evaluator.relinearize after a product (relinearised mode only): prod [count][3] -> out [count][2]
    int relin_to(const u64 *prod, u64 *out, u64 count) {
        const size_t bytes = fhe_relinearize_scratch_bytes(c, cc->dbc, count);
        const size_t m = mark();
        void *scr = alloc((bytes + 7) / 8);
        int rc = FHE_OK;
        if (!dry && count) rc = fhe_relinearize_to(c, cu(prod), 3 * pw, mu(out), 2 * pw, count, cu(cc->evk), cc->dbc, scr, bytes, st);
        release(m);
        return rc;
    }
    // evaluator.relinearize(result, evk) at the end of a Cubic / Linear (placement per Cubic): raw [count][size] -> out [count][2], size - 2
    // key switches (keys for s^(size-1) .. s^2); raw is scratch afterwards
    int relin_n(u64 *raw, u32 size, u64 *out, u64 count) {
        const size_t bytes = fhe_relinearize_n_scratch_bytes(c, size, cc->dbc, count);
        const size_t m = mark();
        void *scr = alloc((bytes + 7) / 8);
        int rc = FHE_OK;
        if (!dry && count) rc = fhe_relinearize_n(c, mu(raw), size, (u64)size * pw, mu(out), 2 * pw, count, cu(cc->evk), cc->dbc, scr, bytes, st);
        release(m);
        return rc;
    }
    // out[omap(c)] = src[c] (ciphertexts of `size` polynomials)
    int scatter(const u64 *src, u32 size, u64 *out, CMap omap, u64 count) {
        if (dry || !count) return FHE_OK;
        const u64 nrp = count * size * k;
        const u32 half_n = n / 2;
        dim3 grid((half_n + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
        k_scatter_ct<<<grid, 256, 0, st>>>((const ulonglong2 *)src, (ulonglong2 *)out, omap, size * k, half_n, nrp);
        KERNEL_CHECK();
        return FHE_OK;
    }
    // a product of the circuit: `raw` forms it; in the relinearised mode it lands in a temporary and is relinearised into `out`
    template <typename F>
    int product(u32 sa, u32 sb, u64 *out, u64 count, F &&raw) {
        if (!relin) return raw(out);
        if (sa != 2 || sb != 2) return fail(FHE_ERR_PARAM, "relinearised mode: products are 2 x 2 (got %u x %u)", sa, sb);
        const size_t m = mark();
        u64 *tmp = alloc(count * 3 * pw);
        int rc = raw(tmp);
        if (!rc) rc = relin_to(tmp, out, count);
        release(m);
        return rc;
    }
    int multiply(const u64 *a, u32 sa, const u64 *b, const u64 *bp, u32 sb, CMap bmap, u64 *out, u64 count) {
        return product(sa, sb, out, count, [&](u64 *o) { return multiply_raw(a, sa, b, bp, sb, bmap, o, count); });
    }
    int multiply_pp(const u64 *ap, u32 sa, const u64 *bp, u32 sb, u64 *out, u64 count) {
        return product(sa, sb, out, count, [&](u64 *o) { return multiply_pp_raw(ap, sa, bp, sb, o, count); });
    }
    int square(const u64 *a, u32 sa, u64 *out, u64 count) {
        return product(sa, sa, out, count, [&](u64 *o) { return square_raw(a, sa, o, count); });
    }
    // a (plain ciphertexts) x b; exactly one of b / bp (prepared) is given; bmap.cnt != 0: bp is a shared batch
    int multiply_raw(const u64 *a, u32 sa, const u64 *b, const u64 *bp, u32 sb, CMap bmap, u64 *out, u64 count) {
        const size_t bytes = fhe_multiply_scratch_bytes(c, sa, sb, count);
        const size_t m = mark();
        void *scr = alloc((bytes + 7) / 8);
        int rc = FHE_OK;
        if (!dry && count) {
            if (bp && bmap.cnt) rc = fhe_multiply_prepared_shared(c, cu(a), nullptr, sa, cu(bp), sb, bmap.cnt, bmap.div, bmap.off, mu(out), count, scr, bytes, st);
            else if (bp) rc = fhe_multiply_prepared(c, cu(a), nullptr, sa, nullptr, cu(bp), sb, mu(out), count, scr, bytes, st);
            else rc = fhe_multiply(c, cu(a), sa, cu(b), sb, mu(out), count, scr, bytes, st);
        }
        release(m);
        return rc;
    }
    // both operands in prepared form (an operand that enters several products is prepared once)
    int multiply_pp_raw(const u64 *ap, u32 sa, const u64 *bp, u32 sb, u64 *out, u64 count) {
        const size_t bytes = fhe_multiply_scratch_bytes(c, sa, sb, count);
        const size_t m = mark();
        void *scr = alloc((bytes + 7) / 8);
        int rc = FHE_OK;
        if (!dry && count) rc = fhe_multiply_prepared(c, nullptr, cu(ap), sa, nullptr, cu(bp), sb, mu(out), count, scr, bytes, st);
        release(m);
        return rc;
    }
    int square_raw(const u64 *a, u32 sa, u64 *out, u64 count) {
        const size_t bytes = fhe_multiply_scratch_bytes(c, sa, sa, count);
        const size_t m = mark();
        void *scr = alloc((bytes + 7) / 8);
        int rc = FHE_OK;
        if (!dry && count) rc = fhe_square(c, cu(a), sa, mu(out), count, scr, bytes, st);
        release(m);
        return rc;
    }
};

// an operand of Cubic / Linear: `count` ciphertexts taken from `base` through `map`
struct Src {
    const u64 *base;
    CMap map;
};

// ------------------------------------------------------------------------------------------------
// Cubic (homo/fhe_resize.h:143-189)
// ------------------------------------------------------------------------------------------------
// p2 / p1: prepared t^2 (size 3; 2 in the relinearised mode) and t (size 2), indexed by the pair number through `tmap` (identity
// map: `count` entries).  Relinearised mode: size == 2, and the three products come back as size-2 ciphertexts (each relinearised
// on its own, as evaluator.relinearize after :176-178 would), so the output has 2 polynomials instead of size + 2.
int cubic_core_ref(Run &R, Src A, Src B, Src C, Src D, u32 size, const u64 *p2, const u64 *p1, CMap tmap, u64 *out, CMap omap, u64 count);
// Placement per Cubic: the reference's sequence unchanged on size-2 operands (result: 4 polynomials, the fused tail included), then
// ONE evaluator.relinearize(result, evk) -- two key switches, keys for s^3 and s^2 -- into the compact size-2 output.
int cubic_core(Run &R, Src A, Src B, Src C, Src D, u32 size, const u64 *p2, const u64 *p1, CMap tmap, u64 *out, CMap omap, u64 count) {
    if (!R.tail) return cubic_core_ref(R, A, B, C, D, size, p2, p1, tmap, out, omap, count);
    if (!count) return FHE_OK;
    if (size != 2) return fail(FHE_ERR_PARAM, "relinearised mode: Cubic takes size-2 ciphertexts");
    const size_t m = R.mark();
    u64 *raw = R.alloc(count * 4 * R.pw);
    TRY(cubic_core_ref(R, A, B, C, D, 2, p2, p1, tmap, raw, ident(), count));
    const bool direct = !omap.idx && !omap.cnt;
    u64 *dst = direct ? out : R.alloc(count * 2 * R.pw);
    TRY(R.relin_n(raw, 4, dst, count));
    if (!direct) TRY(R.scatter(dst, 2, out, omap, count));
    R.release(m);
    return FHE_OK;
}
int cubic_core_ref(Run &R, Src A, Src B, Src C, Src D, u32 size, const u64 *p2, const u64 *p1, CMap tmap, u64 *out, CMap omap, u64 count) {
    if (!count) return FHE_OK;
    const fhe_ctx *c = R.c;
    const size_t m = R.mark();
    const u64 polys = count * size;
    if (polys * R.k > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    u64 *a = R.alloc(polys * R.pw), *b = R.alloc(polys * R.pw), *cc = R.alloc(polys * R.pw);
    if (R.cc->base2) {
        if (!R.dry) {
            k_cubic_coeffs_g<<<(unsigned)(polys * R.k), 256, 0, R.st>>>(A.base, A.map, B.base, B.map, C.base, C.map, D.base, D.map, a, b, cc, c->qb.d_mod, R.k, R.n, size);
            KERNEL_CHECK();
        }
    } else {
        // the Evaluator call sequence of :150-172 (an encoder that does not write 3, 2, 5, 4 as x+1, x, x^2+1, x^2)
        u64 *t0 = R.alloc(polys * R.pw), *t1 = R.alloc(polys * R.pw);
        TRY(R.gather_pad(B.base, size, B.map, t0, size, count));
        TRY(R.mul_plain(t0, a, polys, R.K(3)));                                              // boaz2 = 3B
        TRY(R.add_general(true, a, size, ident(), A.base, size, A.map, a, size, count));     // - A
        TRY(R.gather_pad(C.base, size, C.map, t0, size, count));
        TRY(R.mul_plain(t0, t1, polys, R.K(3)));
        TRY(R.sub(a, t1, a, polys));                                                         // - 3C
        TRY(R.add_general(false, a, size, ident(), D.base, size, D.map, a, size, count));    // + D
        TRY(R.gather_pad(A.base, size, A.map, t0, size, count));
        TRY(R.mul_plain(t0, b, polys, R.K(2)));                                              // boaz5 = 2A
        TRY(R.gather_pad(B.base, size, B.map, t0, size, count));
        TRY(R.mul_plain(t0, t1, polys, R.K(5)));
        TRY(R.sub(b, t1, b, polys));                                                         // - 5B
        TRY(R.gather_pad(C.base, size, C.map, t0, size, count));
        TRY(R.mul_plain(t0, t1, polys, R.K(4)));
        TRY(R.add(b, t1, b, polys));                                                         // + 4C
        TRY(R.add_general(true, b, size, ident(), D.base, size, D.map, b, size, count));     // - D
        TRY(R.gather_pad(C.base, size, C.map, cc, size, count));
        TRY(R.add_general(true, cc, size, ident(), A.base, size, A.map, cc, size, count));   // c = C - A
    }
    if (R.relin && size != 2) return fail(FHE_ERR_PARAM, "relinearised mode: Cubic takes size-2 ciphertexts");
    const u32 so = R.S(size + 2), sc = R.S(size + 1), s2 = R.S(3);      // a t^2 and b t^2, c t, t^2
    if (count * so * R.k > 0x7fffffffULL) return fail(FHE_ERR_PARAM, "too many polynomials for one launch");
    if (!R.relin && R.cc->base2 && !tmap.idx && !c->opt.cubic_unfused && fhe_behz_floor3_supported(c)) {      // the same decision in the dry pass and the real one
        const u64 t_cnt = tmap.cnt ? tmap.cnt : count, t_div = tmap.cnt ? tmap.div : 1, t_off = tmap.cnt ? tmap.off : 0;     // identity = period `count`
        // the three products up to their inverse transforms, then ONE launch that floors, converts back, adds the three, applies
        // encode(0.5) = x^-1 and adds B (behz.hip: k_behz_floor3_combine_pm): pa, pb, pc never exist in memory
        u64 *da = R.alloc(fhe_behz_d_words(c, so, count)), *db = R.alloc(fhe_behz_d_words(c, so, count)), *dc = R.alloc(fhe_behz_d_words(c, so - 1, count));
        const size_t m2 = R.mark();
        u64 *prep = R.alloc(fhe_multiply_operand_words(c, size, count));
        if (!R.dry) {
            TRY(fhe_behz_tensor_shared(c, a, size, p2, 3, t_cnt, t_div, t_off, da, count, prep, R.st));     // a * t3 (t3 = t * t, :175)
            TRY(fhe_behz_tensor_shared(c, b, size, p2, 3, t_cnt, t_div, t_off, db, count, prep, R.st));     // b * t2
            TRY(fhe_behz_tensor_shared(c, cc, size, p1, 2, t_cnt, t_div, t_off, dc, count, prep, R.st));    // c * t
            TRY(fhe_behz_floor3_combine(c, da, db, dc, so, so - 1, B.base, B.map, size, out, omap, count, R.st));    // :181-188
        }
        R.release(m2);
        R.release(m);
        return FHE_OK;
    }
    u64 *pa = R.alloc(count * so * R.pw), *pb = R.alloc(count * so * R.pw), *pc = R.alloc(count * sc * R.pw);
    TRY(R.multiply(a, size, nullptr, p2, s2, tmap, pa, count));         // a * t3 (t3 = t * t, :175)
    TRY(R.multiply(b, size, nullptr, p2, s2, tmap, pb, count));         // b * t2
    TRY(R.multiply(cc, size, nullptr, p1, 2, tmap, pc, count));         // c * t
    if (R.cc->base2) {
        if (!R.dry) {
            k_cubic_combine_g<<<(unsigned)(count * so * R.k), 256, 0, R.st>>>(pa, pb, pc, sc, B.base, B.map, size, out, omap, c->qb.d_mod, R.k, R.n, so);
            KERNEL_CHECK();
        }
    } else {
        TRY(R.add(pa, pb, pa, count * so));                                                  // :181-184
        TRY(R.acc(pa, so, pc, sc, ident(), count));
        TRY(R.mul_plain(pa, pa, count * so, R.K(0.5)));
        if (omap.idx || omap.cnt) return fail(FHE_ERR_PARAM, "output maps need the base-2 encoder");
        TRY(R.add_general(false, pa, so, ident(), B.base, size, B.map, out, so, count));                     // :187
    }
    R.release(m);
    return FHE_OK;
}

// t [count][2] -> prepared t^2 and t in the arena (not released: the caller's mark does that)
int cubic_powers(Run &R, const u64 *t, u64 count, u64 **p2, u64 **p1) {
    const u32 s2 = R.S(3);
    *p2 = R.prepare_alloc(s2, count);
    *p1 = R.prepare_alloc(2, count);
    const size_t m = R.mark();
    u64 *t2 = R.alloc(count * s2 * R.pw);
    TRY(R.square(t, 2, t2, count));                                     // t2 = square(t) = t3 (:174-175); relinearised in that mode
    TRY(R.prepare(t2, s2, count, *p2));
    TRY(R.prepare(t, 2, count, *p1));
    R.release(m);
    return FHE_OK;
}

int run_cubic(Run &R, const u64 *A, const u64 *B, const u64 *C, const u64 *D, u32 size, const u64 *t, u64 *out, u64 count) {
    u64 *p2, *p1;
    TRY(cubic_powers(R, t, count, &p2, &p1));
    return R.finish_sample(size + 2, out, count, [&](u64 *dst) {
        return cubic_core(R, Src{A, ident()}, Src{B, ident()}, Src{C, ident()}, Src{D, ident()}, size, p2, p1, ident(), dst, ident(), count);
    });
}

// ------------------------------------------------------------------------------------------------
// Linear (homo/fhe_resize.h:191-204): (1 - t) A + t B
// ------------------------------------------------------------------------------------------------
// pomt / pt: prepared (1 - t) and t, indexed through tmap; A, B contiguous [count][size]
int linear_core(Run &R, const u64 *A, const u64 *B, u32 size, const u64 *pomt, const u64 *pt, CMap tmap, u64 *out, u64 count) {
    if ((R.relin || R.tail) && size != 2) return fail(FHE_ERR_PARAM, "relinearised mode: Linear takes size-2 ciphertexts");
    if (R.tail) {                                                       // :196-199 unchanged (3 polynomials), then one evaluator.relinearize
        const size_t m = R.mark();
        u64 *x = R.alloc(count * 3 * R.pw), *y = R.alloc(count * 3 * R.pw);
        TRY(R.multiply(A, 2, nullptr, pomt, 2, tmap, x, count));
        TRY(R.multiply(B, 2, nullptr, pt, 2, tmap, y, count));
        TRY(R.add(x, y, x, count * 3));
        TRY(R.relin_n(x, 3, out, count));
        R.release(m);
        return FHE_OK;
    }
    const size_t m = R.mark();
    const u32 so = R.S(size + 1);
    u64 *tmp = R.alloc(count * so * R.pw);
    TRY(R.multiply(A, size, nullptr, pomt, 2, tmap, out, count));       // boaz1 = (1 - t) * A
    TRY(R.multiply(B, size, nullptr, pt, 2, tmap, tmp, count));         // boaz2 = B * t
    TRY(R.add(out, tmp, out, count * so));
    R.release(m);
    return FHE_OK;
}
int linear_operands(Run &R, const u64 *t, u64 count, u64 **pomt, u64 **pt) {
    *pomt = R.prepare_alloc(2, count);
    *pt = R.prepare_alloc(2, count);
    const size_t m = R.mark();
    u64 *omt = R.alloc(count * 2 * R.pw);
    TRY(R.neg(t, omt, count * 2));                                      // :196
    TRY(R.add_plain(omt, 2, count, R.K(1.0)));
    TRY(R.prepare(omt, 2, count, *pomt));
    TRY(R.prepare(t, 2, count, *pt));
    R.release(m);
    return FHE_OK;
}
int run_linear(Run &R, const u64 *A, const u64 *B, u32 size, const u64 *t, u64 *out, u64 count) {
    u64 *pomt, *pt;
    TRY(linear_operands(R, t, count, &pomt, &pt));
    return R.finish_sample(size + 1, out, count, [&](u64 *dst) { return linear_core(R, A, B, size, pomt, pt, ident(), dst, count); });
}

// ------------------------------------------------------------------------------------------------
// SampleBicubic / SampleLinear (homo/fhe_resize.h:222-305) over a batch of output pixels
// ------------------------------------------------------------------------------------------------
int run_sample_bicubic(Run &R, const u64 *pixels, u64 n_pixels, const u32 *taps, const u64 *xfract, const u64 *yfract, u64 *out, u64 count) {
    // the four row Cubics of all pixels as ONE batch of 4 * count (row-major: pair r * count + c), every tap array
    // laid out to match: T[i][r * count + c] = taps[c][4 r + i]
    const u64 rows = 4 * count;
    u32 *d_idx = R.alloc_u32(4 * rows);
    if (taps && R.dry)                                                  // the validating pass of real_run: nothing is enqueued on a bad tap
        for (u64 c = 0; c < count * 16; ++c)
            if (taps[c] >= n_pixels) return fail(FHE_ERR_PARAM, "tap %u of pixel %llu is outside the %llu source pixels", (unsigned)(c % 16), (unsigned long long)(c / 16), (unsigned long long)n_pixels);
    if (!R.dry) {
        std::vector<u32> h(4 * rows);
        for (u64 c = 0; c < count; ++c)
            for (u32 r = 0; r < 4; ++r)
                for (u32 i = 0; i < 4; ++i) h[i * rows + r * count + c] = taps[c * 16 + 4 * r + i];
        TRY(R.stage(h.data(), h.size(), d_idx));
    }
    u64 *px2, *px1;
    TRY(cubic_powers(R, xfract, count, &px2, &px1));
    const u32 sr = R.O(4);                                              // size of a row Cubic's result
    u64 *cols = R.alloc(rows * sr * R.pw);                              // [4][count][sr][k][n]
    const CMap xm = periodic(1, count);                                 // pair r * count + c multiplies xfract[c]
    TRY(cubic_core(R, Src{pixels, by_index(d_idx)}, Src{pixels, by_index(d_idx + rows)}, Src{pixels, by_index(d_idx + 2 * rows)},
                   Src{pixels, by_index(d_idx + 3 * rows)}, 2, px2, px1, xm, cols, ident(), rows));
    u64 *py2, *py1;
    TRY(cubic_powers(R, yfract, count, &py2, &py1));
    const size_t cw = count * sr * R.pw;
    return R.finish_sample(R.O(6), out, count, [&](u64 *dst) {                // :303; FHE_RELIN_PER_SAMPLE: + one relinearize of the pixel, 6 -> 2
        return cubic_core(R, Src{cols, ident()}, Src{cols + cw, ident()}, Src{cols + 2 * cw, ident()}, Src{cols + 3 * cw, ident()}, sr, py2, py1,
                          ident(), dst, ident(), count);
    });
}

int run_sample_linear(Run &R, const u64 *pixels, u64 n_pixels, const u32 *taps, const u64 *xfract, const u64 *yfract, u64 *out, u64 count) {
    // the two row Linears as one batch of 2 * count: A = (p00 | p01), B = (p10 | p11)
    const u64 rows = 2 * count;
    u32 *d_idx = R.alloc_u32(2 * rows);
    if (taps && R.dry)
        for (u64 c = 0; c < count * 4; ++c)
            if (taps[c] >= n_pixels) return fail(FHE_ERR_PARAM, "tap %u of pixel %llu is outside the %llu source pixels", (unsigned)(c % 4), (unsigned long long)(c / 4), (unsigned long long)n_pixels);
    if (!R.dry) {
        std::vector<u32> h(2 * rows);
        for (u64 c = 0; c < count; ++c)
            for (u32 i = 0; i < 4; ++i) h[(i & 1) * rows + (i >> 1) * count + c] = taps[c * 4 + i];      // i = 0: p00, 1: p10, 2: p01, 3: p11
        TRY(R.stage(h.data(), h.size(), d_idx));
    }
    u64 *A = R.alloc(rows * 2 * R.pw), *B = R.alloc(rows * 2 * R.pw);
    TRY(R.gather_pad(pixels, 2, by_index(d_idx), A, 2, rows));
    TRY(R.gather_pad(pixels, 2, by_index(d_idx + rows), B, 2, rows));
    u64 *pomx, *ptx;
    TRY(linear_operands(R, xfract, count, &pomx, &ptx));
    const u32 sr = R.O(3);
    u64 *cols = R.alloc(rows * sr * R.pw);                              // [2][count][sr][k][n]
    TRY(linear_core(R, A, B, 2, pomx, ptx, periodic(1, count), cols, rows));
    u64 *pomy, *pty;
    TRY(linear_operands(R, yfract, count, &pomy, &pty));
    return R.finish_sample(R.O(4), out, count, [&](u64 *dst) { return linear_core(R, cols, cols + count * sr * R.pw, sr, pomy, pty, ident(), dst, count); });
}

// ------------------------------------------------------------------------------------------------
// ResizeImage with shared offsets (one ciphertext per output column / row)
// ------------------------------------------------------------------------------------------------
struct ResizeIndex {
    std::vector<u32> colx;        // [dst_w][4] clamped xi-1 .. xi+2
    std::vector<u32> rows_of;     // [dst_h][4] clamped yi-1 .. yi+2
};
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
void resize_index(u32 src_w, u32 src_h, u32 dst_w, u32 dst_h, ResizeIndex &ix) {
    ix.colx.resize((size_t)dst_w * 4);
    ix.rows_of.resize((size_t)dst_h * 4);
    for (u32 x = 0; x < dst_w; ++x) {
        const float u = (float)((float)x / (float)(dst_w - 1) * (float)src_w - 0.5);      // homo/fhe_resize.h:382
        const int xi = (int)u;
        for (int i = 0; i < 4; ++i) ix.colx[x * 4 + i] = (u32)clampi(xi - 1 + i, 0, (int)src_w - 1);
    }
    for (u32 y = 0; y < dst_h; ++y) {
        const float v = (float)((float)y / (float)(dst_h - 1) * (float)src_h - 0.5);      // :351
        const int yi = (int)v;
        for (int j = 0; j < 4; ++j) ix.rows_of[y * 4 + j] = (u32)clampi(yi - 1 + j, 0, (int)src_h - 1);
    }
}

// A shard of the destination rows (the whole image: row0 = 0, row1 = dst_h, src_row0 = 0, n_src_rows = src_h).  `pixels`
// holds the source rows [src_row0, src_row0 + n_src_rows) only -- the shard's rows plus its halo (fhe_resize_source_rows) --
// yfract the offsets of the rows [row0, row1) only, and out / the consumer's bands the pixels of those rows.
struct RowShard {
    u32 row0, row1, src_row0, n_src_rows;
};
int run_resize_shared(Run &R, const u64 *pixels, u32 src_w, u32 src_h, u32 dst_w, u32 dst_h, RowShard sh, const u64 *xfract, const u64 *yfract, u64 *out,
                      u32 batch, u32 band_rows, fhe_band_consumer consume, void *user) {
    ResizeIndex ix;
    resize_index(src_w, src_h, dst_w, dst_h, ix);
    if (!band_rows) band_rows = 4;
    const u32 rows_per_call = batch / dst_w ? batch / dst_w : 1;
    const u32 n_rows = sh.row1 - sh.row0;
    for (u32 y = sh.row0; y < sh.row1; ++y)
        for (int j = 0; j < 4; ++j) {
            const u32 r = ix.rows_of[y * 4 + j];
            if (r < sh.src_row0 || r - sh.src_row0 >= sh.n_src_rows)
                return fail(FHE_ERR_PARAM, "destination row %u needs source row %u, outside the resident rows [%u, %u)", y, r, sh.src_row0, sh.src_row0 + sh.n_src_rows);
        }
    // t2 (= t3) and the prepared operands once per column / row
    u64 *px2, *px1, *py2, *py1;
    TRY(cubic_powers(R, xfract, dst_w, &px2, &px1));
    TRY(cubic_powers(R, yfract, n_rows, &py2, &py1));
    // the row Cubics' cache: one slot of dst_w size-4 ciphertexts per live source row.  The window only moves down;
    // the number of slots is the largest number of rows alive at once, found by walking the bands
    std::vector<std::vector<u32>> band_need;
    u32 max_live = 0;
    {
        std::set<u32> live;
        for (u32 y0 = sh.row0; y0 < sh.row1; y0 += band_rows) {
            std::vector<u32> need;
            for (u32 y = y0; y < y0 + band_rows && y < sh.row1; ++y)
                for (int j = 0; j < 4; ++j) need.push_back(ix.rows_of[y * 4 + j]);
            std::sort(need.begin(), need.end());
            need.erase(std::unique(need.begin(), need.end()), need.end());
            live.insert(need.begin(), need.end());                      // the band's new rows are formed first ...
            if (live.size() > max_live) max_live = (u32)live.size();
            live.erase(live.begin(), live.lower_bound(need.front()));   // ... then the rows above the window are dropped
            band_need.push_back(need);
        }
    }
    const u32 sr = R.O(4), sout = R.F(6);                               // polynomials of a row Cubic's result and of an output pixel
    const size_t slot_words = (size_t)dst_w * sr * R.pw;
    u64 *cache = R.alloc((size_t)max_live * slot_words);
    const u32 call_px = rows_per_call * dst_w;
    u32 *d_idx = R.alloc_u32((size_t)5 * call_px);                      // four tap arrays + the output slots of one call
    u64 *band = out ? nullptr : R.alloc((size_t)call_px * sout * R.pw);
    std::map<u32, u32> slot_of;                                         // live source row -> slot
    std::vector<u32> free_slots;
    for (u32 s = max_live; s-- > 0;) free_slots.push_back(s);
    std::vector<u32> h((size_t)5 * call_px);
    size_t band_no = 0;
    for (u32 y0 = sh.row0; y0 < sh.row1; y0 += band_rows, ++band_no) {
        const u32 y1 = y0 + band_rows < sh.row1 ? y0 + band_rows : sh.row1;
        const std::vector<u32> &need = band_need[band_no];
        std::vector<u32> fresh;
        for (u32 r : need) if (!slot_of.count(r)) fresh.push_back(r);
        // row Cubics (:296-299) of the new source rows: a function of (output column, source row) only
        for (size_t s0 = 0; s0 < fresh.size(); s0 += rows_per_call) {
            const u32 nr = (u32)(fresh.size() - s0 < rows_per_call ? fresh.size() - s0 : rows_per_call);
            const u32 cnt = nr * dst_w;
            for (u32 i = 0; i < nr; ++i) {
                if (free_slots.empty()) return fail(FHE_ERR_PARAM, "internal: row cache exhausted");
                const u32 r = fresh[s0 + i], slot = free_slots.back();
                free_slots.pop_back();
                slot_of[r] = slot;
                for (u32 x = 0; x < dst_w; ++x) {
                    for (u32 t = 0; t < 4; ++t) h[(size_t)t * cnt + i * dst_w + x] = (r - sh.src_row0) * src_w + ix.colx[x * 4 + t];
                    h[(size_t)4 * cnt + i * dst_w + x] = slot * dst_w + x;
                }
            }
            TRY(R.stage(h.data(), (size_t)5 * cnt, d_idx));
            // pairs are ordered (row, column), column fastest: pair c multiplies the column's xfract / xfract^2 = entry c % dst_w
            TRY(cubic_core(R, Src{pixels, by_index(d_idx)}, Src{pixels, by_index(d_idx + cnt)}, Src{pixels, by_index(d_idx + 2 * (size_t)cnt)},
                           Src{pixels, by_index(d_idx + 3 * (size_t)cnt)}, 2, px2, px1, periodic(1, dst_w), cache, by_index(d_idx + 4 * (size_t)cnt), cnt));
        }
        for (auto it = slot_of.begin(); it != slot_of.end();) {        // the window only moves down
            if (it->first < need.front()) { free_slots.push_back(it->second); it = slot_of.erase(it); }
            else ++it;
        }
        // column Cubics (:303) of the band's output rows
        for (u32 ya = y0; ya < y1; ya += rows_per_call) {
            const u32 nr = y1 - ya < rows_per_call ? y1 - ya : rows_per_call;
            const u32 cnt = nr * dst_w;
            for (u32 i = 0; i < nr; ++i)
                for (u32 x = 0; x < dst_w; ++x)
                    for (u32 j = 0; j < 4; ++j) h[(size_t)j * cnt + i * dst_w + x] = slot_of[ix.rows_of[(ya + i) * 4 + j]] * dst_w + x;
            TRY(R.stage(h.data(), (size_t)4 * cnt, d_idx));
            u64 *dst = out ? out + (size_t)(ya - sh.row0) * dst_w * sout * R.pw : band;
            // pixel c of the call sits in output row ya + c / dst_w: entry ya - row0 + c / dst_w of the prepared yfract batches
            TRY(R.finish_sample(R.O(6), dst, cnt, [&](u64 *d6) {
                return cubic_core(R, Src{cache, by_index(d_idx)}, Src{cache, by_index(d_idx + cnt)}, Src{cache, by_index(d_idx + 2 * (size_t)cnt)},
                                  Src{cache, by_index(d_idx + 3 * (size_t)cnt)}, sr, py2, py1, periodic(dst_w, n_rows, ya - sh.row0), d6, ident(), cnt);
            }));
            if (consume && !R.dry) {
                const int rc = consume(user, (u64)ya * dst_w, cu(dst), cnt, (fhe_stream)R.st);
                if (rc) return fail(rc < 0 ? rc : FHE_ERR_PARAM, "band consumer failed");
            }
        }
    }
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------
// decode path (homo/fhe_decode.h)
// ------------------------------------------------------------------------------------------------
const double kSinCoeffs[5] = {0.5, -1.0 / 24.0, 1.0 / 720.0, -1.0 / 40320.0, 1.0 / 3628800.0};      // :66-112
const double kCosCoeffs[5] = {-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0};     // :146-192
const u32 kTermSize[5] = {3, 5, 7, 9, 11};

// homomorphic_sin / homomorphic_cos: even Taylor polynomial of degree 10 in (x - 3 pi / 2).
// The reference rebuilds every power from a fresh copy of the shifted argument (11 squares, 4 multiplies, :66-112); the
// repeated squares are the same ring elements bit for bit, so each is formed once -- for ALL arguments of a call in one batch,
// sines and cosines alike (the powers do not know which polynomial they will enter).  A segment [first, first + count_t) of the
// arguments then gets its coefficients: res: [count] outputs of size 11, output c = Enc(0) zero[zmap(c)] + constant +
// sum_i coeff_i * power_i of argument first + tmap(c) (:113-118; `broadcast`: count != count_t, the sine polynomials of
// approximated_step do not depend on the position).
// The five multiply_plain calls and the sum are two launches where the context has the pseudo-Mersenne transforms
// (fhe_multiply_plain_sum: one inverse transform per output polynomial); otherwise five multiply_plain calls + k_taylor_sum.
struct TaylorSeg {
    u64 first, count_t;
    const double *coeffs;
    const u64 *zero;
    CMap zmap;
    double constant;
    CMap tmap;
    bool broadcast;
    u64 *res;
    u64 count;
};
int taylor_eval(Run &R, const u64 *x, u64 count_all, const TaylorSeg *segs, int nsegs) {
    const size_t m = R.mark();
    const u64 ct = count_all;
    u64 *pwr[5];
    u32 tsz[5];                                                          // 3, 5, 7, 9, 11; all 2 in the relinearised mode (every power is relinearised where it is formed)
    for (int i = 0; i < 5; ++i) tsz[i] = R.S(kTermSize[i]);
    const u32 sres = R.S(11);
    for (int i = 0; i < 5; ++i) pwr[i] = R.alloc(ct * tsz[i] * R.pw);             // s2, s4, s6, s8, s10
    {
        const size_t m1 = R.mark();
        u64 *sx = R.alloc(ct * 2 * R.pw), *psx = R.prepare_alloc(2, ct), *ps4 = R.prepare_alloc(tsz[1], ct), *tmp = R.alloc(ct * R.S(10) * R.pw);
        TRY(R.copy(sx, x, ct * 2));
        TRY(R.add_plain(sx, 2, ct, R.K(-3 * M_PI / 2.0)));               // :57 / :137
        TRY(R.prepare(sx, 2, ct, psx));                                  // enters five products
        TRY(R.multiply_pp(psx, 2, psx, 2, pwr[0], ct));                  // s2
        TRY(R.square(pwr[0], tsz[0], pwr[1], ct));                       // s4
        TRY(R.prepare(pwr[1], tsz[1], ct, ps4));                         // enters two
        TRY(R.multiply_pp(ps4, tsz[1], ps4, tsz[1], pwr[3], ct));        // s8
        TRY(R.multiply_pp(ps4, tsz[1], psx, 2, tmp, ct));                // s5
        TRY(R.multiply(tmp, R.S(6), nullptr, psx, 2, ident(), pwr[2], ct));      // s6
        TRY(R.multiply(pwr[3], tsz[3], nullptr, psx, 2, ident(), tmp, ct));      // s9
        TRY(R.multiply(tmp, R.S(10), nullptr, psx, 2, ident(), pwr[4], ct));     // s10
        R.release(m1);
    }
    for (int sg = 0; sg < nsegs; ++sg) {
        const TaylorSeg &S = segs[sg];
        const size_t ms = R.mark();
        const CircConst *kc[5];
        u64 *src[5];
        for (int i = 0; i < 5; ++i) {
            kc[i] = R.K(S.coeffs[i]);
            if (!kc[i]) return FHE_ERR_PARAM;
            src[i] = pwr[i] + S.first * tsz[i] * R.pw;
        }
        if (fhe_multiply_plain_sum_supported(R.c)) {
            if (!S.broadcast) {
                TRY(R.mul_plain_sum(5, src, tsz, kc, S.zero, S.zmap, 2, S.res, sres, S.count));
            } else {
                u64 *sum = R.alloc(S.count_t * sres * R.pw);
                TRY(R.mul_plain_sum(5, src, tsz, kc, nullptr, ident(), 0, sum, sres, S.count_t));
                TRY(R.add_general(false, S.zero, 2, S.zmap, sum, sres, S.tmap, S.res, sres, S.count));
            }
        } else {
            for (int i = 0; i < 5; ++i) TRY(R.mul_plain(src[i], src[i], S.count_t * tsz[i], kc[i]));   // every power is dead after its product
            if (!R.dry && S.count) {
                TaylorTerms T;
                for (int i = 0; i < 5; ++i) { T.t[i] = (const ulonglong2 *)src[i]; T.size[i] = tsz[i]; }
                const u64 nrp = S.count * sres * R.k;
                dim3 grid((R.n / 2 + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
                k_taylor_sum<<<grid, 256, 0, R.st>>>((const ulonglong2 *)S.zero, S.zmap, T, S.tmap, (ulonglong2 *)S.res, R.c->qb.d_mod, R.k, R.n / 2, nrp, sres);
                KERNEL_CHECK();
            }
        }
        TRY(R.add_plain(S.res, sres, S.count, R.K(S.constant)));
        R.release(ms);
    }
    R.release(m);
    return FHE_OK;
}

int run_sincos(Run &R, int cosine, const u64 *x, const u64 *zero, u64 *out, u64 count) {
    const TaylorSeg seg = {0, count, cosine ? kCosCoeffs : kSinCoeffs, zero, ident(), cosine ? 1.0 : -1.0, ident(), false, out, count};
    return taylor_eval(R, x, count, &seg, 1);
}

// approximated_step (:202-242) for one run, output positions [pos0, pos1) of the npos = width * height the reference walks
// (the whole run: pos0 = 0, pos1 = npos; a shard of the positions is the multi-GPU partition of the loop at :224).  Batch
// index of the cosine polynomials = (j - 1) * np + (i - pos0), np = pos1 - pos0; only the cheap offset chain (:228-229) is
// walked serially -- from position 0, so a shard replays the add_plain steps of the positions before it -- and the sine
// polynomial is evaluated once per harmonic.  zeros: [np][degree][2][2][k][n], out: [np][so + 1][k][n] (relinearised mode: [np][2][k][n])
int run_step(Run &R, const u64 *amplitude, const u64 *index, const u64 *count_ct, int order, int degree, double delta, u32 pos0, u32 pos1,
             const u64 *zeros, u64 *out) {
    const size_t m = R.mark();
    const u32 np = pos1 - pos0;
    const u64 nb = (u64)np * degree;
    u64 *b = R.alloc(2 * R.pw), *offset = R.alloc(2 * R.pw);
    TRY(R.mul_plain(count_ct, b, 2, R.K(0.5)));                         // :214-215
    TRY(R.add(index, b, offset, 2));                                    // :216-217
    TRY(R.add_plain(offset, 2, 1, R.K(-0.5)));                          // :218
    TRY(R.neg(offset, offset, 2));                                      // :219
    TRY(R.add_plain(b, 2, 1, R.K(delta - 0.5)));                        // :220
    const u32 so = degree >= 1 ? R.S(21) : 2, sty = R.S(11);            // the harmonic sum; a Taylor polynomial
    u64 *cacc = R.alloc((u64)np * so * R.pw);
    {
        u64 *c0 = R.alloc(2 * R.pw);
        TRY(R.mul_plain(b, c0, 2, R.K(1.0 / (double)order)));           // :222-223, the same for every position
        TRY(R.gather_pad(c0, 2, periodic(1, 1), cacc, so, np));
    }
    if (degree >= 1) {
        const size_t m2 = R.mark();
        std::vector<double> factor(degree);
        for (int j = 1; j <= degree; ++j) factor[j - 1] = ((float)j) * M_PI / ((double)order);      // :225
        u64 *args = R.alloc((nb + (u64)degree) * 2 * R.pw), *cos_arg = args, *sin_arg = args + nb * 2 * R.pw;     // one batch of Taylor arguments
        // :228-229: cos_arg(offset), then add_plain(offset, encode(i)) INSIDE the harmonic loop -- the one serial chain of the
        // circuit.  add_plain is an exact addition of Delta m' to c_0, so the value of `offset` at step (i, j) is offset + the sum
        // of the plaintexts added before it: the sums are formed on the host (a few coefficients each: encode(i) has
        // ceil(log2(i + 1)) of them) and ONE launch writes all np * degree arguments -- 2 np degree launches before round 4.
        {
            const fhe_ctx *c = R.c;
            u32 len = 0;
            for (u32 i = 0; i < pos1; ++i) {
                const CircConst *kc = R.K((double)i);
                if (!kc) return FHE_ERR_PARAM;
                if (kc->plain.size() > len) len = (u32)kc->plain.size();
            }
            if (!len) len = 1;
            u64 *d_chain = R.alloc(nb * R.k * len);
            if (!R.dry) {
                std::vector<u64> run((size_t)R.k * len, 0), tab((size_t)nb * R.k * len);
                for (u32 i = 0; i < pos1; ++i) {
                    const CircConst *kc = R.K((double)i);
                    for (int j = 0; j < degree; ++j) {
                        if (i >= pos0) std::copy(run.begin(), run.end(), tab.begin() + ((size_t)j * np + (i - pos0)) * R.k * len);
                        for (u32 pr = 0; pr < R.k; ++pr) {
                            const u64 qi = c->qb.primes[pr];
                            for (size_t x = 0; x < kc->plain.size(); ++x) {
                                const u64 coef = kc->plain[x];
                                if (!coef) continue;
                                u64 v = hostmath::mulmod(c->delta_mod[pr], coef % qi, qi);    // Delta m' as add_plain forms it (ensure_scaled)
                                if (coef >= c->upper_half_threshold) v = hostmath::addmod(v, c->upper_half_increment[pr], qi);
                                run[(size_t)pr * len + x] = hostmath::addmod(run[(size_t)pr * len + x], v, qi);
                            }
                        }
                    }
                }
                TRY(R.stage((const u32 *)tab.data(), tab.size() * 2, (u32 *)d_chain));
                dim3 grid((R.n + 255) / 256, 2 * R.k, (unsigned)(nb < 16384 ? nb : 16384));
                k_offset_chain<<<grid, 256, 0, R.st>>>(offset, d_chain, len, cos_arg, c->qb.d_mod, R.k, R.n, nb);
                KERNEL_CHECK();
            }
        }
        for (int j = 0; j < degree; ++j) {
            u64 *ca = cos_arg + (u64)j * np * 2 * R.pw;
            TRY(R.mul_plain(ca, ca, (u64)np * 2, R.K(factor[j])));                            // :230
            TRY(R.mul_plain(b, sin_arg + (u64)j * 2 * R.pw, 2, R.K(factor[j])));              // :226-227
        }
        u64 *co = R.alloc(nb * sty * R.pw), *si = R.alloc(nb * sty * R.pw);
        u32 *d_idx = R.alloc_u32(2 * nb);
        if (!R.dry) {
            std::vector<u32> h(2 * nb);                                 // Enc(0) of (position i, harmonic j): sin at 2 (i * degree + j), cos next to it
            for (u32 i = 0; i < np; ++i)
                for (int j = 0; j < degree; ++j) {
                    h[(u64)j * np + i] = (u32)(2 * ((u64)i * degree + j));
                    h[nb + (u64)j * np + i] = (u32)(2 * ((u64)i * degree + j) + 1);
                }
            TRY(R.stage(h.data(), h.size(), d_idx));
        }
        const TaylorSeg segs[2] = {{0, nb, kCosCoeffs, zeros, by_index(d_idx + nb), 1.0, ident(), false, co, nb},
                                   {nb, (u64)degree, kSinCoeffs, zeros, by_index(d_idx), -1.0, periodic(np, degree), true, si, nb}};
        TRY(taylor_eval(R, args, nb + (u64)degree, segs, 2));
        u64 *prod = R.alloc(nb * so * R.pw);
        TRY(R.multiply(si, sty, co, nullptr, sty, ident(), prod, nb));                           // :234-235
        if (fhe_multiply_plain_sum_supported(R.c) && degree <= FHE_PLAIN_SUM_MAX_TERMS) {       // :236-237 for every harmonic in one launch
            u64 *src[FHE_PLAIN_SUM_MAX_TERMS];
            const CircConst *kc[FHE_PLAIN_SUM_MAX_TERMS];
            u32 sizes[FHE_PLAIN_SUM_MAX_TERMS];
            for (int j = 0; j < degree; ++j) {
                src[j] = prod + (u64)j * np * so * R.pw;
                sizes[j] = so;
                kc[j] = R.K(2.0 / (M_PI * ((float)(j + 1))));
                if (!kc[j]) return FHE_ERR_PARAM;
            }
            TRY(R.mul_plain_sum(degree, src, sizes, kc, cacc, ident(), so, cacc, so, np));
        } else {
            for (int j = 0; j < degree; ++j) {
                u64 *pj = prod + (u64)j * np * so * R.pw;
                TRY(R.mul_plain(pj, pj, (u64)np * so, R.K(2.0 / (M_PI * ((float)(j + 1))))));      // :236
                TRY(R.add(cacc, pj, cacc, (u64)np * so));                                         // :237
            }
        }
        R.release(m2);
    }
    u64 *pamp = R.prepare_alloc(2, 1);
    TRY(R.prepare(amplitude, 2, 1, pamp));
    TRY(R.multiply(cacc, so, nullptr, pamp, 2, periodic(1, 1), out, np));                      // :239
    R.release(m);
    return FHE_OK;
}

// positions [pos0, pos1) of one channel: acc0, zeros and out hold those positions only; `index` (every shard's own copy)
// advances through all runs exactly as in the whole-channel evaluation
int run_decode_channel(Run &R, const u64 *runs, u32 pairs, u64 *index, const u64 *acc0, const u64 *zeros, int order, int degree, double delta,
                       u32 pos0, u32 pos1, u64 *out) {
    const u32 np = pos1 - pos0;
    const u32 so = pairs ? R.S(fhe_approximated_step_out_size(degree)) : 2;
    TRY(R.gather_pad(acc0, 2, ident(), out, so, np));                   // the channel's Enc(0) accumulators (server_decode.cpp:124-128)
    if (!pairs) return FHE_OK;
    const size_t m = R.mark();
    u64 *run = R.alloc((u64)np * so * R.pw);
    const size_t zstride = (size_t)np * degree * 2 * 2 * R.pw;
    for (u32 p = 0; p < pairs; ++p) {
        const u64 *elem = runs + (size_t)p * 4 * R.pw, *cnt = elem + 2 * R.pw;
        TRY(run_step(R, elem, index, cnt, order, degree, delta, pos0, pos1, zeros + p * zstride, run));       // :133
        TRY(R.add(out, run, out, (u64)np * so));                        // :134-136
        TRY(R.add(index, cnt, index, 2));                               // :137
    }
    R.release(m);
    return FHE_OK;
}

bool args_ok(const fhe_circuits *cc) { return cc && cc->c && cc->c->behz; }
// the decode circuits have no Cubic / Linear whose end the per-Cubic placement could relinearise at
int decode_mode_ok(const fhe_circuits *cc) {
    if (cc && cc->dbc && cc->placement != FHE_RELIN_EVERY_PRODUCT)
        return fail(FHE_ERR_PARAM, "this handle relinearises per Cubic / per sample (FHE_RELIN_PER_CUBIC, FHE_RELIN_PER_SAMPLE): the decode circuits take "
                                   "FHE_RELIN_EVERY_PRODUCT or the reference's mode");
    return FHE_OK;
}

template <typename F>
size_t dry_bytes(const fhe_circuits *cc, F &&f) {
    if (!args_ok(cc)) return 0;
    Run R(cc, nullptr, 0, nullptr, true);
    R.query = true;
    if (f(R)) return 0;
    return R.high + 256;
}
template <typename F>
int real_run(const fhe_circuits *cc, void *scratch, size_t bytes, fhe_stream s, F &&f) {
    if (!args_ok(cc)) return fail(FHE_ERR_PARAM, "null circuits handle (or a context without ct x ct tables)");
    if (!scratch) return fail(FHE_ERR_PARAM, "null scratch");
    const uintptr_t al = ((uintptr_t)scratch + 255) & ~(uintptr_t)255;
    const size_t lost = al - (uintptr_t)scratch;
    // the same host logic once without launches: argument errors surface before anything is enqueued, and the
    // allocator's high-water mark is compared with the scratch actually supplied
    Run D(cc, nullptr, 0, s, true);
    const int rc = f(D);
    if (rc) return rc;
    if (bytes < lost || D.high > bytes - lost) return fail(FHE_ERR_PARAM, "scratch too small: %zu bytes given, %zu needed (use the circuit's *_scratch_bytes())", bytes, D.high + 256);
    Run R(cc, (void *)al, bytes - lost, s, false);
    return f(R);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// seal::Evaluator::add / sub on unequal sizes, batched (include/fhe_hip.h): k_add_general with identity maps
extern "C" int fhe_add_sizes(const fhe_ctx *c, const uint64_t *a, uint32_t size_a, const uint64_t *b, uint32_t size_b, uint64_t *out, uint64_t count, int subtract,
                             fhe_stream s) {
    if (!c || !a || !b || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (size_a < 1 || size_b < 1) return fail(FHE_ERR_PARAM, "ciphertext sizes must be at least 1");
    if (!count) return FHE_OK;
    const u32 so = size_a > size_b ? size_a : size_b, half_n = c->n / 2;
    const u64 nrp = count * so * c->k;
    dim3 grid((half_n + 255) / 256, (unsigned)(nrp < 32768 ? nrp : 32768));
    hipStream_t st = (hipStream_t)s;
    if (subtract) k_add_general<true><<<grid, 256, 0, st>>>((const ulonglong2 *)a, size_a, ident(), (const ulonglong2 *)b, size_b, ident(), (ulonglong2 *)out, so, c->qb.d_mod, c->k, half_n, nrp);
    else k_add_general<false><<<grid, 256, 0, st>>>((const ulonglong2 *)a, size_a, ident(), (const ulonglong2 *)b, size_b, ident(), (ulonglong2 *)out, so, c->qb.d_mod, c->k, half_n, nrp);
    KERNEL_CHECK();
    return FHE_OK;
}

extern "C" int fhe_circuits_create(const fhe_ctx *ctx, int int_coeffs, int frac_coeffs, fhe_circuits **out) {
    return fhe_circuits_create_relin(ctx, int_coeffs, frac_coeffs, nullptr, 0, out);
}
extern "C" uint32_t fhe_circuits_relin_dbc(const fhe_circuits *cc) { return cc ? cc->dbc : 0; }
extern "C" uint32_t fhe_circuits_out_size(const fhe_circuits *cc, int circuit, uint32_t arg) {
    if (!cc) return 0;
    u32 ref;
    switch (circuit) {
        case FHE_CIRC_CUBIC: ref = arg + 2; break;
        case FHE_CIRC_LINEAR: ref = arg + 1; break;
        case FHE_CIRC_SAMPLE_BICUBIC: ref = 6; break;
        case FHE_CIRC_SAMPLE_LINEAR: ref = 4; break;
        case FHE_CIRC_SINCOS: ref = 11; break;
        case FHE_CIRC_STEP: case FHE_CIRC_DECODE: ref = fhe_approximated_step_out_size((int)arg); break;
        default: return 0;
    }
    return cc->dbc && ref > 2 ? 2 : ref;
}
extern "C" int fhe_circuits_create_relin(const fhe_ctx *ctx, int int_coeffs, int frac_coeffs, const uint64_t *d_evk_ntt, uint32_t dbc, fhe_circuits **out) {
    return fhe_circuits_create_relin_at(ctx, int_coeffs, frac_coeffs, d_evk_ntt, dbc, FHE_RELIN_EVERY_PRODUCT, out);
}
extern "C" uint32_t fhe_circuits_relin_placement(const fhe_circuits *cc) { return cc ? cc->placement : 0; }
extern "C" int fhe_circuits_create_relin_at(const fhe_ctx *ctx, int int_coeffs, int frac_coeffs, const uint64_t *d_evk_ntt, uint32_t dbc, uint32_t placement,
                                            fhe_circuits **out) {
    if (!ctx || !out) return fail(FHE_ERR_PARAM, "null argument");
    *out = nullptr;
    if (placement > FHE_RELIN_PER_SAMPLE) return fail(FHE_ERR_PARAM, "unknown relinearisation placement %u", placement);
    if ((d_evk_ntt != nullptr) != (dbc != 0)) return fail(FHE_ERR_PARAM, "evaluation keys and a decomposition bit count come together");
    if (dbc > 60) return fail(FHE_ERR_PARAM, "decomposition bit count out of range");
    if (int_coeffs < 1 || frac_coeffs < 0 || (u32)(int_coeffs + frac_coeffs) > ctx->n) return fail(FHE_ERR_PARAM, "encoder coefficient counts do not fit the polynomial");
    if (int erc = fhe_behz_ensure(ctx)) return erc;                     // a circuits handle is a statement of intent to multiply ciphertexts
    std::unique_ptr<fhe_circuits> cc(new fhe_circuits);
    cc->c = ctx;
    cc->evk = (const u64 *)d_evk_ntt;
    cc->dbc = dbc;
    cc->placement = dbc ? placement : FHE_RELIN_EVERY_PRODUCT;
    cc->ic = int_coeffs;
    cc->fc = frac_coeffs;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipHostMalloc((void **)&cc->pinned, fhe_circuits::kSlots * fhe_circuits::kSlotBytes, hipHostMallocDefault));
    for (int i = 0; i < fhe_circuits::kSlots; ++i) {
        if (hipEventCreateWithFlags(&cc->slot_ev[i], hipEventDisableTiming) != hipSuccess) {
            fhe_circuits_destroy(cc.release());
            return fail(FHE_ERR_HIP, "event creation failed");
        }
    }
    // does the encoder write Cubic's constants the way the fused passes assume?  (base 2: 3 = x + 1, 2 = x, 5 = x^2 + 1, 4 = x^2, 0.5 = -x^(n-1))
    auto is = [&](double v, std::initializer_list<std::pair<u32, u64>> terms) {
        const CircConst *kc = get_const(cc.get(), v);
        if (!kc || kc->nnz != terms.size()) return false;
        for (auto &t : terms)
            if (t.first >= kc->plain.size() || kc->plain[t.first] != t.second) return false;
        return true;
    };
    const u64 t = ctx->t;
    cc->base2 = is(3, {{0, 1}, {1, 1}}) && is(2, {{1, 1}}) && is(5, {{0, 1}, {2, 1}}) && is(4, {{2, 1}}) && is(0.5, {{ctx->n - 1, t - 1}});
    *out = cc.release();
    return FHE_OK;
}
extern "C" int fhe_circuits_destroy(fhe_circuits *cc) {
    if (!cc) return FHE_OK;
    for (auto &kv : cc->consts) {
        if (kv.second->d_ntt) (void)hipFree(kv.second->d_ntt);
        if (kv.second->d_scaled) (void)hipFree(kv.second->d_scaled);
    }
    for (int i = 0; i < fhe_circuits::kSlots; ++i) {
        if (cc->slot_ev[i]) {
            if (cc->slot_used[i]) (void)hipEventSynchronize(cc->slot_ev[i]);
            (void)hipEventDestroy(cc->slot_ev[i]);
        }
    }
    if (cc->pinned) (void)hipHostFree(cc->pinned);
    delete cc;
    return FHE_OK;
}

extern "C" int fhe_resize_sample_plan(uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, int bicubic, uint32_t *taps, double *xfract,
                                      double *yfract) {
    if (!src_w || !src_h || dst_w < 2 || dst_h < 2) return fail(FHE_ERR_PARAM, "image too small for the sampler (the reference divides by width - 1)");
    if ((u64)src_w * src_h > 0xffffffffULL) return fail(FHE_ERR_PARAM, "source image too large");
    const u32 nt = bicubic ? 16 : 4;
    for (u32 y = 0; y < dst_h; ++y) {
        const float v = (float)((float)y / (float)(dst_h - 1) * (float)src_h - 0.5);          // homo/fhe_resize.h:351
        const int yi = (int)v;                                                                // :264 / :229
        for (u32 x = 0; x < dst_w; ++x) {
            const float u = (float)((float)x / (float)(dst_w - 1) * (float)src_w - 0.5);      // :382
            const int xi = (int)u;
            const size_t o = (size_t)y * dst_w + x;
            if (xfract) xfract[o] = (double)(u - floorf(u));                                  // :262
            if (yfract) yfract[o] = (double)(v - floorf(v));                                  // :266
            if (!taps) continue;
            u32 *tp = taps + o * nt;
            if (bicubic) {
                for (int dy = -1; dy <= 2; ++dy)
                    for (int dx = -1; dx <= 2; ++dx)
                        tp[(dy + 1) * 4 + dx + 1] = (u32)clampi(yi + dy, 0, (int)src_h - 1) * src_w + (u32)clampi(xi + dx, 0, (int)src_w - 1);
            } else {
                for (int i = 0; i < 4; ++i)                                                   // p00, p10, p01, p11 (:237-240)
                    tp[i] = (u32)clampi(yi + (i >> 1), 0, (int)src_h - 1) * src_w + (u32)clampi(xi + (i & 1), 0, (int)src_w - 1);
            }
        }
    }
    return FHE_OK;
}

extern "C" size_t fhe_cubic_scratch_bytes(const fhe_circuits *cc, uint32_t size, uint64_t count) {
    return dry_bytes(cc, [&](Run &R) { return run_cubic(R, nullptr, nullptr, nullptr, nullptr, size, nullptr, nullptr, count); });
}
extern "C" int fhe_cubic(const fhe_circuits *cc, const uint64_t *A, const uint64_t *B, const uint64_t *C, const uint64_t *D, uint32_t size,
                         const uint64_t *t, uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!A || !B || !C || !D || !t || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (size < 1) return fail(FHE_ERR_PARAM, "ciphertext sizes must be at least 1");
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) {
        return run_cubic(R, (const u64 *)A, (const u64 *)B, (const u64 *)C, (const u64 *)D, size, (const u64 *)t, (u64 *)out, count);
    });
}
extern "C" size_t fhe_linear_scratch_bytes(const fhe_circuits *cc, uint32_t size, uint64_t count) {
    return dry_bytes(cc, [&](Run &R) { return run_linear(R, nullptr, nullptr, size, nullptr, nullptr, count); });
}
extern "C" int fhe_linear(const fhe_circuits *cc, const uint64_t *A, const uint64_t *B, uint32_t size, const uint64_t *t, uint64_t *out,
                          uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!A || !B || !t || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (size < 1) return fail(FHE_ERR_PARAM, "ciphertext sizes must be at least 1");
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) { return run_linear(R, (const u64 *)A, (const u64 *)B, size, (const u64 *)t, (u64 *)out, count); });
}

extern "C" size_t fhe_sample_bicubic_scratch_bytes(const fhe_circuits *cc, uint64_t count) {
    return dry_bytes(cc, [&](Run &R) { return run_sample_bicubic(R, nullptr, 0, nullptr, nullptr, nullptr, nullptr, count); });
}
extern "C" int fhe_sample_bicubic(const fhe_circuits *cc, const uint64_t *pixels, uint64_t n_pixels, const uint32_t *taps, const uint64_t *xfract,
                                  const uint64_t *yfract, uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!pixels || !taps || !xfract || !yfract || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (!count) return FHE_OK;
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) {
        return run_sample_bicubic(R, (const u64 *)pixels, n_pixels, taps, (const u64 *)xfract, (const u64 *)yfract, (u64 *)out, count);
    });
}
extern "C" size_t fhe_sample_linear_scratch_bytes(const fhe_circuits *cc, uint64_t count) {
    return dry_bytes(cc, [&](Run &R) { return run_sample_linear(R, nullptr, 0, nullptr, nullptr, nullptr, nullptr, count); });
}
extern "C" int fhe_sample_linear(const fhe_circuits *cc, const uint64_t *pixels, uint64_t n_pixels, const uint32_t *taps, const uint64_t *xfract,
                                 const uint64_t *yfract, uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!pixels || !taps || !xfract || !yfract || !out) return fail(FHE_ERR_PARAM, "null argument");
    if (!count) return FHE_OK;
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) {
        return run_sample_linear(R, (const u64 *)pixels, n_pixels, taps, (const u64 *)xfract, (const u64 *)yfract, (u64 *)out, count);
    });
}

static int resize_args(const fhe_circuits *cc, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, uint32_t batch) {
    if (!args_ok(cc)) return fail(FHE_ERR_PARAM, "null circuits handle (or a context without ct x ct tables)");
    if (!src_w || !src_h || dst_w < 2 || dst_h < 2) return fail(FHE_ERR_PARAM, "image too small for the sampler (the reference divides by width - 1)");
    if (!batch) return fail(FHE_ERR_PARAM, "batch must be positive");
    if (!cc->base2) return fail(FHE_ERR_PARAM, "the shared-offset resize needs the base-2 fractional encoder");
    if ((u64)src_w * src_h > 0xffffffffULL || (u64)dst_w * dst_h > 0xffffffffULL) return fail(FHE_ERR_PARAM, "image too large");
    return FHE_OK;
}
extern "C" size_t fhe_resize_bicubic_shared_scratch_bytes(const fhe_circuits *cc, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h,
                                                          uint32_t batch, uint32_t band_rows, int has_out) {
    if (resize_args(cc, src_w, src_h, dst_w, dst_h, batch)) return 0;
    return dry_bytes(cc, [&](Run &R) {
        return run_resize_shared(R, nullptr, src_w, src_h, dst_w, dst_h, RowShard{0, dst_h, 0, src_h}, nullptr, nullptr, has_out ? (u64 *)256 : nullptr, batch, band_rows, nullptr, nullptr);
    });
}
extern "C" int fhe_resize_bicubic_shared(const fhe_circuits *cc, const uint64_t *pixels, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h,
                                         const uint64_t *xfract, const uint64_t *yfract, uint64_t *out, uint32_t batch, uint32_t band_rows,
                                         fhe_band_consumer consume, void *user, void *scratch, size_t scratch_bytes, fhe_stream s) {
    TRY(resize_args(cc, src_w, src_h, dst_w, dst_h, batch));
    if (!pixels || !xfract || !yfract) return fail(FHE_ERR_PARAM, "null argument");
    if (!out && !consume) return fail(FHE_ERR_PARAM, "neither an output buffer nor a consumer");
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) {
        return run_resize_shared(R, (const u64 *)pixels, src_w, src_h, dst_w, dst_h, RowShard{0, dst_h, 0, src_h}, (const u64 *)xfract, (const u64 *)yfract, (u64 *)out, batch, band_rows,
                                 consume, user);
    });
}

// ---- a shard of the destination rows (multi-GPU partition of ResizeImage's outer loop, homo/fhe_resize.h:350) ----
extern "C" int fhe_resize_source_rows(uint32_t src_h, uint32_t dst_h, uint32_t row0, uint32_t row1, int bicubic, uint32_t *first, uint32_t *count) {
    if (!src_h || dst_h < 2 || row0 >= row1 || row1 > dst_h || !first || !count) return fail(FHE_ERR_PARAM, "bad row range");
    u32 lo = 0xffffffffu, hi = 0;
    for (u32 y = row0; y < row1; ++y) {
        const float v = (float)((float)y / (float)(dst_h - 1) * (float)src_h - 0.5);          // homo/fhe_resize.h:351
        const int yi = (int)v;
        const u32 a = (u32)clampi(bicubic ? yi - 1 : yi, 0, (int)src_h - 1), z = (u32)clampi(bicubic ? yi + 2 : yi + 1, 0, (int)src_h - 1);
        if (a < lo) lo = a;
        if (z > hi) hi = z;
    }
    *first = lo;
    *count = hi - lo + 1;
    return FHE_OK;
}
static int shard_args(uint32_t src_h, uint32_t dst_h, uint32_t row0, uint32_t row1, uint32_t src_row0, uint32_t n_src_rows) {
    if (row0 >= row1 || row1 > dst_h) return fail(FHE_ERR_PARAM, "destination rows [%u, %u) are not a range of the %u output rows", row0, row1, dst_h);
    if (!n_src_rows || src_row0 >= src_h || n_src_rows > src_h - src_row0) return fail(FHE_ERR_PARAM, "resident source rows [%u, +%u) are not a range of the %u source rows", src_row0, n_src_rows, src_h);
    return FHE_OK;
}
extern "C" size_t fhe_resize_bicubic_shared_rows_scratch_bytes(const fhe_circuits *cc, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h,
                                                               uint32_t row0, uint32_t row1, uint32_t src_row0, uint32_t n_src_rows, uint32_t batch,
                                                               uint32_t band_rows, int has_out) {
    if (resize_args(cc, src_w, src_h, dst_w, dst_h, batch) || shard_args(src_h, dst_h, row0, row1, src_row0, n_src_rows)) return 0;
    return dry_bytes(cc, [&](Run &R) {
        return run_resize_shared(R, nullptr, src_w, src_h, dst_w, dst_h, RowShard{row0, row1, src_row0, n_src_rows}, nullptr, nullptr, has_out ? (u64 *)256 : nullptr, batch,
                                 band_rows, nullptr, nullptr);
    });
}
extern "C" int fhe_resize_bicubic_shared_rows(const fhe_circuits *cc, const uint64_t *pixels, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h,
                                              uint32_t row0, uint32_t row1, uint32_t src_row0, uint32_t n_src_rows, const uint64_t *xfract,
                                              const uint64_t *yfract, uint64_t *out, uint32_t batch, uint32_t band_rows, fhe_band_consumer consume,
                                              void *user, void *scratch, size_t scratch_bytes, fhe_stream s) {
    TRY(resize_args(cc, src_w, src_h, dst_w, dst_h, batch));
    TRY(shard_args(src_h, dst_h, row0, row1, src_row0, n_src_rows));
    if (!pixels || !xfract || !yfract) return fail(FHE_ERR_PARAM, "null argument");
    if (!out && !consume) return fail(FHE_ERR_PARAM, "neither an output buffer nor a consumer");
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) {
        return run_resize_shared(R, (const u64 *)pixels, src_w, src_h, dst_w, dst_h, RowShard{row0, row1, src_row0, n_src_rows}, (const u64 *)xfract, (const u64 *)yfract,
                                 (u64 *)out, batch, band_rows, consume, user);
    });
}

extern "C" size_t fhe_homomorphic_sincos_scratch_bytes(const fhe_circuits *cc, uint64_t count) {
    if (decode_mode_ok(cc)) return 0;
    return dry_bytes(cc, [&](Run &R) { return run_sincos(R, 0, nullptr, nullptr, nullptr, count); });
}
extern "C" int fhe_homomorphic_sincos(const fhe_circuits *cc, int cosine, const uint64_t *x, const uint64_t *zero, uint64_t *out, uint64_t count,
                                      void *scratch, size_t scratch_bytes, fhe_stream s) {
    if (!x || !zero || !out) return fail(FHE_ERR_PARAM, "null argument");
    TRY(decode_mode_ok(cc));
    if (!count) return FHE_OK;
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) { return run_sincos(R, cosine, (const u64 *)x, (const u64 *)zero, (u64 *)out, count); });
}

extern "C" uint32_t fhe_approximated_step_out_size(int degree) { return degree >= 1 ? 22 : 3; }
static int step_args(const fhe_circuits *cc, int order, int degree, uint32_t npos) {
    if (!args_ok(cc)) return fail(FHE_ERR_PARAM, "null circuits handle (or a context without ct x ct tables)");
    TRY(decode_mode_ok(cc));
    if (order < 1 || degree < 0 || degree > 4096) return fail(FHE_ERR_PARAM, "order must be positive and degree in [0, 4096]");
    if (!npos || (u64)npos * (degree ? degree : 1) > (1u << 24)) return fail(FHE_ERR_PARAM, "width * height * degree out of range");
    return FHE_OK;
}
static int range_args(uint32_t npos, uint32_t pos0, uint32_t pos1) {
    if (pos0 >= pos1 || pos1 > npos) return fail(FHE_ERR_PARAM, "positions [%u, %u) are not a range of the %u output positions", pos0, pos1, npos);
    return FHE_OK;
}
extern "C" size_t fhe_approximated_step_range_scratch_bytes(const fhe_circuits *cc, int degree, uint32_t npos, uint32_t pos0, uint32_t pos1) {
    if (step_args(cc, 1, degree, npos) || range_args(npos, pos0, pos1)) return 0;
    return dry_bytes(cc, [&](Run &R) { return run_step(R, nullptr, nullptr, nullptr, 64, degree, 0.5, pos0, pos1, nullptr, nullptr); });
}
extern "C" size_t fhe_approximated_step_scratch_bytes(const fhe_circuits *cc, int degree, uint32_t npos) {
    return fhe_approximated_step_range_scratch_bytes(cc, degree, npos, 0, npos);
}
extern "C" int fhe_approximated_step_range(const fhe_circuits *cc, const uint64_t *amplitude, const uint64_t *index, const uint64_t *count_ct, int order,
                                           int degree, double delta, uint32_t width, uint32_t height, uint32_t pos0, uint32_t pos1, const uint64_t *zeros,
                                           uint64_t *out, void *scratch, size_t scratch_bytes, fhe_stream s) {
    const u64 np64 = (u64)width * height;
    if (np64 > 0xffffffffULL) return fail(FHE_ERR_PARAM, "width * height out of range");
    TRY(step_args(cc, order, degree, (u32)np64));
    TRY(range_args((u32)np64, pos0, pos1));
    if (!amplitude || !index || !count_ct || !out || (degree > 0 && !zeros)) return fail(FHE_ERR_PARAM, "null argument");
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) {
        return run_step(R, (const u64 *)amplitude, (const u64 *)index, (const u64 *)count_ct, order, degree, delta, pos0, pos1, (const u64 *)zeros, (u64 *)out);
    });
}
extern "C" int fhe_approximated_step(const fhe_circuits *cc, const uint64_t *amplitude, const uint64_t *index, const uint64_t *count_ct, int order,
                                     int degree, double delta, uint32_t width, uint32_t height, const uint64_t *zeros, uint64_t *out, void *scratch,
                                     size_t scratch_bytes, fhe_stream s) {
    const u64 np64 = (u64)width * height;
    if (!np64 || np64 > 0xffffffffULL) return fail(FHE_ERR_PARAM, "width * height out of range");
    return fhe_approximated_step_range(cc, amplitude, index, count_ct, order, degree, delta, width, height, 0, (u32)np64, zeros, out, scratch, scratch_bytes, s);
}
extern "C" size_t fhe_decode_channel_range_scratch_bytes(const fhe_circuits *cc, int degree, uint32_t npos, uint32_t pos0, uint32_t pos1, uint32_t pairs) {
    if (step_args(cc, 1, degree, npos) || range_args(npos, pos0, pos1)) return 0;
    return dry_bytes(cc, [&](Run &R) { return run_decode_channel(R, nullptr, pairs ? 1 : 0, nullptr, nullptr, nullptr, 64, degree, 0.5, pos0, pos1, nullptr); });
}
extern "C" size_t fhe_decode_channel_scratch_bytes(const fhe_circuits *cc, int degree, uint32_t npos, uint32_t pairs) {
    return fhe_decode_channel_range_scratch_bytes(cc, degree, npos, 0, npos, pairs);
}
extern "C" int fhe_decode_channel_range(const fhe_circuits *cc, const uint64_t *runs, uint32_t pairs, uint64_t *index, const uint64_t *acc0,
                                        const uint64_t *zeros, int order, int degree, double delta, uint32_t width, uint32_t height, uint32_t pos0,
                                        uint32_t pos1, uint64_t *out, void *scratch, size_t scratch_bytes, fhe_stream s) {
    const u64 np64 = (u64)width * height;
    if (np64 > 0xffffffffULL) return fail(FHE_ERR_PARAM, "width * height out of range");
    TRY(step_args(cc, order, degree, (u32)np64));
    TRY(range_args((u32)np64, pos0, pos1));
    if (!acc0 || !out || (pairs && (!runs || !index || (degree > 0 && !zeros)))) return fail(FHE_ERR_PARAM, "null argument");
    return real_run(cc, scratch, scratch_bytes, s, [&](Run &R) {
        return run_decode_channel(R, (const u64 *)runs, pairs, (u64 *)index, (const u64 *)acc0, (const u64 *)zeros, order, degree, delta, pos0, pos1, (u64 *)out);
    });
}
extern "C" int fhe_decode_channel(const fhe_circuits *cc, const uint64_t *runs, uint32_t pairs, uint64_t *index, const uint64_t *acc0,
                                  const uint64_t *zeros, int order, int degree, double delta, uint32_t width, uint32_t height, uint64_t *out,
                                  void *scratch, size_t scratch_bytes, fhe_stream s) {
    const u64 np64 = (u64)width * height;
    if (!np64 || np64 > 0xffffffffULL) return fail(FHE_ERR_PARAM, "width * height out of range");
    return fhe_decode_channel_range(cc, runs, pairs, index, acc0, zeros, order, degree, delta, width, height, 0, (u32)np64, out, scratch, scratch_bytes, s);
}
