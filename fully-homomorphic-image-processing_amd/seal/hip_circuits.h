// seal/hip_circuits.h -- the batched resize / decode circuits of include/fhe_circuits.h behind SEAL-typed arguments,
// for C++ hosts (the reference's server mains are C++): what `circuits.py` is to the Python harness.
//
//   seal::hip::CiphertextBatch   `count` ciphertexts of one size in ONE device allocation ([count][size][k][n]); loads and
//                                saves the same stream records as seal::Ciphertext (a saved batch is a concatenation of
//                                Ciphertext::save records), converts to / from std::vector<seal::Ciphertext>
//   seal::hip::DeviceEncryptor   the servers' own encryptions as device batches (fhe_encrypt_batch): encrypt_values / encrypt_zeros ->
//                                CiphertextBatch; with the key and index of a seal::Encryptor it makes that object's ciphertexts
//   seal::hip::DeviceDecryptor   seal::Decryptor::decrypt (+ noise budgets) of a whole CiphertextBatch in one fhe_decrypt_batch
//   seal::hip::Circuits          one fhe_circuits handle + scratch: cubic, linear, sample_bicubic, sample_linear,
//                                resize_bicubic (shared offsets), homomorphic_sin / _cos, approximated_step, decode_channel
//
// Each method is one library call: the taps are index arrays, every temporary lives in the handle's scratch buffer, no
// per-ciphertext work happens on the host.  Results are bit-identical to the reference's functions of the same names
// (homo/fhe_resize.h:143-392, homo/fhe_decode.h:48-242, homo/server_decode.cpp:120-137) called one ciphertext at a
// time through seal::Evaluator with the same server-side encryptions (oracle/ref_vs_batched_main.cpp checks exactly that).
#ifndef FHE_SEAL_HIP_CIRCUITS_H
#define FHE_SEAL_HIP_CIRCUITS_H

#include "fhe_circuits.h"
#include "seal/seal.h"

namespace seal {
namespace hip {

class CiphertextBatch {
public:
    CiphertextBatch() : count_(0), size_(0), k_(0), n_(0) {}
    CiphertextBatch(const SEALContext &ctx, size_t count, uint32_t size) { shape(ctx, count, size); }
    void shape(const SEALContext &ctx, size_t count, uint32_t size) {
        const detail::CtxState &s = *ctx.state();
        count_ = count; size_ = size; k_ = s.k; n_ = s.n;
        buf_.resize(count * ct_words());
    }
    size_t count() const { return count_; }
    uint32_t size() const { return size_; }
    size_t ct_words() const { return (size_t)size_ * k_ * n_; }
    // an EMPTY batch has no allocation, and the C ABI refuses null pointers before it looks at the count: it gets the address of a two-word
    // placeholder, so that count == 0 is the no-op include/fhe_hip.h defines (the Python host does the same, evaluator._ptr)
    uint64_t *ptr() { return buf_.ptr() ? buf_.ptr() : placeholder(); }
    const uint64_t *ptr() const { return buf_.ptr() ? buf_.ptr() : placeholder(); }
    uint64_t *at(size_t i) { return buf_.ptr() + i * ct_words(); }
    const uint64_t *at(size_t i) const { return buf_.ptr() + i * ct_words(); }

    static uint64_t *placeholder() { static detail::DevBuf *p = new detail::DevBuf(2); return p->ptr(); }      // never freed: no destruction-order questions at exit

    // gather / scatter between one-allocation-per-ciphertext objects and the batch (device copies)
    static CiphertextBatch from(const SEALContext &ctx, const std::vector<Ciphertext> &v) {
        CiphertextBatch b;
        if (v.empty()) return b;
        b.shape(ctx, v.size(), (uint32_t)v[0].size());
        for (size_t i = 0; i < v.size(); ++i) {
            if ((uint32_t)v[i].size() != b.size_ || v[i].k() != b.k_ || v[i].n() != b.n_) throw std::invalid_argument("CiphertextBatch: ciphertexts of one size and context only");
            detail::check(fhe_copy(b.at(i), v[i].ptr(), b.ct_words() * 8, nullptr), "copy");
        }
        return b;
    }
    Ciphertext get(size_t i) const {
        if (i >= count_) throw std::out_of_range("CiphertextBatch::get");
        Ciphertext c;
        c.shape(size_, k_, n_);
        detail::check(fhe_copy(c.ptr(), at(i), ct_words() * 8, nullptr), "copy");
        return c;
    }
    void set(size_t i, const Ciphertext &c) {
        if (i >= count_ || (uint32_t)c.size() != size_ || c.k() != k_ || c.n() != n_) throw std::invalid_argument("CiphertextBatch::set");
        detail::check(fhe_copy(at(i), c.ptr(), ct_words() * 8, nullptr), "copy");
    }
    // `count` stream records (Ciphertext::save format) -> the batch, through one host staging buffer
    void load(const SEALContext &ctx, std::istream &is, size_t count, uint32_t size = 2) {
        shape(ctx, count, size);
        std::vector<uint64_t> h(count * ct_words());
        const detail::CtxState &s = *ctx.state();
        for (size_t i = 0; i < count; ++i) {
            char magic[8];
            uint32_t hdr[4];
            is.read(magic, 8);
            is.read((char *)hdr, sizeof hdr);
            if (!is || std::memcmp(magic, "FHEHIP1", 7) != 0) throw std::invalid_argument("stream does not hold a ciphertext");
            if (hdr[0] != size || hdr[1] != k_ || hdr[2] != n_) throw std::invalid_argument("ciphertext record does not match the batch");
            uint64_t *dst = h.data() + i * ct_words();
            is.read((char *)dst, (std::streamsize)(ct_words() * 8));
            if (!is) throw std::invalid_argument("truncated ciphertext stream");
            for (size_t p = 0; p < (size_t)size * k_; ++p) {            // canonical residues only (every kernel assumes them)
                const uint64_t q = s.q[p % k_], *v = dst + p * n_;
                uint64_t bad = 0;
                for (uint32_t c = 0; c < n_; ++c) bad |= (uint64_t)(v[c] >= q);
                if (bad) throw std::invalid_argument("ciphertext holds residues that are not reduced modulo the coefficient moduli");
            }
        }
        if (!h.empty()) buf_.upload(h.data(), h.size());
    }
    void save(std::ostream &os) const {
        std::vector<uint64_t> h(count_ * ct_words());
        if (!h.empty()) { buf_.download(h.data(), h.size()); detail::check(fhe_stream_sync(nullptr), "sync"); }
        const char magic[8] = {'F', 'H', 'E', 'H', 'I', 'P', '1', 0};
        const uint32_t hdr[4] = {size_, k_, n_, 0};
        for (size_t i = 0; i < count_; ++i) {
            os.write(magic, 8);
            os.write((const char *)hdr, sizeof hdr);
            os.write((const char *)(h.data() + i * ct_words()), (std::streamsize)(ct_words() * 8));
        }
    }
    std::vector<uint64_t> to_host() const {
        std::vector<uint64_t> h(count_ * ct_words());
        if (!h.empty()) { buf_.download(h.data(), h.size()); detail::check(fhe_stream_sync(nullptr), "sync"); }
        return h;
    }
private:
    detail::DevBuf buf_;
    size_t count_;
    uint32_t size_, k_, n_;
};

// The encryptions a server makes inside the reference's loops -- frac(x), frac(y) per output pixel (homo/fhe_resize.h:230,234,262,266), an
// encode(0) per homomorphic_sin / cos, accumulator and index (homo/fhe_decode.h:54,134; homo/server_decode.cpp:121,126) -- as ONE
// batch on the device: FractionalEncoder::encode of every value (fhe_frac_encode_batch) and Enc(m) = (Delta m' + pk0 u + e1,
// pk1 u + e2) with u, e1, e2 from the ChaCha20 stream of (key, index) (fhe_encrypt_batch; include/fhe_hip.h).  The key comes from
// the operating system's generator unless one is passed (tests); encryption i of this object's life uses stream i.
class DeviceEncryptor {
public:
    DeviceEncryptor(const SEALContext &ctx, const PublicKey &pk, int int_coeffs = 100, int frac_coeffs = 100, const uint8_t *key = nullptr, uint64_t first_index = 0)
        : ctx_(ctx), ic_(int_coeffs), fc_(frac_coeffs), next_(first_index) {
        const detail::CtxState &s = *ctx.state();
        if (pk.buf.words() != 2 * s.poly_words()) throw std::invalid_argument("public key does not match the context");
        pk_ntt_.resize(2 * s.poly_words());
        detail::check(fhe_ntt_forward(s.h, pk.buf.ptr(), pk_ntt_.ptr(), 2, nullptr), "ntt");
        if (key) std::memcpy(key_.data(), key, 32);
        else key_ = detail::Sampler::fresh_key();
    }
    CiphertextBatch encrypt_values(const std::vector<double> &values) {
        const detail::CtxState &s = *ctx_.state();
        detail::DevBuf plain(values.size() * s.n);
        if (!values.empty()) detail::check(fhe_frac_encode_batch(s.h, values.data(), values.size(), ic_, fc_, plain.ptr(), nullptr), "frac_encode_batch");
        return run(values.empty() ? nullptr : plain.ptr(), values.size());
    }
    CiphertextBatch encrypt_zeros(size_t count) { return run(nullptr, count); }
    uint64_t next_index() const { return next_; }
private:
    CiphertextBatch run(const uint64_t *plain, size_t count) {
        const detail::CtxState &s = *ctx_.state();
        CiphertextBatch out(ctx_, count, 2);
        if (!count) return out;
        std::lock_guard<std::mutex> lk(mu_);                                 // the stream index and the scratch buffer are per object
        if (next_ + count < next_) throw std::runtime_error("DeviceEncryptor: the encryption index would wrap");
        const size_t need = (fhe_encrypt_scratch_bytes(s.h, count) + 7) / 8;
        if (scratch_.words() < need) scratch_.resize(need);
        detail::check(fhe_encrypt_batch(s.h, pk_ntt_.ptr(), plain, count, key_.data(), next_, out.ptr(), scratch_.ptr(), scratch_.words() * 8, nullptr), "encrypt_batch");
        next_ += count;
        return out;
    }
    SEALContext ctx_;
    detail::DevBuf pk_ntt_, scratch_;
    std::array<uint8_t, 32> key_;
    int ic_, fc_;
    uint64_t next_;
    std::mutex mu_;
};

// seal::Decryptor::decrypt of a whole batch (the clients' loops: homo/client_jpeg.cpp:266-280, homo/client_resize.cpp:190-210): ONE
// fhe_decrypt_batch -- phase and exact rounding on the device -- and one download of the plaintext coefficients
class DeviceDecryptor {
public:
    DeviceDecryptor(const SEALContext &ctx, const SecretKey &sk) : ctx_(ctx) {
        const detail::CtxState &s = *ctx.state();
        if (sk.buf.words() != s.poly_words()) throw std::invalid_argument("secret key does not match the context");
        sk_ntt_.resize(s.poly_words());
        detail::check(fhe_ntt_forward(s.h, sk.buf.ptr(), sk_ntt_.ptr(), 1, nullptr), "ntt");
    }
    // plaintexts in batch order; budgets (optional) receives seal::Decryptor::invariant_noise_budget of each ciphertext
    std::vector<Plaintext> decrypt(const CiphertextBatch &cts, std::vector<int> *budgets = nullptr) {
        const detail::CtxState &s = *ctx_.state();
        std::vector<Plaintext> out;
        const size_t count = cts.count();
        if (!count) return out;
        std::lock_guard<std::mutex> lk(mu_);                                 // the scratch buffer is per object
        const size_t need = (fhe_decrypt_scratch_bytes(s.h, cts.size(), count) + 7) / 8, tail = count * s.n + (count + 1) / 2;
        if (scratch_.words() < need + tail) scratch_.resize(need + tail);
        uint64_t *d_plain = scratch_.ptr() + need;
        uint32_t *d_bits = (uint32_t *)(d_plain + count * s.n);
        detail::check(fhe_decrypt_batch(s.h, sk_ntt_.ptr(), cts.ptr(), cts.size(), count, d_plain, d_bits, scratch_.ptr(), need * 8, nullptr), "decrypt_batch");
        std::vector<uint64_t> host(tail);
        scratch_.download(host.data(), tail, need);
        detail::check(fhe_stream_sync(nullptr), "sync");
        const uint32_t *bits = (const uint32_t *)(host.data() + count * s.n);
        const int qbits = (int)fhe_ctx_modulus_bits(s.h);
        for (size_t i = 0; i < count; ++i) {
            out.push_back(Plaintext(std::vector<uint64_t>(host.begin() + i * s.n, host.begin() + (i + 1) * s.n)));
            if (budgets) budgets->push_back(std::max(0, qbits - (int)bits[i] - 1));
        }
        return out;
    }
private:
    SEALContext ctx_;
    detail::DevBuf sk_ntt_, scratch_;
    std::mutex mu_;
};

// sample plan of ResizeImage (homo/fhe_resize.h:350-351,381-382 and the tap order of SampleBicubic / SampleLinear)
struct SamplePlan {
    std::vector<uint32_t> taps;         // [dst_w * dst_h][16 or 4]
    std::vector<double> xfract, yfract; // frac(u), frac(v) per output pixel
    uint32_t taps_per_pixel;
};
inline SamplePlan resize_sample_plan(uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, bool bicubic) {
    SamplePlan p;
    p.taps_per_pixel = bicubic ? 16 : 4;
    const size_t npx = (size_t)dst_w * dst_h;
    p.taps.resize(npx * p.taps_per_pixel);
    p.xfract.resize(npx);
    p.yfract.resize(npx);
    detail::check(fhe_resize_sample_plan(src_w, src_h, dst_w, dst_h, bicubic ? 1 : 0, p.taps.data(), p.xfract.data(), p.yfract.data()), "resize_sample_plan");
    return p;
}

class Circuits {
public:
    // the encoder arguments of seal::FractionalEncoder(t, poly, int_coeffs, frac_coeffs, 2) (homo/server_resize.cpp:110)
    explicit Circuits(const SEALContext &ctx, int int_coeffs = 100, int frac_coeffs = 100) : ctx_(ctx), h_(nullptr), evk_() {
        detail::check(fhe_circuits_create(ctx.state()->h, int_coeffs, frac_coeffs, &h_), "circuits");
    }
    // The RELINEARISED mode (include/fhe_circuits.h fhe_circuits_create_relin; SURVEY.md section 8(f) #4, not what the reference
    // does): every multiply / square of every circuit is followed by evaluator.relinearize with these keys
    // (KeyGenerator::generate_evaluation_keys(dbc, keys): the `dbc` the reference parses and never uses,
    // homo/client_resize.cpp:26,47,72), so every result below has TWO polynomials (out_size()).  The keys are copied.
    // per_cubic = true: the second placement (FHE_RELIN_PER_CUBIC): the reference's Cubic / Linear unchanged and ONE relinearize of each
    // result (size 4 / 3 -> 2); the keys then come from generate_evaluation_keys(dbc, 2, keys) (s^2 and s^3).  Resize circuits only.
    // placement 2 (FHE_RELIN_PER_SAMPLE): the samplers unchanged and ONE relinearize of every output pixel (6 -> 2); keys from
    // generate_evaluation_keys(dbc, 4, keys) (s^2 .. s^5).  `placement` takes the constants of include/fhe_circuits.h (or false / true).
    Circuits(const SEALContext &ctx, const EvaluationKeys &evk, int int_coeffs = 100, int frac_coeffs = 100, int placement = FHE_RELIN_EVERY_PRODUCT) : ctx_(ctx), h_(nullptr), evk_(evk.buf) {
        const bool per_cubic = placement == FHE_RELIN_PER_CUBIC;
        if (per_cubic && evk.count < 2) throw std::invalid_argument("per-Cubic relinearisation needs the keys for s^2 and s^3: generate_evaluation_keys(dbc, 2, keys)");
        if (placement == FHE_RELIN_PER_SAMPLE && evk.count < 4) throw std::invalid_argument("per-sample relinearisation needs the keys for s^2 .. s^5: generate_evaluation_keys(dbc, 4, keys)");
        evk.require_for(*ctx.state(), placement == FHE_RELIN_PER_SAMPLE ? 4u : per_cubic ? 2u : 1u, "Circuits");        // the fields may come from a stream
        evk.device_keys();                                     // key objects handed out by mutable_data() are folded back first
        evk_ = evk.buf;
        detail::check(fhe_circuits_create_relin_at(ctx.state()->h, int_coeffs, frac_coeffs, evk_.ptr(), evk.dbc, (uint32_t)placement, &h_),
                      "circuits (relinearised)");
    }
    // the keys a context under FHE_FACADE_RELIN=<dbc> relinearises with (derived from the first secret key seen on it): a host that
    // mixes the facade's one-at-a-time Evaluator calls with the batched circuits gets the SAME bits from both with these
    static EvaluationKeys context_relin_keys(const SEALContext &ctx) {
        const detail::CtxState &s = *ctx.state();
        if (!s.relin_dbc || !s.relin_evk.words()) throw std::runtime_error("the context has no relinearisation keys (FHE_FACADE_RELIN unset, or no secret key seen yet)");
        EvaluationKeys evk;
        evk.buf = s.relin_evk;
        evk.dbc = s.relin_dbc;
        evk.digits = s.relin_digits;
        evk.count = 1;
        evk.k = s.k;
        evk.n = s.n;
        return evk;
    }
    ~Circuits() { if (h_) fhe_circuits_destroy(h_); }
    // polynomials per output ciphertext of a circuit for this handle (FHE_CIRC_*; arg: operand size or degree)
    uint32_t out_size(int circuit, uint32_t arg = 0) const { return fhe_circuits_out_size(h_, circuit, arg); }
    bool relinearises() const { return fhe_circuits_relin_dbc(h_) != 0; }
    Circuits(const Circuits &) = delete;
    Circuits &operator=(const Circuits &) = delete;

    // Cubic (homo/fhe_resize.h:143-189) for A.count() independent tuples; t has size 2; result size A.size() + 2
    CiphertextBatch cubic(const CiphertextBatch &A, const CiphertextBatch &B, const CiphertextBatch &C, const CiphertextBatch &D, const CiphertextBatch &t) {
        same(A, B); same(A, C); same(A, D); need(t, A.count(), 2);
        CiphertextBatch out(ctx_, A.count(), out_size(FHE_CIRC_CUBIC, A.size()));
        const size_t bytes = scratch(fhe_cubic_scratch_bytes(h_, A.size(), A.count()));
        detail::check(fhe_cubic(h_, A.ptr(), B.ptr(), C.ptr(), D.ptr(), A.size(), t.ptr(), out.ptr(), A.count(), scratch_.ptr(), bytes, nullptr), "cubic");
        return out;
    }
    // Linear (homo/fhe_resize.h:191-204); result size A.size() + 1
    CiphertextBatch linear(const CiphertextBatch &A, const CiphertextBatch &B, const CiphertextBatch &t) {
        same(A, B); need(t, A.count(), 2);
        CiphertextBatch out(ctx_, A.count(), out_size(FHE_CIRC_LINEAR, A.size()));
        const size_t bytes = scratch(fhe_linear_scratch_bytes(h_, A.size(), A.count()));
        detail::check(fhe_linear(h_, A.ptr(), B.ptr(), A.size(), t.ptr(), out.ptr(), A.count(), scratch_.ptr(), bytes, nullptr), "linear");
        return out;
    }
    // SampleBicubic / SampleLinear (homo/fhe_resize.h:222-305) for `count` output pixels of one channel: taps = count x 16 (x 4)
    // indices into `pixels`; xfract / yfract = the offsets' encryptions (:230,234 / :262,266), one pair per output pixel
    CiphertextBatch sample_bicubic(const CiphertextBatch &pixels, const uint32_t *taps, const CiphertextBatch &xfract, const CiphertextBatch &yfract) {
        need(pixels, pixels.count(), 2); need(xfract, xfract.count(), 2); need(yfract, xfract.count(), 2);
        CiphertextBatch out(ctx_, xfract.count(), out_size(FHE_CIRC_SAMPLE_BICUBIC));
        const size_t bytes = scratch(fhe_sample_bicubic_scratch_bytes(h_, xfract.count()));
        detail::check(fhe_sample_bicubic(h_, pixels.ptr(), pixels.count(), taps, xfract.ptr(), yfract.ptr(), out.ptr(), xfract.count(), scratch_.ptr(), bytes, nullptr), "sample_bicubic");
        return out;
    }
    CiphertextBatch sample_linear(const CiphertextBatch &pixels, const uint32_t *taps, const CiphertextBatch &xfract, const CiphertextBatch &yfract) {
        need(pixels, pixels.count(), 2); need(xfract, xfract.count(), 2); need(yfract, xfract.count(), 2);
        CiphertextBatch out(ctx_, xfract.count(), out_size(FHE_CIRC_SAMPLE_LINEAR));
        const size_t bytes = scratch(fhe_sample_linear_scratch_bytes(h_, xfract.count()));
        detail::check(fhe_sample_linear(h_, pixels.ptr(), pixels.count(), taps, xfract.ptr(), yfract.ptr(), out.ptr(), xfract.count(), scratch_.ptr(), bytes, nullptr), "sample_linear");
        return out;
    }
    // ResizeImage with SampleBicubic for one channel of a resident image, one offset ciphertext per output column / row
    // (fhe_resize_bicubic_shared).  Returns dst_w * dst_h size-6 ciphertexts, row-major.
    CiphertextBatch resize_bicubic(const CiphertextBatch &pixels, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h,
                                   const CiphertextBatch &xfract, const CiphertextBatch &yfract, uint32_t batch = 256, uint32_t band_rows = 4) {
        need(pixels, (size_t)src_w * src_h, 2); need(xfract, dst_w, 2); need(yfract, dst_h, 2);
        CiphertextBatch out(ctx_, (size_t)dst_w * dst_h, out_size(FHE_CIRC_SAMPLE_BICUBIC));
        const size_t bytes = scratch(fhe_resize_bicubic_shared_scratch_bytes(h_, src_w, src_h, dst_w, dst_h, batch, band_rows, 1));
        detail::check(fhe_resize_bicubic_shared(h_, pixels.ptr(), src_w, src_h, dst_w, dst_h, xfract.ptr(), yfract.ptr(), out.ptr(), batch, band_rows, nullptr, nullptr,
                                                scratch_.ptr(), bytes, nullptr), "resize_bicubic");
        return out;
    }
    // destination rows [row0, row1) only (fhe_resize_bicubic_shared_rows: a shard of the rows, or one step of a streaming server):
    // `pixels` holds source rows [src_row0, src_row0 + n_src_rows) (fhe_resize_source_rows), yfract the offsets of rows [row0, row1)
    CiphertextBatch resize_bicubic_rows(const CiphertextBatch &pixels, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, uint32_t row0, uint32_t row1,
                                        uint32_t src_row0, uint32_t n_src_rows, const CiphertextBatch &xfract, const CiphertextBatch &yfract, uint32_t batch = 256,
                                        uint32_t band_rows = 4) {
        need(pixels, (size_t)src_w * n_src_rows, 2); need(xfract, dst_w, 2); need(yfract, row1 - row0, 2);
        CiphertextBatch out(ctx_, (size_t)dst_w * (row1 - row0), out_size(FHE_CIRC_SAMPLE_BICUBIC));
        const size_t bytes = scratch(fhe_resize_bicubic_shared_rows_scratch_bytes(h_, src_w, src_h, dst_w, dst_h, row0, row1, src_row0, n_src_rows, batch, band_rows, 1));
        detail::check(fhe_resize_bicubic_shared_rows(h_, pixels.ptr(), src_w, src_h, dst_w, dst_h, row0, row1, src_row0, n_src_rows, xfract.ptr(), yfract.ptr(), out.ptr(), batch,
                                                     band_rows, nullptr, nullptr, scratch_.ptr(), bytes, nullptr), "resize_bicubic_rows");
        return out;
    }
    // the same with the output bands handed to `consume` (e.g. a stream writer) instead of being kept resident
    void resize_bicubic(const CiphertextBatch &pixels, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, const CiphertextBatch &xfract,
                        const CiphertextBatch &yfract, fhe_band_consumer consume, void *user, uint32_t batch = 256, uint32_t band_rows = 4) {
        need(pixels, (size_t)src_w * src_h, 2); need(xfract, dst_w, 2); need(yfract, dst_h, 2);
        const size_t bytes = scratch(fhe_resize_bicubic_shared_scratch_bytes(h_, src_w, src_h, dst_w, dst_h, batch, band_rows, 0));
        detail::check(fhe_resize_bicubic_shared(h_, pixels.ptr(), src_w, src_h, dst_w, dst_h, xfract.ptr(), yfract.ptr(), nullptr, batch, band_rows, consume, user,
                                                scratch_.ptr(), bytes, nullptr), "resize_bicubic");
    }
    // homomorphic_sin / homomorphic_cos (homo/fhe_decode.h:48-120 / :128-200); zero = the Enc(0) of :54 / :134
    CiphertextBatch homomorphic_sin(const CiphertextBatch &x, const CiphertextBatch &zero) { return sincos(0, x, zero); }
    CiphertextBatch homomorphic_cos(const CiphertextBatch &x, const CiphertextBatch &zero) { return sincos(1, x, zero); }
    // approximated_step, homomorphic overload (homo/fhe_decode.h:202-242), one run; zeros: width * height * degree * 2
    // Enc(0)s in the reference's call order
    CiphertextBatch approximated_step(const Ciphertext &amplitude, const Ciphertext &index, const Ciphertext &count, int order, int degree, double delta,
                                      uint32_t width, uint32_t height, const CiphertextBatch &zeros) {
        const size_t npos = (size_t)width * height;
        if (degree > 0) need(zeros, npos * degree * 2, 2);
        CiphertextBatch out(ctx_, npos, out_size(FHE_CIRC_STEP, (uint32_t)degree));
        const size_t bytes = scratch(fhe_approximated_step_scratch_bytes(h_, degree, (uint32_t)npos));
        detail::check(fhe_approximated_step(h_, amplitude.ptr(), index.ptr(), count.ptr(), order, degree, delta, width, height, zeros.ptr(), out.ptr(), scratch_.ptr(), bytes,
                                            nullptr), "approximated_step");
        return out;
    }
    // one channel of the server_decode driver loop (homo/server_decode.cpp:120-137).  runs: (elem, count) per run;
    // index: the channel's Enc(0) index, advanced in place; acc0: width * height Enc(0) accumulators;
    // zeros: runs x width * height x degree x 2 Enc(0)s in call order
    CiphertextBatch decode_channel(const CiphertextBatch &runs, Ciphertext &index, const CiphertextBatch &acc0, const CiphertextBatch &zeros, int order, int degree,
                                   double delta, uint32_t width, uint32_t height) {
        const size_t npos = (size_t)width * height;
        const uint32_t pairs = (uint32_t)(runs.count() / 2);
        need(acc0, npos, 2);
        if (pairs) { need(runs, (size_t)pairs * 2, 2); if (degree > 0) need(zeros, (size_t)pairs * npos * degree * 2, 2); }
        CiphertextBatch out(ctx_, npos, pairs ? out_size(FHE_CIRC_DECODE, (uint32_t)degree) : 2);
        const size_t bytes = scratch(fhe_decode_channel_scratch_bytes(h_, degree, (uint32_t)npos, pairs));
        detail::check(fhe_decode_channel(h_, runs.ptr(), pairs, index.ptr(), acc0.ptr(), zeros.ptr(), order, degree, delta, width, height, out.ptr(), scratch_.ptr(), bytes,
                                         nullptr), "decode_channel");
        return out;
    }
    fhe_circuits *handle() { return h_; }

private:
    CiphertextBatch sincos(int cosine, const CiphertextBatch &x, const CiphertextBatch &zero) {
        need(x, x.count(), 2); need(zero, x.count(), 2);
        CiphertextBatch out(ctx_, x.count(), out_size(FHE_CIRC_SINCOS));
        const size_t bytes = scratch(fhe_homomorphic_sincos_scratch_bytes(h_, x.count()));
        detail::check(fhe_homomorphic_sincos(h_, cosine, x.ptr(), zero.ptr(), out.ptr(), x.count(), scratch_.ptr(), bytes, nullptr), "homomorphic_sincos");
        return out;
    }
    static void same(const CiphertextBatch &a, const CiphertextBatch &b) {
        if (a.count() != b.count() || a.size() != b.size()) throw std::invalid_argument("circuit operands differ in count or size");
    }
    static void need(const CiphertextBatch &a, size_t count, uint32_t size) {
        if (a.count() != count || a.size() != size) throw std::invalid_argument("circuit operand has the wrong count or size");
    }
    size_t scratch(size_t bytes) {
        if (!bytes) throw std::runtime_error(std::string("circuit scratch query: ") + fhe_last_error());
        if (scratch_.words() * 8 < bytes) scratch_.resize((bytes + 7) / 8);
        return bytes;
    }
    SEALContext ctx_;
    fhe_circuits *h_;
    detail::DevBuf evk_;                // relinearised mode: this handle's copy of the evaluation keys (must outlive h_)
    detail::DevBuf scratch_;
};

}  // namespace hip
}  // namespace seal
#endif
