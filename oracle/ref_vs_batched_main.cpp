// ref_vs_batched_main.cpp -- TEST INFRASTRUCTURE (built into oracle/_ref/, never committed as binary).
//
// One C++ process, two evaluations of the same circuits on the same ciphertexts, compared bit for bit:
//   (1) the REFERENCE's own functions -- Cubic, Linear, SampleBicubic, SampleLinear (/root/reference/homo/fhe_resize.h:143-305),
//       homomorphic_sin, homomorphic_cos, approximated_step (homo/fhe_decode.h:48-242) and the per-channel loop of
//       homo/server_decode.cpp:120-137 -- included unchanged from where they lie and run one ciphertext at a time through
//       the SEAL-shaped facade;
//   (2) the batched C++ host API of this repository, seal::hip::Circuits (seal/hip_circuits.h over include/fhe_circuits.h):
//       one library call per circuit and batch.
// The circuits' server-side encryptions (fractional offsets, Enc(0) accumulators) are supplied to (1) through the
// facade's test hook and to (2) as arguments, from the same ciphertexts.
//
// usage: ref_vs_batched <n> <t>          (FHE_SEAL23_MODULI=1 selects SEAL 2.3's own moduli, as in the other harnesses)
// FHE_FACADE_RELIN=<dbc>: the RELINEARISED mode on both sides -- (1) the reference's unchanged functions with the facade relinearising
// after every multiply / square, (2) seal::hip::Circuits built with the same keys (fhe_circuits_create_relin): every ciphertext has
// two polynomials and the two evaluations must still agree bit for bit.
// FHE_XCHECK_PER_CUBIC=<dbc>: the SECOND placement of the relinearised mode (include/fhe_circuits.h FHE_RELIN_PER_CUBIC): (1) the reference's
// unchanged Cubic / Linear followed by ONE evaluator.relinearize(result, keys) of the facade (keys for s^2 and s^3 from
// generate_evaluation_keys(dbc, 2, keys): a size-4 ciphertext to 2 in one call, as SEAL's relinearize), the samplers composed from those
// calls exactly as homo/fhe_resize.h:237-248,293-303 compose them; (2) seal::hip::Circuits(context, keys, 100, 100, true).  Resize circuits only.
// FHE_XCHECK_PER_SAMPLE=<dbc>: the THIRD placement (FHE_RELIN_PER_SAMPLE): (1) the reference's UNCHANGED Cubic / Linear / SampleBicubic /
// SampleLinear (sizes grow to 6 / 4 as in its own mode) followed by ONE evaluator.relinearize of each result (keys for s^2 .. s^5 from
// generate_evaluation_keys(dbc, 4, keys)); (2) seal::hip::Circuits(context, keys, 100, 100, FHE_RELIN_PER_SAMPLE).
// Built at -O0 with the stack scrubbed before the decode circuits: homomorphic_cos has no return statement
// (homo/fhe_decode.h:200), see ref_decode_circuit_main.cpp.
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include "fhe_resize.h"   // the reference's headers, unchanged
#include "fhe_decode.h"
#include "seal/hip_circuits.h"

using seal::hip::CiphertextBatch;

static void __attribute__((noinline)) scrub_stack() {
    volatile char z[1 << 20];
    for (size_t i = 0; i < sizeof z; i++) z[i] = 0;
}

static std::deque<Ciphertext> g_hook;        // what the reference's encryptor.encrypt calls return, in order
static int g_fail = 0;

static std::vector<uint64_t> host(const Ciphertext &c) {
    std::vector<uint64_t> h((size_t)c.size() * c.k() * c.n());
    c.buffer().download(h.data(), h.size());
    fhe_stream_sync(nullptr);
    return h;
}
static void expect_equal(const char *what, size_t i, const Ciphertext &ref, const CiphertextBatch &got, size_t gi) {
    const std::vector<uint64_t> a = host(ref), b = host(got.get(gi));
    if ((uint32_t)ref.size() != got.size() || a != b) {
        std::printf("MISMATCH %s[%zu]: reference size %d, batched size %u\n", what, i, ref.size(), got.size());
        ++g_fail;
    }
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s n t\n", argv[0]); return 2; }
    const int n = std::atoi(argv[1]);
    const uint64_t t = std::strtoull(argv[2], nullptr, 0);
    EncryptionParameters params;
    char poly_mod[32];
    std::snprintf(poly_mod, sizeof poly_mod, "1x^%i + 1", n);
    params.set_poly_modulus(poly_mod);
    params.set_coeff_modulus(coeff_modulus_128(n));
    params.set_plain_modulus(t);
    SEALContext context(params);
    KeyGenerator keygen(context);
    PublicKey public_key = keygen.public_key();
    Encryptor encryptor(context, public_key);
    Evaluator evaluator(context);
    FractionalEncoder encoder(context.plain_modulus(), context.poly_modulus(), 100, 100, 2);       // homo/server_resize.cpp:110
    seal::detail::encrypt_hook() = [](const seal::Plaintext &, seal::Ciphertext &out) -> bool {
        if (g_hook.empty()) { std::fprintf(stderr, "encrypt hook exhausted\n"); std::exit(2); }
        out = g_hook.front();
        g_hook.pop_front();
        return true;
    };
    const seal::detail::CtxState &st = *context.state();
    uint64_t seed = 1;
    auto random_batch = [&](size_t count, uint32_t size) {           // random-residue ciphertexts (every op is defined on them)
        CiphertextBatch b(context, count, size);
        seal::detail::check(fhe_fill_random(st.h, b.ptr(), count * size, 0x5EA12026ULL + 977 * seed++, 0, nullptr), "fill");
        return b;
    };
    const int per_cubic = std::getenv("FHE_XCHECK_PER_CUBIC") ? std::atoi(std::getenv("FHE_XCHECK_PER_CUBIC")) : 0;
    const int per_sample = std::getenv("FHE_XCHECK_PER_SAMPLE") ? std::atoi(std::getenv("FHE_XCHECK_PER_SAMPLE")) : 0;
    const bool relin = std::getenv("FHE_FACADE_RELIN") != nullptr || per_cubic;      // either way: every operand and result has two polynomials
    EvaluationKeys evk2;
    if (per_cubic) keygen.generate_evaluation_keys(per_cubic, 2, evk2);
    if (per_sample) keygen.generate_evaluation_keys(per_sample, 4, evk2);
    std::unique_ptr<seal::hip::Circuits> circ_p(per_sample ? new seal::hip::Circuits(context, evk2, 100, 100, FHE_RELIN_PER_SAMPLE)
                                                : per_cubic ? new seal::hip::Circuits(context, evk2, 100, 100, true)
                                                : relin ? new seal::hip::Circuits(context, seal::hip::Circuits::context_relin_keys(context), 100, 100)
                                                        : new seal::hip::Circuits(context, 100, 100));
    seal::hip::Circuits &circ = *circ_p;
    if ((relin || per_sample) != circ.relinearises()) { std::printf("MISMATCH: handle mode\n"); return 1; }
    // the reference's function, then the one relinearize of the per-Cubic placement
    auto ref_cubic = [&](Ciphertext &res, Ciphertext a, Ciphertext b, Ciphertext c, Ciphertext d, Ciphertext tt) {
        Cubic(res, a, b, c, d, tt, evaluator, encoder, encryptor);                                  // homo/fhe_resize.h:143
        if (per_cubic || per_sample) evaluator.relinearize(res, evk2);
    };
    auto ref_linear = [&](Ciphertext &res, Ciphertext a, Ciphertext b, Ciphertext tt) {
        Linear(res, a, b, tt, evaluator, encoder, encryptor);                                       // homo/fhe_resize.h:191
        if (per_cubic || per_sample) evaluator.relinearize(res, evk2);
    };

    {   // an EMPTY batch (a shard that owns nothing) is a no-op with an empty result of the circuit's output size, not a "null argument"
        CiphertextBatch E(context, 0, 2);
        CiphertextBatch c0 = circ.cubic(E, E, E, E, E), l0 = circ.linear(E, E, E);
        if (c0.count() != 0 || l0.count() != 0 || c0.size() != circ.out_size(FHE_CIRC_CUBIC, 2) || l0.size() != circ.out_size(FHE_CIRC_LINEAR, 2)) {
            std::printf("MISMATCH: empty batch\n");
            return 1;
        }
    }
    // ---- Cubic at level 1 (size 2 -> 4) and level 2 (size 4 -> 6), Linear 2 -> 3 and 3 -> 4 ----------------------------
    // (relinearised mode: every operand has two polynomials, so only the first of each)
    for (uint32_t size : relin ? std::vector<uint32_t>{2u} : std::vector<uint32_t>{2u, 4u}) {
        const size_t cnt = 3;
        CiphertextBatch A = random_batch(cnt, size), B = random_batch(cnt, size), C = random_batch(cnt, size), D = random_batch(cnt, size), T = random_batch(cnt, 2);
        CiphertextBatch got = circ.cubic(A, B, C, D, T);
        for (size_t i = 0; i < cnt; ++i) {
            Ciphertext res;
            ref_cubic(res, A.get(i), B.get(i), C.get(i), D.get(i), T.get(i));
            expect_equal(size == 2 ? "Cubic(2)" : "Cubic(4)", i, res, got, i);
        }
    }
    for (uint32_t size : relin ? std::vector<uint32_t>{2u} : std::vector<uint32_t>{2u, 3u}) {
        const size_t cnt = 3;
        CiphertextBatch A = random_batch(cnt, size), B = random_batch(cnt, size), T = random_batch(cnt, 2);
        CiphertextBatch got = circ.linear(A, B, T);
        for (size_t i = 0; i < cnt; ++i) {
            Ciphertext res;
            ref_linear(res, A.get(i), B.get(i), T.get(i));
            expect_equal(size == 2 ? "Linear(2)" : "Linear(3)", i, res, got, i);
        }
    }
    std::printf("\n");

    // ---- SampleBicubic / SampleLinear over an image (the sampling loop of ResizeImage, :350-388) -------------------------
    const int W = 7, H = 6, w = 5, h = 4;
    CiphertextBatch chan[3] = {random_batch(W * H, 2), random_batch(W * H, 2), random_batch(W * H, 2)};
    SImageData image;
    image.width = W; image.height = H; image.start = 0;
    for (int p = 0; p < W * H; ++p) image.pixels.push_back({chan[0].get(p), chan[1].get(p), chan[2].get(p)});
    for (int bicubic = 0; bicubic < 2; ++bicubic) {
        seal::hip::SamplePlan plan = seal::hip::resize_sample_plan(W, H, w, h, bicubic);
        CiphertextBatch xf = random_batch(w * h, 2), yf = random_batch(w * h, 2);
        CiphertextBatch got[3];
        for (int ch = 0; ch < 3; ++ch)
            got[ch] = bicubic ? circ.sample_bicubic(chan[ch], plan.taps.data(), xf, yf) : circ.sample_linear(chan[ch], plan.taps.data(), xf, yf);
        for (int y = 0; y < h; ++y) {
            float v = float(y) / float(h - 1) * float(H) - 0.5;                                     // :351
            for (int x = 0; x < w; ++x) {
                float u = float(x) / float(w - 1) * float(W) - 0.5;                                 // :382
                std::vector<Ciphertext> sample(3);
                if (per_cubic) {            // the samplers' composition (:237-248, :293-303) with the relinearising Cubic / Linear above
                    const uint32_t *tp = plan.taps.data() + (size_t)(y * w + x) * (bicubic ? 16 : 4);
                    for (int ch = 0; ch < 3; ++ch) {
                        Ciphertext xo = xf.get(y * w + x), yo = yf.get(y * w + x);
                        if (bicubic) {
                            Ciphertext col[4];
                            for (int r = 0; r < 4; ++r) ref_cubic(col[r], chan[ch].get(tp[4 * r]), chan[ch].get(tp[4 * r + 1]), chan[ch].get(tp[4 * r + 2]), chan[ch].get(tp[4 * r + 3]), xo);
                            ref_cubic(sample[ch], col[0], col[1], col[2], col[3], yo);
                        } else {
                            Ciphertext c0, c1;
                            ref_linear(c0, chan[ch].get(tp[0]), chan[ch].get(tp[1]), xo);
                            ref_linear(c1, chan[ch].get(tp[2]), chan[ch].get(tp[3]), xo);
                            ref_linear(sample[ch], c0, c1, yo);
                        }
                    }
                } else {
                g_hook.push_back(xf.get(y * w + x));
                g_hook.push_back(yf.get(y * w + x));
                if (bicubic) SampleBicubic(sample, image, u, v, evaluator, encoder, encryptor);     // :386
                else SampleLinear(sample, image, u, v, evaluator, encoder, encryptor);              // :384
                if (per_sample) for (int ch = 0; ch < 3; ++ch) evaluator.relinearize(sample[ch], evk2);      // the ONE relinearize of the placement, 6 / 4 -> 2
                }
                for (int ch = 0; ch < 3; ++ch) expect_equal(bicubic ? "SampleBicubic" : "SampleLinear", (size_t)(y * w + x) * 3 + ch, sample[ch], got[ch], y * w + x);
            }
        }
    }
    {   // shared offsets: one ciphertext per output column / row; every pixel must equal SampleBicubic with (xf[x], yf[y])
        CiphertextBatch xf = random_batch(w, 2), yf = random_batch(h, 2);
        CiphertextBatch got = circ.resize_bicubic(chan[1], W, H, w, h, xf, yf, 8, 3);
        SImageData one;
        one.width = W; one.height = H; one.start = 0;
        for (int p = 0; p < W * H; ++p) one.pixels.push_back({chan[1].get(p), chan[1].get(p), chan[1].get(p)});
        for (int y = 0; y < h; ++y) {
            float v = float(y) / float(h - 1) * float(H) - 0.5;
            for (int x = 0; x < w; ++x) {
                float u = float(x) / float(w - 1) * float(W) - 0.5;
                std::vector<Ciphertext> sample(3);
                if (per_cubic) {
                    seal::hip::SamplePlan plan = seal::hip::resize_sample_plan(W, H, w, h, 1);
                    const uint32_t *tp = plan.taps.data() + (size_t)(y * w + x) * 16;
                    Ciphertext col[4];
                    for (int r = 0; r < 4; ++r) ref_cubic(col[r], chan[1].get(tp[4 * r]), chan[1].get(tp[4 * r + 1]), chan[1].get(tp[4 * r + 2]), chan[1].get(tp[4 * r + 3]), xf.get(x));
                    ref_cubic(sample[0], col[0], col[1], col[2], col[3], yf.get(y));
                } else {
                g_hook.push_back(xf.get(x));
                g_hook.push_back(yf.get(y));
                SampleBicubic(sample, one, u, v, evaluator, encoder, encryptor);
                if (per_sample) evaluator.relinearize(sample[0], evk2);
                }
                expect_equal("resize_bicubic(shared)", (size_t)y * w + x, sample[0], got, (size_t)y * w + x);
            }
        }
    }
    std::printf("\n");

    // ---- decode path (not in the per-Cubic placement: no Cubic to end; the library refuses such a handle there) ------------
    if (per_cubic || per_sample) {
        bool refused = false;
        try { CiphertextBatch X = random_batch(1, 2), Z = random_batch(1, 2); circ.homomorphic_sin(X, Z); } catch (const std::exception &) { refused = true; }
        if (!refused) { std::printf("MISMATCH: the per-Cubic handle evaluated a decode circuit\n"); ++g_fail; }
    } else {
    {
        const size_t cnt = 2;
        CiphertextBatch X = random_batch(cnt, 2), Z = random_batch(cnt, 2);
        CiphertextBatch s = circ.homomorphic_sin(X, Z), c = circ.homomorphic_cos(X, Z);
        for (size_t i = 0; i < cnt; ++i) {
            Ciphertext x = X.get(i), res;
            g_hook.push_back(Z.get(i));
            scrub_stack();
            homomorphic_sin(x, res, evaluator, encoder, encryptor);                                 // homo/fhe_decode.h:48
            expect_equal("homomorphic_sin", i, res, s, i);
            Ciphertext x2 = X.get(i), res2;
            g_hook.push_back(Z.get(i));
            scrub_stack();
            homomorphic_cos(x2, res2, evaluator, encoder, encryptor);                               // :128
            expect_equal("homomorphic_cos", i, res2, c, i);
        }
    }
    {
        const int width = 3, height = 1, degree = 2, order = 64, pairs = 2;
        const double delta = 0.5;
        const size_t npos = (size_t)width * height;
        CiphertextBatch runs = random_batch(2 * pairs, 2), acc0 = random_batch(npos, 2), zeros = random_batch((size_t)pairs * npos * degree * 2, 2), idx0 = random_batch(1, 2);
        // one run through approximated_step alone
        {
            CiphertextBatch z1(context, npos * degree * 2, 2);
            for (size_t i = 0; i < z1.count(); ++i) z1.set(i, zeros.get(i));
            Ciphertext amp = runs.get(0), index = idx0.get(0), cnt = runs.get(1);
            CiphertextBatch got = circ.approximated_step(amp, index, cnt, order, degree, delta, width, height, z1);
            for (size_t i = 0; i < z1.count(); ++i) g_hook.push_back(z1.get(i));
            std::vector<Ciphertext> run;
            scrub_stack();
            approximated_step(amp, index, cnt, order, degree, delta, width, height, run, evaluator, encoder, encryptor);      // :202
            for (size_t i = 0; i < npos; ++i) expect_equal("approximated_step", i, run[i], got, i);
        }
        // the per-channel driver loop, homo/server_decode.cpp:120-137, with the homomorphic overload
        Ciphertext index_b = idx0.get(0);
        CiphertextBatch got = circ.decode_channel(runs, index_b, acc0, zeros, order, degree, delta, width, height);
        Ciphertext index = idx0.get(0);                                                             // :121
        std::vector<Ciphertext> channel;
        for (size_t j = 0; j < npos; ++j) channel.push_back(acc0.get(j));                           // :124-128
        for (size_t i = 0; i < zeros.count(); ++i) g_hook.push_back(zeros.get(i));
        for (int j = 0; j < pairs; ++j) {
            std::vector<Ciphertext> run;
            Ciphertext elem = runs.get(2 * j), count = runs.get(2 * j + 1);                         // :131-132
            scrub_stack();
            approximated_step(elem, index, count, order, degree, delta, width, height, run, evaluator, encoder, encryptor);   // :133 (homomorphic overload)
            for (size_t k = 0; k < npos; ++k) evaluator.add(channel[k], run[k]);                    // :134-136
            evaluator.add(index, count);                                                            // :137
        }
        for (size_t k = 0; k < npos; ++k) expect_equal("decode_channel", k, channel[k], got, k);
        if (host(index) != host(index_b)) { std::printf("MISMATCH decode_channel index\n"); ++g_fail; }
    }
    }
    std::printf("\n");
    if (!g_hook.empty()) { std::printf("MISMATCH: %zu hook ciphertexts unused\n", g_hook.size()); ++g_fail; }
    std::printf(g_fail ? "FAILED: %d mismatches\n" : "OK: reference circuits == batched C++ API, bit for bit\n", g_fail);
    return g_fail ? 1 : 0;
}
