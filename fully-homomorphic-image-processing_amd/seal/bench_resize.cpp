// bench_resize.cpp -- BASELINE.json configs[2] from a C++ host: bicubic resize src x src -> dst x dst of one colour channel
// at n = 8192 through seal::hip::Circuits (seal/hip_circuits.h), the batched C ABI of include/fhe_circuits.h.
// Two forms, as bench_circuits.py times them:
//   per-pixel   fhe_sample_bicubic in batches of `batch` output pixels, one offset ciphertext pair per pixel
//               (what the reference's ResizeImage does, homo/fhe_resize.h:381-388: 5 Cubics per pixel)
//   shared      fhe_resize_bicubic_shared, one offset ciphertext per output column / row
// Inputs are synthetic random-residue ciphertexts resident in HBM; the timed region holds library calls only.
// usage: bench_resize [src=128] [dst=64] [batch=256] [n=8192] [seal23=1]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "seal/hip_circuits.h"

using namespace seal;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const uint32_t src = argc > 1 ? std::atoi(argv[1]) : 128, dst = argc > 2 ? std::atoi(argv[2]) : 64, batch = argc > 3 ? std::atoi(argv[3]) : 256;
    const int n = argc > 4 ? std::atoi(argv[4]) : 8192;
    if (argc <= 5 || std::atoi(argv[5])) setenv("FHE_SEAL23_MODULI", "1", 0);       // SEAL 2.3's four 54/55-bit moduli at n = 8192 (preset P8192)
    EncryptionParameters params;
    char poly_mod[32];
    std::snprintf(poly_mod, sizeof poly_mod, "1x^%i + 1", n);
    params.set_poly_modulus(poly_mod);
    params.set_coeff_modulus(coeff_modulus_128(n));
    params.set_plain_modulus(1 << 14);
    SEALContext context(params);
    const detail::CtxState &st = *context.state();
    hip::Circuits circ(context);
    auto random_batch = [&](size_t count, uint64_t seed) {
        hip::CiphertextBatch b(context, count, 2);
        detail::check(fhe_fill_random(st.h, b.ptr(), count * 2, seed, 0, nullptr), "fill");
        return b;
    };
    const size_t n_out = (size_t)dst * dst;
    hip::CiphertextBatch pixels = random_batch((size_t)src * src, 0x5EA12026ULL);
    hip::SamplePlan plan = hip::resize_sample_plan(src, src, dst, dst, true);
    const size_t P = batch < n_out ? batch : n_out;
    hip::CiphertextBatch xf = random_batch(P, 11), yf = random_batch(P, 12);
    {   // warm-up: constants, scratch
        hip::CiphertextBatch xs = random_batch(8, 11), ys = random_batch(8, 12);
        circ.sample_bicubic(pixels, plan.taps.data(), xs, ys);
        detail::check(fhe_stream_sync(nullptr), "sync");
    }
    double t0 = now();
    size_t done = 0;
    hip::CiphertextBatch out;
    for (size_t s = 0; s + P <= n_out; s += P) {
        out = circ.sample_bicubic(pixels, plan.taps.data() + s * 16, xf, yf);
        done += P;
    }
    detail::check(fhe_stream_sync(nullptr), "sync");
    const double per_pixel = now() - t0;
    hip::CiphertextBatch xc = random_batch(dst, 11), yc = random_batch(dst, 12);
    double shared = 0;
    for (int rep = 0; rep < 2; ++rep) {          // the first pass sizes the scratch buffer
        t0 = now();
        hip::CiphertextBatch o = circ.resize_bicubic(pixels, src, src, dst, dst, xc, yc, batch, 4);
        detail::check(fhe_stream_sync(nullptr), "sync");
        shared = now() - t0;
    }
    std::printf("{\"workload\": \"bicubic resize %ux%u -> %ux%u, one channel, C++ host over include/fhe_circuits.h (n=%u, k=%u)\", \"batch_pixels\": %zu, "
                "\"per_pixel\": {\"output_pixels\": %zu, \"seconds\": %.4f, \"pixels_per_s\": %.1f}, "
                "\"shared_offsets\": {\"output_pixels\": %zu, \"seconds\": %.4f, \"pixels_per_s\": %.1f}}\n",
                src, src, dst, dst, st.n, st.k, P, done, per_pixel, done / per_pixel, n_out, shared, n_out / shared);
    return 0;
}
