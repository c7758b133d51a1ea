// ref_decode_circuit_main.cpp -- TEST INFRASTRUCTURE (built into oracle/_ref/, never committed as binary).
//
// Runs the REFERENCE's own decode circuits -- homomorphic_sin, homomorphic_cos and the homomorphic
// overload of approximated_step exactly as written in /root/reference/homo/fhe_decode.h:48-242 --
// compiled unchanged against this repository's SEAL-shaped facade.  Nothing of the reference is
// copied: the header is included from where it lies by oracle/Makefile's `ref` target.  The Enc(0)
// accumulators the circuits create (:54,134) come from FHE_ENCRYPT_HOOK_FILE (oracle/ref_hook.cpp).
//
// homomorphic_cos is declared to return a Ciphertext but has no return statement (:200, undefined
// behaviour).  This file is therefore built at -O0, where g++ lets such a function
// return normally and the caller destroys a never-constructed temporary; scrub_stack() zeroes the
// stack region below main first, so that temporary reads as an empty Ciphertext (null buffer) and its
// destructor is a no-op.  What the circuit leaves in `res` is what gets compared.
//
// usage: ref_decode_circuit <n> <t> sin|cos <in.bin> <out.bin>
//            in: one ct(2);  out: one ct(11)
//        ref_decode_circuit <n> <t> step <in.bin> <out.bin> <order> <degree> <delta> <width> <height>
//            in: amplitude, index, count (3 ct(2));  out: width*height ct(22)
#include <cstdio>
#include <cstring>
#include <vector>

#include "fhe_decode.h"   // the reference's header, unchanged

static void __attribute__((noinline)) scrub_stack() {
    volatile char z[1 << 20];
    for (size_t i = 0; i < sizeof z; i++) z[i] = 0;
}

int main(int argc, char **argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s n t sin|cos|step in.bin out.bin [order degree delta width height]\n", argv[0]); return 2; }
    const int n = std::atoi(argv[1]);
    const uint64_t t = std::strtoull(argv[2], nullptr, 0);
    const std::string mode = argv[3];
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
    FractionalEncoder encoder(context.plain_modulus(), context.poly_modulus(), 100, 100, 2);   // homo/server_decode.cpp:110

    const uint32_t k = (uint32_t)params.coeff_modulus().size();
    const size_t pw = (size_t)k * n;
    const int n_in = mode == "step" ? 3 : 1;
    std::vector<uint64_t> raw((size_t)n_in * 2 * pw);
    FILE *f = std::fopen(argv[4], "rb");
    if (!f || std::fread(raw.data(), 8, raw.size(), f) != raw.size()) { std::fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
    std::fclose(f);
    std::vector<Ciphertext> in(n_in);
    for (int i = 0; i < n_in; i++) {
        in[i].shape(2, k, (uint32_t)n);
        in[i].buffer().upload(raw.data() + (size_t)i * 2 * pw, 2 * pw);
    }
    std::vector<Ciphertext> out;
    scrub_stack();
    if (mode == "sin") {
        Ciphertext res;
        homomorphic_sin(in[0], res, evaluator, encoder, encryptor);                       // homo/fhe_decode.h:48
        out.push_back(res);
    } else if (mode == "cos") {
        Ciphertext res;
        homomorphic_cos(in[0], res, evaluator, encoder, encryptor);                       // homo/fhe_decode.h:128
        out.push_back(res);
    } else if (mode == "step") {
        if (argc < 11) return 2;
        approximated_step(in[0], in[1], in[2], std::atoi(argv[6]), std::atoi(argv[7]), std::atof(argv[8]), std::atoi(argv[9]), std::atoi(argv[10]),
                          out, evaluator, encoder, encryptor);                            // homo/fhe_decode.h:202
    } else return 2;
    f = std::fopen(argv[5], "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
    for (size_t i = 0; i < out.size(); i++) {
        std::vector<uint64_t> h((size_t)out[i].size() * pw);
        out[i].buffer().download(h.data(), h.size());
        if (std::fwrite(h.data(), 8, h.size(), f) != h.size()) return 2;
    }
    std::fclose(f);
    std::printf("sizes:");
    for (size_t i = 0; i < out.size(); i++) std::printf(" %d", out[i].size());
    std::printf("\n");
    return 0;
}
