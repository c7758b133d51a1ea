// ref_circuits_main.cpp -- TEST INFRASTRUCTURE (built into oracle/_ref/, never committed as binary).
//
// Runs the REFERENCE's own circuit code -- encrypted_dct, quantize_fhe and rgb_to_ycc_fhe exactly as
// written in /root/reference/homo/fhe_image.h:196-325 -- compiled unchanged against this repository's
// SEAL-shaped facade (seal/seal.h -> libfhe_hip.so).  Nothing of the reference is copied: the header
// is included from where it lies (-I/root/reference/homo -I/root/reference/include) by
// oracle/Makefile's `_ref` target, which only exists in the build container.
//
// usage: ref_jpeg_circuit <n> <in.bin> <out.bin>
//   in.bin : (64 + 3) ciphertexts, raw u64 [ct][2][k][n]  (64 block cts, then r, g, b)
//   out.bin: same layout after encrypted_dct + quantize_fhe(YQT) on the block and rgb_to_ycc_fhe on r,g,b
#include <cstdio>
#include <vector>

#include "fhe_image.h"   // the reference's header, unchanged

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s n in.bin out.bin\n", argv[0]); return 2; }
    const int n = std::atoi(argv[1]);
    EncryptionParameters params;
    char poly_mod[32];
    std::snprintf(poly_mod, sizeof poly_mod, "1x^%i + 1", n);
    params.set_poly_modulus(poly_mod);
    params.set_coeff_modulus(coeff_modulus_128(n));
    params.set_plain_modulus(PLAIN_MODULUS);
    SEALContext context(params);
    KeyGenerator keygen(context);
    PublicKey public_key = keygen.public_key();
    Encryptor encryptor(context, public_key);
    Evaluator evaluator(context);
    FractionalEncoder encoder(context.plain_modulus(), context.poly_modulus(), N_NUMBER_COEFFS, N_FRACTIONAL_COEFFS, POLY_BASE);

    const uint32_t k = (uint32_t)params.coeff_modulus().size();
    const size_t ctw = (size_t)2 * k * n;
    std::vector<uint64_t> raw(67 * ctw);
    FILE *f = std::fopen(argv[2], "rb");
    if (!f || std::fread(raw.data(), 8, raw.size(), f) != raw.size()) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    std::fclose(f);
    std::vector<Ciphertext> cts(67);
    for (int i = 0; i < 67; i++) {
        cts[i].shape(2, k, (uint32_t)n);
        cts[i].buffer().upload(raw.data() + i * ctw, ctw);
    }
    std::vector<Ciphertext> block(cts.begin(), cts.begin() + 64);
    encrypted_dct(block, evaluator, encoder, encryptor);                    // homo/fhe_image.h:196
    std::vector<double> quant(YQT, YQT + 64);
    quantize_fhe(block, quant, evaluator, encoder, encryptor);              // homo/fhe_image.h:294
    rgb_to_ycc_fhe(cts[64], cts[65], cts[66], evaluator, encoder, encryptor);   // homo/fhe_image.h:310
    std::printf("\n");
    for (int i = 0; i < 64; i++) block[i].buffer().download(raw.data() + i * ctw, ctw);
    for (int i = 64; i < 67; i++) cts[i].buffer().download(raw.data() + i * ctw, ctw);
    f = std::fopen(argv[3], "wb");
    if (!f || std::fwrite(raw.data(), 8, raw.size(), f) != raw.size()) { std::fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
    std::fclose(f);
    return 0;
}
