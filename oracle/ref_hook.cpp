// ref_hook.cpp -- TEST INFRASTRUCTURE, linked into the oracle/_ref/ binaries (never into the product).
//
// The reference's resize and decode circuits encrypt on the SERVER side, inside the circuit
// (homo/fhe_resize.h:230,234,262,266: the fractional sample offsets; homo/fhe_decode.h:54,134: the
// Enc(0) accumulators), so their outputs are randomised.  When the environment variable
// FHE_ENCRYPT_HOOK_FILE names a file of raw size-2 ciphertexts (u64 [2][k][n] each, in call order),
// this translation unit makes seal::Encryptor::encrypt return those instead (the test-only hook that
// seal/seal.h compiles in under -DFHE_FACADE_TEST_HOOKS), which makes the reference's own code
// comparable bit for bit with the oracle and with circuits.py on the same inputs.  Without the
// variable nothing is installed and encryption is the real one.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "seal/seal.h"

namespace {
struct HookInstaller {
    HookInstaller() {
        const char *path = std::getenv("FHE_ENCRYPT_HOOK_FILE");
        if (!path || !*path) return;
        FILE *f = std::fopen(path, "rb");
        if (!f) { std::fprintf(stderr, "ref_hook: cannot open %s\n", path); std::exit(2); }
        seal::detail::encrypt_hook() = [f](const seal::Plaintext &, seal::Ciphertext &out) -> bool {
            const std::vector<seal::detail::KnownModuli> known = seal::detail::known_moduli();      // a snapshot (by value)
            if (known.empty()) { std::fprintf(stderr, "ref_hook: encrypt before any context\n"); std::exit(2); }
            const seal::detail::KnownModuli &m = known.back();
            const size_t words = (size_t)2 * m.k * m.n;
            std::vector<uint64_t> buf(words);
            if (std::fread(buf.data(), 8, words, f) != words) { std::fprintf(stderr, "ref_hook: hook file exhausted\n"); std::exit(2); }
            out.shape(2, m.k, m.n);
            out.buffer().upload(buf.data(), words);
            return true;
        };
    }
} installer;

// FHE_DECODE_LOG_FILE: every value FractionalEncoder::decode returns is appended to this file as a raw double, in call
// order -- what the reference's client sees before its `int pixel = ...; CLAMP(...); (uint8_t) pixel` conversion.
struct DecodeLogInstaller {
    DecodeLogInstaller() {
        const char *path = std::getenv("FHE_DECODE_LOG_FILE");
        if (!path || !*path) return;
        FILE *f = std::fopen(path, "ab");
        if (!f) { std::fprintf(stderr, "ref_hook: cannot open %s\n", path); std::exit(2); }
        seal::detail::decode_hook() = [f](double v) {
            std::fwrite(&v, sizeof v, 1, f);
            std::fflush(f);
        };
    }
} decode_log_installer;
}  // namespace
