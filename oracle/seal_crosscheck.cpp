// seal_crosscheck.cpp -- the ciphertext-level cross-check against SEAL 2.3 (pin 5: "if a real SEAL 2.3
// ever appears on a dev box").  TEST INFRASTRUCTURE: nothing of the product links or runs this.
//
// What it is.  A program written against the PUBLIC API OF `seal/seal.h` ONLY -- the seam the reference itself uses
// (homo/fhe_image.h:13 `#include "seal/seal.h"`): EncryptionParameters, SEALContext, KeyGenerator, Encryptor, Decryptor,
// Evaluator, FractionalEncoder, Ciphertext::resize / mutable_pointer / pointer, EvaluationKeys::mutable_data,
// Evaluator::transform_to_ntt.  No facade hook, no C-ABI call, no oracle call.  It therefore compiles
//   (a) against this repository's facade (fully-homomorphic-image-processing_amd/seal/seal.h) on the oracle-backed C ABI (CPU)
//       and on libfhe_hip.so (MI355X) -- both must print tests/golden/seal_crosscheck.json's values (tests/test_seal_crosscheck.py), and
//   (b) against a real SEAL 2.3 (make -C oracle seal23 SEAL_ROOT=/path/to/SEAL): the same lines, produced by Microsoft's code.
// A line that differs under (b) is a statement about SEAL's ciphertext bits that nothing in /root/reference can make today
// (the reference holds no ciphertext vectors and SEAL is an un-vendored submodule); INTEGRATION.md section "SEAL 2.3 cross-check"
// says what a mismatch in each group of lines would mean.
//
// What it computes.  Inputs are the synthetic residues of BASELINE.md section 3, written straight into ciphertext memory:
// word i of a buffer whose first word has global index F is splitmix64(0x5EA12026 ^ (F + i)) mod q_prime(i), layout
// [polynomial][prime][coefficient] (SEAL's).  Every exact-ring Evaluator operation is defined on arbitrary residues, so no
// encryption (no randomness) is needed for the bit-level lines:
//   encode[i]              FractionalEncoder(t, poly, 100, 100, 2).encode of the 24 golden doubles (tests/golden/make_golden.py CONSTS)
//   add, sub, negate, add32  Evaluator::add / sub / negate, add of sizes 3 + 2
//   add_plain[i] / sub_plain[i] / multiply_plain[i]   with each constant (multiply_plain: the non-zero ones)
//   multiply22 / multiply32 / multiply43, square2 / square3     BEHZ ct x ct
//   relin{16,30}_3 / relin{16,30}_4   Evaluator::relinearize of a size-3 and a size-4 ciphertext with evaluation keys whose CONTENTS
//                          are installed through the API: generate_evaluation_keys(dbc, 2, keys) makes the object, then every
//                          key ciphertext is overwritten with seeded residues (coefficient form) and brought to NTT form with
//                          Evaluator::transform_to_ntt -- the key-switch arithmetic is a function of (ciphertext, key words) only
//   cubic / linear         (built with -DCROSSCHECK_REFERENCE_HEADERS and the reference on the include path) the reference's OWN
//                          Cubic and Linear, homo/fhe_resize.h:143-204, on those inputs
//   sin_plain              homomorphic_sin of the reference's own header (homo/fhe_decode.h:48-120) on a REAL encryption of 4.0 under a
//                          fresh key pair: its Enc(0) is drawn inside the header, so the ciphertext is random -- the line is the
//                          DECRYPTED plaintext polynomial (deterministic while the noise budget lasts) and the decoded value
// Each line: `<name> <sha256 of the u64 words, little endian> <16 sampled words in hex>`.
#include "seal/seal.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#ifdef CROSSCHECK_REFERENCE_HEADERS
#include "fhe_resize.h"
#include "fhe_decode.h"
#endif

using namespace seal;

namespace {
// ---- SHA-256 (FIPS 180-4), enough for a few megabytes per line ---------------------------------------------------------------
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    unsigned char buf[64];
    uint64_t len = 0;
    size_t fill = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const unsigned char *p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
            0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
            0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
            0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
            0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void *data, size_t n) {
        const unsigned char *p = (const unsigned char *)data;
        len += n;
        while (n) {
            const size_t take = 64 - fill < n ? 64 - fill : n;
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex() {
        const uint64_t bits = len * 8;
        const unsigned char one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        unsigned char lb[8];
        for (int i = 0; i < 8; ++i) lb[i] = (unsigned char)(bits >> (56 - 8 * i));
        update(lb, 8);
        char out[65];
        for (int i = 0; i < 8; ++i) std::snprintf(out + 8 * i, 9, "%08x", h[i]);
        return std::string(out, 64);
    }
};

uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
constexpr uint64_t SEED = 0x5EA12026ULL;

void line(const char *name, const uint64_t *w, size_t words) {
    Sha256 s;
    s.update(w, words * 8);                                     // x86-64 / little endian hosts only, like everything else here
    std::printf("%s %s", name, s.hex().c_str());
    for (int j = 0; j < 16; ++j) std::printf(" %llx", (unsigned long long)w[((uint64_t)j * 2654435761ULL + 7) % words]);
    std::printf("\n");
}

struct Params {
    EncryptionParameters parms;
    std::vector<uint64_t> q;
    int n;
};

// the synthetic residues, through the API: resize + raw pointer write
void fill(Ciphertext &ct, const Params &P, int size, uint64_t first) {
    ct.resize(P.parms, size);
    uint64_t *p = ct.mutable_pointer();
    uint64_t idx = first;
    for (int poly = 0; poly < size; ++poly)
        for (size_t i = 0; i < P.q.size(); ++i)
            for (int c = 0; c < P.n; ++c, ++idx) p[((size_t)poly * P.q.size() + i) * P.n + c] = splitmix64(SEED ^ idx) % P.q[i];
}
void show(const std::string &name, const Ciphertext &ct, const Params &P) { line(name.c_str(), ct.pointer(), (size_t)ct.size() * P.q.size() * P.n); }
void show_plain(const std::string &name, const Plaintext &p, int n) {
    std::vector<uint64_t> c((size_t)n, 0);
    for (int i = 0; i < p.coeff_count() && i < n; ++i) c[i] = p[i];
    line(name.c_str(), c.data(), c.size());
}

const double CONSTS[24] = {0.541196100, 0.765366865, -1.847759065, 1.175875602, 0.298631336, 2.053119869, 3.072711026, 1.501321110, -0.899976223, -2.562915447,
                           -1.961570560, -0.390180644, 0.125, 128.0, 3.0, 0.5, -0.168736, 1 / 16.0, 1 / 99.0, -4.71238898038469, 0.0, 1.0, -1.0, 255.0};
}  // namespace

int main() {
    // SEAL 2.3.1's coeff_modulus_128(4096) as this repository records it (from memory) -- given EXPLICITLY, so the
    // lines do not depend on what coeff_modulus_128 returns; the first line says whether the two agree on this build
    Params P;
    P.n = 4096;
    P.q = {0x7FFFFFFF380001ULL, 0x3FFFFFFF000001ULL};
    P.parms.set_poly_modulus("1x^4096 + 1");
    std::vector<SmallModulus> mods;
    for (uint64_t v : P.q) mods.push_back(SmallModulus(v));
    P.parms.set_coeff_modulus(mods);
    P.parms.set_plain_modulus(1 << 14);                         // homo/fhe_image.h:26
    {
        const std::vector<SmallModulus> def = coeff_modulus_128(4096);
        bool same = def.size() == P.q.size();
        for (size_t i = 0; same && i < def.size(); ++i) same = def[i].value() == P.q[i];
        std::printf("# coeff_modulus_128(4096) of this build %s the two primes used here (informative, not compared)\n", same ? "IS" : "is NOT");
    }
    SEALContext context(P.parms);
    Evaluator evaluator(context);
    FractionalEncoder encoder(context.plain_modulus(), context.poly_modulus(), 100, 100, 2);      // homo/server_jpeg.cpp:100, homo/fhe_image.h:23-24
    const size_t ctw = 2 * P.q.size() * P.n;                    // words of a size-2 ciphertext: the spacing of the inputs' first indices

    Ciphertext A, B, C3, D4;
    fill(A, P, 2, 0);
    fill(B, P, 2, 1 * ctw);
    fill(C3, P, 3, 2 * ctw);
    fill(D4, P, 4, 4 * ctw);
    show("input_A", A, P);
    show("input_D4", D4, P);

    { Ciphertext x(A); evaluator.add(x, B); show("add", x, P); }
    { Ciphertext x(A); evaluator.sub(x, B); show("sub", x, P); }
    { Ciphertext x(A); evaluator.negate(x); show("negate", x, P); }
    { Ciphertext x(C3); evaluator.add(x, A); show("add32", x, P); }
    { Ciphertext x(A); evaluator.sub(x, C3); show("sub23", x, P); }       // the destination grows (homo/fhe_resize.h:181-184)
    for (int i = 0; i < 24; ++i) {
        const Plaintext p = encoder.encode(CONSTS[i]);
        show_plain("encode[" + std::to_string(i) + "]", p, P.n);
        { Ciphertext x(A); evaluator.add_plain(x, p); show("add_plain[" + std::to_string(i) + "]", x, P); }
        { Ciphertext x(A); evaluator.sub_plain(x, p); show("sub_plain[" + std::to_string(i) + "]", x, P); }
        if (CONSTS[i] != 0.0) { Ciphertext x(A); evaluator.multiply_plain(x, p); show("multiply_plain[" + std::to_string(i) + "]", x, P); }
    }
    Ciphertext P3;
    { Ciphertext x(A); evaluator.multiply(x, B); show("multiply22", x, P); P3 = x; }
    { Ciphertext x(C3); evaluator.multiply(x, A); show("multiply32", x, P); }
    { Ciphertext x(D4); evaluator.multiply(x, C3); show("multiply43", x, P); }
    { Ciphertext x(A); evaluator.square(x); show("square2", x, P); }
    { Ciphertext x(C3); evaluator.square(x); show("square3", x, P); }

    // relinearize with installed keys: the object from the key generator (shape, decomposition bit count), the contents ours
    KeyGenerator keygen(context);
    for (int dbc : {16, 30}) {
        EvaluationKeys evk;
        keygen.generate_evaluation_keys(dbc, 2, evk);           // keys for s^2 and s^3
        std::vector<std::vector<Ciphertext>> &keys = evk.mutable_data();
        uint64_t first = (uint64_t)(100 + dbc) * ctw;
        for (size_t j = 0; j < keys.size(); ++j)
            for (size_t l = 0; l < keys[j].size(); ++l, first += ctw) {
                fill(keys[j][l], P, 2, first);                  // coefficient form ...
                evaluator.transform_to_ntt(keys[j][l]);         // ... to the library's own NTT form
            }
        std::printf("# dbc %d: %zu key sets of %zu keys\n", dbc, keys.size(), keys.empty() ? (size_t)0 : keys[0].size());
        { Ciphertext x(P3); evaluator.relinearize(x, evk); show("relin" + std::to_string(dbc) + "_3", x, P); }
        { Ciphertext x(D4); evaluator.relinearize(x, evk); show("relin" + std::to_string(dbc) + "_4", x, P); }
    }

#ifdef CROSSCHECK_REFERENCE_HEADERS
    {
        PublicKey pk = keygen.public_key();
        SecretKey sk = keygen.secret_key();
        Encryptor encryptor(context, pk);
        Decryptor decryptor(context, sk);
        Ciphertext t, E2, r;
        fill(t, P, 2, 9 * ctw);
        fill(E2, P, 2, 10 * ctw);
        Ciphertext a(A), b(B), c(E2), d(t);
        Cubic(r, a, b, c, d, t, evaluator, encoder, encryptor);              // homo/fhe_resize.h:143-189 (prints its own timing: a ',' line)
        std::printf("\n");
        show("cubic", r, P);
        Linear(r, a, b, t, evaluator, encoder, encryptor);                   // :191-204
        std::printf("\n");
        show("linear", r, P);
        // the deepest circuit the reference has, on a real encryption: the decrypted plaintext polynomial is the deterministic part.
        // n = 8192 with the four 2.3.1 primes: the twelfth power needs the room (homo/client_decode.cpp runs it at 8192 as well)
        Params Q;
        Q.n = 8192;
        Q.q = {0x7FFFFFFF380001ULL, 0x7FFFFFFEF00001ULL, 0x3FFFFFFF000001ULL, 0x3FFFFFFEF40001ULL};
        Q.parms.set_poly_modulus("1x^8192 + 1");
        std::vector<SmallModulus> m8;
        for (uint64_t v : Q.q) m8.push_back(SmallModulus(v));
        Q.parms.set_coeff_modulus(m8);
        Q.parms.set_plain_modulus(1 << 14);
        SEALContext ctx8(Q.parms);
        KeyGenerator kg8(ctx8);
        Encryptor enc8(ctx8, kg8.public_key());
        Decryptor dec8(ctx8, kg8.secret_key());
        Evaluator ev8(ctx8);
        FractionalEncoder fe8(ctx8.plain_modulus(), ctx8.poly_modulus(), 100, 100, 2);
        Ciphertext x, res;
        enc8.encrypt(fe8.encode(4.0), x);
        homomorphic_sin(x, res, ev8, fe8, enc8);                             // homo/fhe_decode.h:48-120
        Plaintext out;
        dec8.decrypt(res, out);
        show_plain("sin_plain", out, Q.n);
        std::printf("sin_value %.9f (sin(4.0) = %.9f; the header's degree-10 Taylor polynomial around 3 pi / 2) size %d budget %s\n", fe8.decode(out), std::sin(4.0),
                    res.size(), dec8.invariant_noise_budget(res) > 0 ? "positive" : "EXHAUSTED");
    }
#endif
    return 0;
}
