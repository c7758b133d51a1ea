// facade_test.cpp -- exercises the SEAL-shaped facade (seal/seal.h) end to end on the GPU:
// keygen, encrypt, every Evaluator operation, relinearize, save/load, decrypt, and the fused
// DCT+quant helper against the plaintext dct() model restated from homo/fhe_image.h:400-484.
// Exit code 0 = all checks passed.  Built by seal/Makefile; run by tests/test_gpu_facade.py.
#include <cstdio>
#include <sstream>

#include "seal/seal.h"
#include "seal/hip_circuits.h"

using namespace seal;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static void plain_line(double *v, int stride, bool scale) {
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = v[i * stride];
    double tmp0 = d[0] + d[7], tmp7 = d[0] - d[7], tmp1 = d[1] + d[6], tmp6 = d[1] - d[6];
    double tmp2 = d[2] + d[5], tmp5 = d[2] - d[5], tmp3 = d[3] + d[4], tmp4 = d[3] - d[4];
    double tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    double o[8];
    o[0] = tmp10 + tmp11;
    o[4] = tmp10 - tmp11;
    double z1 = (tmp12 + tmp13) * 0.541196100;
    o[2] = z1 + tmp13 * 0.765366865;
    o[6] = z1 + tmp12 * -1.847759065;
    z1 = tmp4 + tmp7;
    double z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7, z5 = (z3 + z4) * 1.175875602;
    tmp4 *= 0.298631336; tmp5 *= 2.053119869; tmp6 *= 3.072711026; tmp7 *= 1.501321110;
    z1 *= -0.899976223; z2 *= -2.562915447; z3 *= -1.961570560; z4 *= -0.390180644;
    z3 += z5; z4 += z5;
    o[7] = tmp4 + z1 + z3; o[5] = tmp5 + z2 + z4; o[3] = tmp6 + z2 + z3; o[1] = tmp7 + z1 + z4;
    for (int i = 0; i < 8; i++) v[i * stride] = scale ? o[i] / 8.0 : o[i];
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 4096;
    EncryptionParameters params;
    char poly_mod[32];
    std::snprintf(poly_mod, sizeof poly_mod, "1x^%i + 1", n);
    params.set_poly_modulus(poly_mod);
    params.set_coeff_modulus(coeff_modulus_128(n));
    params.set_plain_modulus(1 << 14);
    SEALContext context(params);
    std::printf("poly_modulus %s, coeff_modulus %d bits, plain_modulus %llu\n", context.poly_modulus().to_string().c_str(),
                context.total_coeff_modulus().significant_bit_count(), (unsigned long long)context.plain_modulus().value());
    KeyGenerator keygen(context);
    PublicKey pk = keygen.public_key();
    SecretKey sk = keygen.secret_key();
    Encryptor encryptor(context, pk);
    Decryptor decryptor(context, sk);
    Evaluator evaluator(context);
    FractionalEncoder encoder(context.plain_modulus(), context.poly_modulus(), 100, 100, 2);
    auto dec = [&](const Ciphertext &c) { Plaintext p; decryptor.decrypt(c, p); return encoder.decode(p); };

    Ciphertext a, b;
    encryptor.encrypt(encoder.encode(37.25), a);
    encryptor.encrypt(encoder.encode(-2.5), b);
    CHECK(a.size() == 2, "fresh ciphertext size");
    CHECK(dec(a) == 37.25 && dec(b) == -2.5, "encrypt/decrypt round trip: %g %g", dec(a), dec(b));
    CHECK(decryptor.invariant_noise_budget(a) > 50, "fresh noise budget %d", decryptor.invariant_noise_budget(a));

    { Ciphertext c(a); evaluator.add(c, b); CHECK(dec(c) == 34.75, "add -> %g", dec(c)); }
    { Ciphertext c(a); evaluator.sub(c, b); CHECK(dec(c) == 39.75, "sub -> %g", dec(c)); }
    { Ciphertext c(a); evaluator.negate(c); CHECK(dec(c) == -37.25, "negate -> %g", dec(c)); }
    { Ciphertext c(a); evaluator.add_plain(c, encoder.encode(0.75)); CHECK(dec(c) == 38.0, "add_plain -> %g", dec(c)); }
    { Ciphertext c(a); evaluator.sub_plain(c, encoder.encode(128.0)); CHECK(dec(c) == 37.25 - 128.0, "sub_plain -> %g", dec(c)); }
    { Ciphertext c(a); evaluator.multiply_plain(c, encoder.encode(0.541196100)); CHECK(std::fabs(dec(c) - 37.25 * 0.541196100) < 1e-9, "multiply_plain -> %.12g", dec(c)); }
    if (!std::getenv("FHE_FACADE_RELIN"))
    { Ciphertext c(a); evaluator.multiply(c, b); CHECK(c.size() == 3 && dec(c) == 37.25 * -2.5, "multiply -> size %d value %g", c.size(), dec(c));
      Ciphertext d(c); evaluator.add(d, a); CHECK(d.size() == 3 && dec(d) == 37.25 * -2.5 + 37.25, "add 3+2 -> %g", dec(d));
      Ciphertext e(a); evaluator.sub(e, c); CHECK(e.size() == 3 && dec(e) == 37.25 - 37.25 * -2.5, "sub 2-3 -> %g", dec(e));
      EvaluationKeys evk; keygen.generate_evaluation_keys(30, evk);
      evaluator.relinearize(c, evk); CHECK(c.size() == 2 && dec(c) == 37.25 * -2.5, "relinearize -> size %d value %g", c.size(), dec(c));
      CHECK(decryptor.invariant_noise_budget(c) > 0, "budget after relinearize"); }
    const bool auto_relin = std::getenv("FHE_FACADE_RELIN") != nullptr;       // every product comes back with two polynomials
    { Ciphertext c(b); evaluator.square(c); CHECK(c.size() == (auto_relin ? 2 : 3) && dec(c) == 6.25, "square -> size %d value %g", c.size(), dec(c));
      Ciphertext d(b); evaluator.multiply(d, b); CHECK(dec(d) == 6.25, "multiply(x,x) -> %g", dec(d)); }
    if (auto_relin) {      // FHE_FACADE_RELIN=<dbc>: a chain of products stays at size 2 and decrypts (keys derived from the secret key the Decryptor was given)
        Ciphertext c(a);
        evaluator.multiply(c, b);
        CHECK(c.size() == 2, "auto-relinearised product has size %d", c.size());
        evaluator.multiply(c, b);
        evaluator.add(c, a);
        CHECK(c.size() == 2 && dec(c) == 37.25 * 6.25 + 37.25, "two relinearised products + add -> size %d value %g", c.size(), dec(c));
        CHECK(decryptor.invariant_noise_budget(c) > 0, "budget after two relinearised products");
    }
    { std::stringstream ss; a.save(ss); b.save(ss); Ciphertext c, d; c.load(ss); d.load(ss);
      CHECK(dec(c) == 37.25 && dec(d) == -2.5, "save/load stream of two ciphertexts");
      std::stringstream ks; pk.save(ks); sk.save(ks); PublicKey pk2; SecretKey sk2; pk2.load(ks); sk2.load(ks);
      Encryptor e2(context, pk2); Decryptor d2(context, sk2); Ciphertext x; e2.encrypt(encoder.encode(5.0), x); Plaintext p; d2.decrypt(x, p);
      CHECK(encoder.decode(p) == 5.0, "key save/load"); }
    { bool threw = false; try { Ciphertext empty; evaluator.negate(empty); } catch (const std::invalid_argument &) { threw = true; } CHECK(threw, "empty ciphertext must throw"); }
    // load() does not trust the stream: header fields are bounded, key records have a fixed polynomial count, and
    // residues must be reduced modulo the moduli of a context of this process
    {
        std::stringstream ss; a.save(ss);
        std::string rec = ss.str();
        auto rejects = [](const std::string &bytes, int which) {
            std::stringstream in(bytes);
            try {
                if (which == 0) { Ciphertext c; c.load(in); } else if (which == 1) { PublicKey k; k.load(in); } else { SecretKey k; k.load(in); }
            } catch (const std::invalid_argument &) { return true; }
            return false;
        };
        CHECK(!rejects(rec, 0), "a saved ciphertext loads");
        std::string bad = rec; uint32_t huge = 0x7fffffffu; std::memcpy(&bad[8], &huge, 4);
        CHECK(rejects(bad, 0), "polynomial count out of range must be rejected before allocating");
        bad = rec; uint32_t n_bad = 12345; std::memcpy(&bad[16], &n_bad, 4);
        CHECK(rejects(bad, 0), "non-power-of-two degree must be rejected");
        bad = rec; uint64_t big = ~0ULL; std::memcpy(&bad[24 + 8 * 17], &big, 8);
        CHECK(rejects(bad, 0), "a residue that is not reduced must be rejected");
        CHECK(rejects(rec.substr(0, rec.size() - 8), 0), "a truncated record must be rejected");
        CHECK(rejects(rec, 2), "a size-2 record is not a secret key");
        std::stringstream ks; sk.save(ks);
        CHECK(rejects(ks.str(), 1), "a secret-key record is not a public key");
    }
    // two encryptions of the same plaintext differ (fresh randomness from the OS-keyed generator every time)
    {
        Ciphertext c1, c2; encryptor.encrypt(encoder.encode(1.0), c1); encryptor.encrypt(encoder.encode(1.0), c2);
        std::stringstream s1, s2; c1.save(s1); c2.save(s2);
        CHECK(s1.str() != s2.str() && dec(c1) == 1.0 && dec(c2) == 1.0, "encryption is randomised");
    }

    // plaintext values: the encoder hands out one shared immutable object per distinct double; a write goes to a private copy
    {
        Plaintext p1 = encoder.encode(0.707106781), p2 = encoder.encode(0.707106781), p3 = encoder.encode(-0.707106781);
        CHECK(p1.frozen_data() && p1.frozen_data() == p2.frozen_data() && p3.frozen_data() != p1.frozen_data(), "one object per distinct value");
        const int len = p1.significant_coeff_count();
        Plaintext q(p1);
        q.data()[0] = 5;                                   // copy on write
        CHECK(!q.frozen_data() && q[0] == 5 && p1[0] != 5 && p2[0] != 5 && p1.significant_coeff_count() == len, "a write must not reach the shared object");
        CHECK(encoder.decode(p2) == encoder.decode(encoder.encode(0.707106781)), "the memo returns the same coefficients");
        Plaintext r = q, u = q;
        u.data()[1] = 7;
        CHECK(r[1] != 7 && u[1] == 7, "two handles of one private plaintext separate on write");
        Ciphertext c1(a), c2(a), c3(a);
        evaluator.multiply_plain(c1, p1); evaluator.multiply_plain(c2, p2);
        evaluator.multiply_plain(c3, Plaintext(std::vector<uint64_t>(p1.data())));   // the same coefficients through the by-value path
        std::stringstream s1, s2, s3; c1.save(s1); c2.save(s2); c3.save(s3);
        CHECK(s1.str() == s2.str() && s1.str() == s3.str(), "products by the shared object and by an equal private plaintext are the same bytes");
        Plaintext z; CHECK(z.significant_coeff_count() == 0 && z.coeff_count() == 0 && z.to_string() == "0", "empty plaintext");
        bool threw = false; try { Ciphertext c(a); evaluator.multiply_plain(c, encoder.encode(0.0)); } catch (const std::invalid_argument &) { threw = true; }
        CHECK(threw, "multiply_plain by encode(0) must throw");
    }

    {   // Encryptor::encrypt is one stream of the library's keyed device sampler (fhe_encrypt_batch): a batch made by
        // seal::hip::DeviceEncryptor under the same (key, index) holds the same ciphertexts, bit for bit
        Encryptor e2(context, pk);
        const uint64_t i0 = e2.next_index();
        const std::vector<double> vals = {0.71875, -3.5, 0.0, 19.0, 0.333251953125};
        std::vector<Ciphertext> one(vals.size());
        for (size_t i = 0; i < vals.size(); ++i) e2.encrypt(encoder.encode(vals[i]), one[i]);
        CHECK(e2.next_index() == i0 + vals.size(), "every encryption takes the next stream of the key");
        hip::DeviceEncryptor de(context, pk, 100, 100, e2.sampler_key().data(), i0);
        hip::CiphertextBatch batch = de.encrypt_values(vals);
        for (size_t i = 0; i < vals.size(); ++i) {
            std::stringstream s1, s2;
            one[i].save(s1);
            batch.get(i).save(s2);
            CHECK(s1.str() == s2.str(), "encryption %d: batch and single call differ", (int)i);
            CHECK(dec(batch.get(i)) == vals[i] && decryptor.invariant_noise_budget(batch.get(i)) > 20, "batched encryption %d decrypts to %g", (int)i, dec(batch.get(i)));
        }
        hip::CiphertextBatch z = de.encrypt_zeros(2);
        std::stringstream z0, z1;
        z.get(0).save(z0); z.get(1).save(z1);
        CHECK(dec(z.get(0)) == 0.0 && dec(z.get(1)) == 0.0 && z0.str() != z1.str(), "two encryptions of zero decrypt to zero and differ");
        Encryptor e3(context, pk);
        CHECK(e3.sampler_key() != e2.sampler_key(), "every Encryptor draws its own key");
        // and the receiving side: one fhe_decrypt_batch for the whole batch = seal::Decryptor one ciphertext at a time (plaintext and budget)
        hip::DeviceDecryptor dd(context, sk);
        std::vector<int> budgets;
        std::vector<Plaintext> plains = dd.decrypt(batch, &budgets);
        CHECK(plains.size() == vals.size() && budgets.size() == vals.size(), "batched decryption returns one plaintext and budget per ciphertext");
        for (size_t i = 0; i < plains.size(); ++i) {
            Plaintext p;
            decryptor.decrypt(one[i], p);
            CHECK(plains[i].data() == p.data() && encoder.decode(plains[i]) == vals[i], "batched decryption %d", (int)i);
            CHECK(budgets[i] == decryptor.invariant_noise_budget(one[i]), "batched budget %d: %d", (int)i, budgets[i]);
        }
        {   // a product (size 3) through both paths
            Ciphertext prod(one[0]);
            evaluator.multiply(prod, one[1]);
            std::vector<Ciphertext> v(1, prod);
            std::vector<int> b3;
            std::vector<Plaintext> p3 = dd.decrypt(hip::CiphertextBatch::from(context, v), &b3);
            CHECK(encoder.decode(p3[0]) == vals[0] * vals[1] && b3[0] == decryptor.invariant_noise_budget(prod) && b3[0] > 0, "batched decryption of a product: %g, budget %d",
                  encoder.decode(p3[0]), b3[0]);
        }
    }

    // fused block circuit on one encrypted 8x8 block
    const std::vector<double> yqt = {16,11,10,16,24,40,51,61,12,12,14,19,26,58,60,55,14,13,16,24,40,57,69,56,14,17,22,29,51,87,80,62,
                                     18,22,37,56,68,109,103,77,24,35,55,64,81,104,113,92,49,64,78,87,103,121,120,101,72,92,95,98,112,100,103,99};
    std::vector<Ciphertext> block(64);
    double plain[64];
    for (int i = 0; i < 64; i++) {
        plain[i] = (double)((37 * (i % 8) + 101 * (i / 8) + 13) % 256) - 128.0;
        encryptor.encrypt(encoder.encode(plain[i]), block[i]);
    }
    hip::dct8x8_quant(context, block, &yqt);
    for (int r = 0; r < 8; r++) plain_line(plain + 8 * r, 1, false);
    for (int c = 0; c < 8; c++) plain_line(plain + c, 8, true);
    double worst = 0;
    int min_budget = 1 << 30;
    for (int i = 0; i < 64; i++) {
        worst = std::max(worst, std::fabs(dec(block[i]) - plain[i] / yqt[i]));
        min_budget = std::min(min_budget, decryptor.invariant_noise_budget(block[i]));
    }
    CHECK(worst < 1e-6 && min_budget > 0, "fused DCT+quant known answer: max err %g, min budget %d", worst, min_budget);
    std::printf("fused DCT+quant: max |err| %.3g, min noise budget %d bits\n", worst, min_budget);
    std::printf(failures ? "FACADE TEST FAILED (%d)\n" : "FACADE TEST OK\n", failures);
    return failures ? 1 : 0;
}
