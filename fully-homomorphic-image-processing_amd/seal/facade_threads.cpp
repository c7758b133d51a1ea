// facade_threads.cpp -- the SEAL facade (seal/seal.h) under several host threads, and the failure paths of its lazy graph.
//
// SEAL's Evaluator may be used from several threads on distinct ciphertexts; the reference itself is single-threaded
// (homo/server_jpeg.cpp:113-153), but a server that handles requests concurrently is the obvious next host.  The lazy mode
// (default) records every Evaluator call into ONE per-context graph, so recording, flushing and materialising are serialised
// by a per-context mutex; the device-buffer pool, the Evaluator's plaintext caches and the encoder's memo have their own.
// Checked here, in both modes (FHE_FACADE_EAGER=1 for the eager one), on the GPU (libfhe_hip.so) and -- built by
// oracle/Makefile targets `asan` / `tsan` -- on the CPU backend under AddressSanitizer / UBSan / ThreadSanitizer:
//   1. two contexts, one thread each, the same call sequence: the same decrypted values;
//   2. ONE context and ONE Evaluator shared by T threads, each on its own ciphertexts, flushing at different moments
//      (decrypts interleave with the other threads' recordings);
//   3. ciphertext copies handed from thread to thread (handle counts are atomic);
//   4. a flush that throws: values it did not reach are marked failed -- a later Evaluator call on one of them, or a save /
//      decrypt, throws std::runtime_error instead of dereferencing a null buffer (needs -DFHE_FACADE_TEST_HOOKS);
//   5. a pending ciphertext that outlives its SEALContext: decrypting it throws, nothing dangles.
// Exit code 0 = all checks passed.
#include <atomic>
#include <cstdio>
#include <sstream>
#include <thread>

#include "seal/seal.h"

using namespace seal;

static std::atomic<int> failures(0);
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static EncryptionParameters make_params(int n) {
    EncryptionParameters params;
    char poly_mod[32];
    std::snprintf(poly_mod, sizeof poly_mod, "1x^%i + 1", n);
    params.set_poly_modulus(poly_mod);
    params.set_coeff_modulus(coeff_modulus_128(n));
    params.set_plain_modulus(1 << 14);
    return params;
}

// a small circuit with linear ops, a plaintext product and one ciphertext product: (x + y) * 0.5 - y, then squared
static double circuit(Evaluator &ev, FractionalEncoder &enc, const Ciphertext &x, const Ciphertext &y, Decryptor &dec, int rounds) {
    double got = 0;
    for (int r = 0; r < rounds; ++r) {
        Ciphertext a(x);
        ev.add(a, y);
        ev.multiply_plain(a, enc.encode(0.5));
        ev.sub(a, y);
        Ciphertext b(a);                         // alias of a pending value
        ev.square(b);
        ev.add_plain(b, enc.encode((double)r));
        Plaintext p;
        dec.decrypt(b, p);                       // observation: flushes whatever all threads have recorded so far
        got = enc.decode(p);
    }
    return got;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 4096;
    const int T = argc > 2 ? std::atoi(argv[2]) : 4;
    const int rounds = argc > 3 ? std::atoi(argv[3]) : 6;

    // ---- 1. two contexts, one thread each -------------------------------------------------------------------------------
    {
        double out[2] = {0, 0};
        auto job = [&](int which) {
            SEALContext context(make_params(n));
            KeyGenerator keygen(context);
            Encryptor encryptor(context, keygen.public_key());
            Decryptor decryptor(context, keygen.secret_key());
            Evaluator evaluator(context);
            FractionalEncoder encoder(context.plain_modulus(), context.poly_modulus(), 100, 100, 2);
            Ciphertext x, y;
            encryptor.encrypt(encoder.encode(6.5), x);
            encryptor.encrypt(encoder.encode(-1.25), y);
            out[which] = circuit(evaluator, encoder, x, y, decryptor, rounds);
        };
        std::thread t0(job, 0), t1(job, 1);
        t0.join(); t1.join();
        const double want = ((6.5 - 1.25) * 0.5 + 1.25) * ((6.5 - 1.25) * 0.5 + 1.25) + (rounds - 1);
        CHECK(out[0] == want && out[1] == want, "two contexts: %g %g, want %g", out[0], out[1], want);
    }

    // ---- 2. + 3. one context, one Evaluator, T threads ------------------------------------------------------------------
    SEALContext context(make_params(n));
    KeyGenerator keygen(context);
    Encryptor encryptor(context, keygen.public_key());
    Decryptor decryptor(context, keygen.secret_key());
    FractionalEncoder encoder(context.plain_modulus(), context.poly_modulus(), 100, 100, 2);
    {
        Evaluator evaluator(context);
        std::vector<Ciphertext> xs(T), ys(T);
        for (int i = 0; i < T; ++i) {
            encryptor.encrypt(encoder.encode(2.0 + i), xs[i]);
            encryptor.encrypt(encoder.encode(0.5 * i), ys[i]);
        }
        std::vector<double> got(T, 0.0);
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i)
            th.emplace_back([&, i] {
                Decryptor my_dec(context, keygen.secret_key());        // a Decryptor holds per-object scratch: one per thread, as with SEAL
                got[i] = circuit(evaluator, encoder, xs[i], ys[i], my_dec, rounds + i);
            });
        for (auto &t : th) t.join();
        for (int i = 0; i < T; ++i) {
            const double v = (2.0 + i + 0.5 * i) * 0.5 - 0.5 * i, want = v * v + (rounds + i - 1);
            CHECK(got[i] == want, "shared evaluator, thread %d: %g, want %g", i, got[i], want);
        }
        // copies of ONE pending value made and dropped by all threads at once, then each thread extends its copy
        Ciphertext shared(xs[0]);
        evaluator.add(shared, ys[1]);                                  // pending
        std::vector<double> got2(T, 0.0);
        th.clear();
        for (int i = 0; i < T; ++i)
            th.emplace_back([&, i] {
                Decryptor my_dec(context, keygen.secret_key());
                for (int r = 0; r < 50; ++r) { Ciphertext c(shared); (void)c; }
                Ciphertext mine(shared);
                evaluator.add_plain(mine, encoder.encode((double)i));
                Plaintext p;
                my_dec.decrypt(mine, p);
                got2[i] = encoder.decode(p);
            });
        for (auto &t : th) t.join();
        for (int i = 0; i < T; ++i) CHECK(got2[i] == 2.0 + 0.5 + i, "copies of a pending value, thread %d: %g", i, got2[i]);
        // ONE Encryptor shared by all threads (SEAL's Encryptor is used that way by the reference's helpers): every call takes its own
        // stream index of the object's sampler key under the object's mutex -- all results decrypt, no index is handed out twice
        const uint64_t before = encryptor.next_index();
        std::vector<double> got3(T, 0.0);
        th.clear();
        for (int i = 0; i < T; ++i)
            th.emplace_back([&, i] {
                Decryptor my_dec(context, keygen.secret_key());
                Ciphertext c;
                for (int r = 0; r < 8; ++r) encryptor.encrypt(encoder.encode(1.5 * i + r), c);
                Plaintext p;
                my_dec.decrypt(c, p);
                got3[i] = encoder.decode(p);
            });
        for (auto &t : th) t.join();
        for (int i = 0; i < T; ++i) CHECK(got3[i] == 1.5 * i + 7, "shared encryptor, thread %d: %g", i, got3[i]);
        CHECK(encryptor.next_index() == before + 8ull * T, "shared encryptor: %llu indices used, want %d", (unsigned long long)(encryptor.next_index() - before), 8 * T);
    }

    // ---- 4. a flush that throws ------------------------------------------------------------------------------------------
#ifdef FHE_FACADE_TEST_HOOKS
    if (!std::getenv("FHE_FACADE_EAGER")) {
        Evaluator evaluator(context);
        Ciphertext x, y;
        encryptor.encrypt(encoder.encode(3.0), x);
        encryptor.encrypt(encoder.encode(4.0), y);
        Ciphertext a(x), b(y);
        evaluator.add(a, y);                                           // level 1
        evaluator.negate(b);                                           // level 1, another group
        Ciphertext c(a);
        evaluator.multiply_plain(c, encoder.encode(0.25));             // level 2
        detail::fail_groups_after() = 1;                               // the first group runs, the second one throws
        bool threw = false;
        try { evaluator.flush(); } catch (const std::runtime_error &) { threw = true; }
        detail::fail_groups_after() = -1;
        CHECK(threw, "the injected failure must surface from flush()");
        int ok = 0, dead = 0;
        for (Ciphertext *ct : {&a, &b, &c}) {
            try { Plaintext p; decryptor.decrypt(*ct, p); ++ok; } catch (const std::runtime_error &) { ++dead; }
        }
        CHECK(ok == 1 && dead == 2, "one value computed before the failure, two marked failed: ok %d dead %d", ok, dead);
        threw = false;
        try { evaluator.add(c, x); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw, "an Evaluator call on a value whose flush failed must throw at once");
        threw = false;
        try { std::stringstream ss; c.save(ss); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw, "saving a value whose flush failed must throw");
        Ciphertext d(x);                                               // the context stays usable
        evaluator.add(d, y);
        Plaintext p;
        decryptor.decrypt(d, p);
        CHECK(encoder.decode(p) == 7.0, "the context works after a failed flush: %g", encoder.decode(p));
    }
#endif

    // ---- 5. a pending value that outlives its context --------------------------------------------------------------------
    if (!std::getenv("FHE_FACADE_EAGER")) {
        Ciphertext orphan;
        {
            SEALContext short_lived(make_params(n));
            KeyGenerator kg(short_lived);
            Encryptor e2(short_lived, kg.public_key());
            Evaluator ev2(short_lived);
            FractionalEncoder enc2(short_lived.plain_modulus(), short_lived.poly_modulus(), 100, 100, 2);
            Ciphertext x;
            e2.encrypt(enc2.encode(1.0), x);
            orphan = x;
            ev2.negate(orphan);                                        // pending when the context goes away
        }
        bool threw = false;
        try { std::stringstream ss; orphan.save(ss); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw, "a pending value whose context is gone must throw, not flush through a dangling pointer");
        Evaluator evaluator(context);
        threw = false;
        try { evaluator.negate(orphan); } catch (const std::exception &) { threw = true; }
        CHECK(threw, "a pending value of another (dead) context is not an operand");
    }

    std::printf(failures.load() ? "FACADE THREADS TEST FAILED (%d)\n" : "FACADE THREADS TEST OK\n", failures.load());
    return failures.load() ? 1 : 0;
}
