// server_decode_hip.cpp -- the batched server_decode as a C++ host (no Python, no torch): the driver loop of
// homo/server_decode.cpp:113-148 over seal/hip_circuits.h (the C ABI of include/fhe_circuits.h and include/fhe_hip.h).
// The reference, per colour channel: encrypts an index and width * height accumulators (encode(0), :121,126), reads the channel's
// run-length pairs one (elem, count) at a time (:131-132), calls approximated_step per run (:133; inside it homomorphic_sin and
// homomorphic_cos encrypt an encode(0) per (position, harmonic), homo/fhe_decode.h:54,134), accumulates (:134-136), advances the
// index (:137); then saves position-major with the three channels interleaved (:139-143).  Here, per channel:
//   ONE fhe_encrypt_batch for every Enc(0) of the channel in the reference's call order (seal::hip::DeviceEncryptor),
//   ONE load of the channel's 2 * pairs records (seal::hip::CiphertextBatch::load),
//   ONE fhe_decode_channel (seal::hip::Circuits::decode_channel),
// and the same output stream (the homomorphic overload of approximated_step; the reference's main hands its debugging Decryptor
// to the decrypting overload, which needs the secret key on the server -- DESIGN.md section 8).  Byte-identical to
// fully-homomorphic-image-processing_amd/server.py server_decode given the same sampler key (tests/test_gpu_server.py).
//
// usage: server_decode_hip <in.ct> <out.ct> <public key file> <width> <height> <pairs R> <pairs G> <pairs B>
//                          [order=64] [degree=12] [delta=0.5] [n=8192] [plain_modulus=16384] [sampler key: 64 hex digits]
//   The sampler key is for reproducible tests only: without it the key comes from getrandom() (a (key, index) pair must never repeat).
//   FHE_SEAL23_MODULI=1 selects SEAL 2.3.1's coefficient moduli (the presets SEAL23_* / P8192 of the Python harness).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <fstream>

#include <unistd.h>

#include "seal/hip_circuits.h"

using namespace seal;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    if (argc < 9) {
        std::fprintf(stderr, "usage: %s in.ct out.ct pubkey width height pairsR pairsG pairsB [order=64] [degree=12] [delta=0.5] [n=8192] [plain_modulus=16384] [key hex | -] [evaluation keys]\n", argv[0]);
        return 2;
    }
    const char *in_path = argv[1], *out_path = argv[2], *pk_path = argv[3];
    const uint32_t width = (uint32_t)std::atoi(argv[4]), height = (uint32_t)std::atoi(argv[5]);
    const long pairs[3] = {std::atol(argv[6]), std::atol(argv[7]), std::atol(argv[8])};
    const int order = argc > 9 ? std::atoi(argv[9]) : 64, degree = argc > 10 ? std::atoi(argv[10]) : 12;
    const double delta = argc > 11 ? std::atof(argv[11]) : 0.5;
    const int n = argc > 12 ? std::atoi(argv[12]) : 8192;
    const uint64_t t = argc > 13 ? std::strtoull(argv[13], nullptr, 10) : 16384;
    const size_t npos = (size_t)width * height;
    if (!npos || pairs[0] < 0 || pairs[1] < 0 || pairs[2] < 0 || degree < 0) return 2;
    uint8_t key[32];
    const bool have_key = argc > 14 && std::strlen(argv[14]) == 64;
    for (int i = 0; have_key && i < 32; ++i) { unsigned v = 0; std::sscanf(argv[14] + 2 * i, "%2x", &v); key[i] = (uint8_t)v; }
    bool out_open = false;
    try {
        EncryptionParameters params;
        char poly_mod[32];
        std::snprintf(poly_mod, sizeof poly_mod, "1x^%i + 1", n);
        params.set_poly_modulus(poly_mod);
        params.set_coeff_modulus(coeff_modulus_128(n));
        params.set_plain_modulus(t);
        SEALContext context(params);
        PublicKey pk;
        {
            std::ifstream kf(pk_path, std::ios::binary);
            if (!kf) throw std::invalid_argument("cannot open the public key file");
            pk.load(kf);
        }
        // argv[15]: an evaluation-key file (seal::EvaluationKeys::save) -> the RELINEARISED mode (relinearize after every product: records of two
        // polynomials instead of 22; the decode circuits take that placement only, include/fhe_circuits.h)
        EvaluationKeys evk;
        const char *evk_path = argc > 15 && std::strcmp(argv[15], "-") != 0 ? argv[15] : nullptr;
        if (evk_path) {
            std::ifstream ef(evk_path, std::ios::binary);
            if (!ef) throw std::invalid_argument("cannot open the evaluation key file");
            evk.load(ef);
        }
        std::unique_ptr<hip::Circuits> circ_p(evk_path ? new hip::Circuits(context, evk, 100, 100) : new hip::Circuits(context, 100, 100));   // FractionalEncoder(t, poly, 100, 100, 2): homo/server_decode.cpp
        hip::Circuits &circ = *circ_p;
        hip::DeviceEncryptor enc(context, pk, 100, 100, have_key ? key : nullptr, 0);
        std::ifstream in(in_path, std::ios::binary);
        if (!in) throw std::invalid_argument("cannot open the input stream");
        const double t0 = now();
        std::vector<uint64_t> host[3];
        uint32_t so[3];
        size_t encryptions = 0;
        for (int ch = 0; ch < 3; ++ch) {
            const size_t p = (size_t)pairs[ch], per_run = npos * (size_t)degree * 2;
            hip::CiphertextBatch z = enc.encrypt_zeros(1 + npos + p * per_run);  // the reference's order: index (:121), accumulators (:126), then per run, position, harmonic: sin, cos
            encryptions += z.count();
            Ciphertext index = z.get(0);
            hip::CiphertextBatch acc0(context, npos, 2), zeros(context, p * per_run, 2), runs;
            detail::check(fhe_copy(acc0.ptr(), z.at(1), npos * z.ct_words() * 8, nullptr), "copy");
            if (p != 0 && per_run != 0) detail::check(fhe_copy(zeros.ptr(), z.at(1 + npos), p * per_run * z.ct_words() * 8, nullptr), "copy");
            runs.load(context, in, 2 * p, 2);                                   // (elem, count) per run (:131-132); validates the residues
            hip::CiphertextBatch out = circ.decode_channel(runs, index, acc0, zeros, order, degree, delta, width, height);
            so[ch] = out.size();
            host[ch] = out.to_host();
        }
        // :139-143: position-major, the three channels interleaved
        std::ofstream os(out_path, std::ios::binary | std::ios::trunc);
        if (!os) throw std::invalid_argument("cannot open the output stream");
        out_open = true;
        const detail::CtxState &st = *context.state();
        const char magic[8] = {'F', 'H', 'E', 'H', 'I', 'P', '1', 0};
        for (size_t i = 0; i < npos; ++i)
            for (int ch = 0; ch < 3; ++ch) {
                const uint32_t hdr[4] = {so[ch], st.k, st.n, 0};
                const size_t words = (size_t)so[ch] * st.k * st.n;
                os.write(magic, 8);
                os.write((const char *)hdr, sizeof hdr);
                os.write((const char *)(host[ch].data() + i * words), (std::streamsize)(words * 8));
            }
        os.flush();
        if (!os) throw std::runtime_error("short write on the output stream");
        const double dt = now() - t0;
        const long total = pairs[0] + pairs[1] + pairs[2];
        std::printf("{\"workload\": \"server_decode stream, C++ host over seal/hip_circuits.h (n=%u, k=%u), %ux%u, runs per channel [%ld, %ld, %ld], order %d, degree %d\", "
                    "\"encryptions\": %zu, \"runs\": %ld, \"seconds\": %.4f, \"ms_per_run\": %.2f}\n",
                    st.n, st.k, width, height, pairs[0], pairs[1], pairs[2], order, degree, encryptions, total, dt, total ? dt * 1e3 / total : 0.0);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "server_decode_hip: %s\n", e.what());
        // a failed job (a residue that is not reduced, a foreign or truncated record) must not leave a complete-looking output behind
        if (out_open && truncate(out_path, 0) != 0) std::fprintf(stderr, "server_decode_hip: could not truncate %s\n", out_path);
        return 1;
    }
    return 0;
}
