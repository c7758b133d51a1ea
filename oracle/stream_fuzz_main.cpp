// TEST INFRASTRUCTURE ONLY.  stream_fuzz_main.cpp -- the untrusted-input surface of the host code under AddressSanitizer / UBSan:
//   * seal::Ciphertext::load / PublicKey::load / SecretKey::load (seal/seal.h load_host: the replacement of the loads at
//     homo/server_jpeg.cpp:117-123, homo/fhe_resize.h:335-341, homo/server_decode.cpp:131-143) on arbitrary bytes;
//   * seal::EvaluationKeys::load (the key file of seal/server_resize_hip.cpp / server_decode_hip.cpp) on arbitrary bytes, and
//     Evaluator::relinearize with whatever loaded (the consumer-side check EvaluationKeys::require_for);
//   * fhe_io_open + fhe_io_transfer and fhe_io_read_records (csrc/stream_io.hip, include/fhe_stream.h -- the PRODUCT's own
//     translation unit, compiled here as plain C++) on arbitrary files with arbitrary record shapes and ranges.
// Input: a bundle file written by tests/test_sanitizers.py (hypothesis generates the cases):
//   u32 n_cases, then per case: u32 polys, k, n, first_record, count, threads, u32 n_bytes, the bytes.
// Output: one line per case "case <i> load=<0|1> pk=<0|1> sk=<0|1> evk=<0|1|2> transfer=<rc> read=<rc>" (evk 2: loaded AND relinearised a size-3 ciphertext); every rejection must be an
// exception / error code, never a crash or a sanitizer report.  The device behind the facade is the CPU oracle's C ABI.
#include <fcntl.h>
#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "seal/seal.h"
#include "fhe_stream.h"

// the one symbol csrc/stream_io.hip takes from the rest of the library
static char g_io_err[256];
int fhe_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_io_err, sizeof g_io_err, fmt, ap);
    va_end(ap);
    return code;
}

static bool read_u32(std::istream &is, uint32_t &v) { return (bool)is.read((char *)&v, 4); }

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <bundle> <scratch file>\n", argv[0]); return 2; }
    using namespace seal;
    // a context so that load() has moduli to check residues against (n = 1024 / 2048 share one 54-bit prime)
    EncryptionParameters params;
    params.set_poly_modulus("1x^1024 + 1");
    params.set_coeff_modulus(coeff_modulus_128(1024));
    params.set_plain_modulus(1 << 14);
    SEALContext context(params);
    // a second context of another degree: streams for IT load here too, and must be refused by the consumers of the first
    EncryptionParameters params2;
    params2.set_poly_modulus("1x^2048 + 1");
    params2.set_coeff_modulus(coeff_modulus_128(2048));
    params2.set_plain_modulus(1 << 14);
    SEALContext context2(params2);
    // a size-3 ciphertext for the relinearize attempts
    KeyGenerator keygen(context);
    Encryptor encryptor(context, keygen.public_key());
    Evaluator evaluator(context);
    Ciphertext product3;
    encryptor.encrypt(Plaintext(std::vector<uint64_t>{3}), product3);
    evaluator.square(product3);
    std::ifstream in(argv[1], std::ios::binary);
    uint32_t n_cases = 0;
    if (!read_u32(in, n_cases)) return 2;
    for (uint32_t c = 0; c < n_cases; ++c) {
        uint32_t polys, k, n, first, count, threads, len;
        if (!(read_u32(in, polys) && read_u32(in, k) && read_u32(in, n) && read_u32(in, first) && read_u32(in, count) && read_u32(in, threads) && read_u32(in, len))) return 2;
        std::string bytes(len, '\0');
        if (len && !in.read(&bytes[0], len)) return 2;
        int ok[4] = {0, 0, 0, 0};
        for (int which = 0; which < 4; ++which) {
            std::stringstream ss(bytes);
            try {
                if (which == 0) { Ciphertext ct; ct.load(ss); if (ct.size() >= 1) { std::stringstream out; ct.save(out); } }
                else if (which == 1) { PublicKey pk; pk.load(ss); }
                else if (which == 2) { SecretKey sk; sk.load(ss); }
                else {
                    EvaluationKeys evk;
                    evk.load(ss);
                    ok[3] = 1;
                    // a stream that loads may still not be usable on THIS context (other degree, digit count that does not fit dbc, too few
                    // powers): relinearize must say so before the library reads behind the buffer
                    Ciphertext c3 = product3;
                    evaluator.relinearize(c3, evk);
                    if (c3.size() == 2) ok[3] = 2;
                    continue;
                }
                ok[which] = 1;
            } catch (const std::invalid_argument &) {
            } catch (const std::runtime_error &) {
            }
        }
        { std::ofstream f(argv[2], std::ios::binary | std::ios::trunc); f.write(bytes.data(), (std::streamsize)bytes.size()); }
        int rc_t = -99, rc_r = -99;
        // bound the destination: the harness, like any caller, sizes its buffer from the arguments it passes
        const uint64_t words = (uint64_t)polys * k * n * count;
        if (words <= (1u << 22)) {
            std::vector<uint64_t> dst((size_t)words + 1, 0);
            fhe_io_file *f = nullptr;
            rc_t = fhe_io_open(argv[2], 0, 0, &f);
            if (rc_t == 0) { rc_t = fhe_io_transfer(f, first, count, polys, k, n, dst.data(), threads); fhe_io_close(f); }
            const int fd = open(argv[2], O_RDONLY);
            if (fd >= 0) { rc_r = fhe_io_read_records(fd, first, count, polys, k, n, dst.data(), threads); close(fd); }
        }
        std::printf("case %u load=%d pk=%d sk=%d evk=%d transfer=%d read=%d\n", c, ok[0], ok[1], ok[2], ok[3], rc_t, rc_r);
    }
    std::printf("FUZZ BUNDLE DONE\n");
    return 0;
}
