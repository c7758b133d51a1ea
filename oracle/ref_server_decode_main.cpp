// ref_server_decode_main.cpp -- TEST INFRASTRUCTURE (built into oracle/_ref/, never committed as binary).
//
// The REFERENCE's server_decode driver (/root/reference/homo/server_decode.cpp, its whole main(): parameter file,
// key loading, the per-channel loop :120-137 with `index += count` and the channel accumulation, the interleaved
// save :139-143) compiled UNCHANGED against this repository's SEAL-shaped facade -- the file is #included from where
// it lies (oracle/Makefile target `ref`), nothing of it is copied.
//
// One redirection: the reference's main passes its debugging Decryptor to approximated_step and thereby selects the
// DECRYPTING overload (homo/fhe_decode.h:244-282), which needs the secret key on the server.  The object-like macro
// below sends that call to the HOMOMORPHIC overload of the same header (:202-242) -- the path BASELINE.json's
// north_star names -- dropping the Decryptor argument.  The header is included first, so its own definitions are not
// touched by the macro (the include guard makes server_decode.cpp's own #include a no-op).
//
// homomorphic_cos has no return statement (:200, undefined behaviour): as in
// ref_decode_circuit_main.cpp this file is built at -O0 and the stack below the call is scrubbed first, so the
// never-constructed temporary the caller destroys reads as an empty Ciphertext.
#include "seal/seal.h"
#include "fhe_image.h"
#include "fhe_decode.h"   // the reference's header, unchanged

static void __attribute__((noinline)) scrub_stack() {
    volatile char z[1 << 20];
    for (size_t i = 0; i < sizeof z; i++) z[i] = 0;
}
static void step_homomorphic(Ciphertext &amplitude, Ciphertext &index, Ciphertext &count, int order, int degree, double delta, int width, int height,
                             std::vector<Ciphertext> &run, Evaluator &evaluator, FractionalEncoder &encoder, Encryptor &encryptor, Decryptor &) {
    scrub_stack();
    approximated_step(amplitude, index, count, order, degree, delta, width, height, run, evaluator, encoder, encryptor);      // homo/fhe_decode.h:202
}
#define approximated_step step_homomorphic
#include "server_decode.cpp"   // the reference's main, unchanged
