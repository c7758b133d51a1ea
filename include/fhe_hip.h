/*
 * fhe_hip.h -- C ABI of libfhe_hip.so: BFV ciphertext arithmetic on MI355X (gfx950).
 *
 * This is the drop-in boundary for the ONE hot path of wfus/Fully-Homomorphic-Image-Processing:
 * the arithmetic that homo/server_jpeg.cpp, server_resize.cpp and server_decode.cpp reach through
 * Microsoft SEAL v2.3's C++ class API (seal::Evaluator et al.).  The reference has no FFI of its
 * own; its seam is `#include "seal/seal.h"` (homo/fhe_image.h:13).  The SEAL-shaped C++ facade in
 * fully-homomorphic-image-processing_amd/seal/seal.h forwards every Evaluator call to the entry
 * points below, so homo/fhe_image.h, fhe_resize.h and fhe_decode.h compile unchanged against it.
 * INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types.  `stream` is a hipStream_t passed as void*
 *    (NULL = the default stream).  Device pointers are ordinary HIP device addresses (hipMalloc,
 *    torch.empty(device="cuda").data_ptr(), ...).  Calls are asynchronous on `stream`.
 *  - every function returns FHE_OK (0) or a negative error code; fhe_last_error() gives the text
 *    (thread-local).  Nothing throws.  There is NO CPU fallback: without a HIP device every
 *    compute entry point fails with FHE_ERR_HIP.
 *  - ciphertext layout (SEAL-logical): u64 [ciphertext][poly j][prime i][coeff c], every residue
 *    fully reduced to [0, q_i).  "n_polys" counts RNS polynomials, i.e. size * number of cts.
 *  - threading: a context is immutable for its users after fhe_ctx_create and may be shared by any number of
 *    host threads issuing calls on their own streams.  Two pieces of state are created later, both behind
 *    their own synchronisation and never freed or moved while the context lives: the ct x ct tables
 *    (auxiliary base, its twiddles, base-conversion constants), built under std::call_once by the FIRST
 *    entry point that multiplies ciphertexts (fhe_multiply*, fhe_square, fhe_relinearize,
 *    fhe_circuits_create, fhe_arith_path) -- threads racing that first call are safe, all see the same
 *    tables or the same error --, and the rgb_to_ycc constant cache (mutex-guarded).  Two restrictions:
 *    (1) the pipelined DCT mode (FHE_DCT_PIPELINE=1) uses one second stream owned by the context, so at most
 *    one fhe_dct8x8_quant call per context may be in flight in that mode; (2) fhe_ctx_destroy must not
 *    race with any other call on the same context.  Plans and scratch buffers belong to their caller.
 *  - devices: a context belongs to the device given to fhe_ctx_create; launches, fhe_dev_alloc and
 *    fhe_stream_create act on the CALLING THREAD's current HIP device.  fhe_ctx_create makes its device
 *    current for the creating thread; a host that drives several GPUs from one process uses one thread per
 *    device (or calls fhe_ctx_bind_thread before switching): seal/multi_gpu_dct.cpp is the worked example.
 *  - experiment switches (FHE_DCT_*, FHE_NTT_*, FHE_BEHZ_* environment variables; csrc/internal.h lists
 *    them) are read ONCE, by fhe_ctx_create, and are fixed for the life of that context: no launch path
 *    reads the environment.  The defaults are the measured-best kernels; every alternative gives the same
 *    bits (the parity tests create a second context with the variable set).  Creating a context costs a few
 *    milliseconds and a few MB of device tables (twiddles of the coefficient base; one stream and four
 *    events more in contexts created with FHE_DCT_PIPELINE=1); the ct x ct tables (twiddles for the k+1 auxiliary primes, base-conversion constants) are added
 *    by the first call that multiplies ciphertexts, so a DCT-only server neither pays for them nor can fail on
 *    the auxiliary-prime search (FHE_BEHZ_EAGER=1 builds them in fhe_ctx_create as before round 4).
 *  - "NTT form" buffers use a library-internal slot order; they are only meaningful to this
 *    library (produced by fhe_plain_prepare / fhe_ntt_forward, consumed by the matching calls).
 */
#ifndef FHE_HIP_H
#define FHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHE_OK 0
#define FHE_ERR_PARAM (-1)  /* invalid argument / unsupported parameter set   */
#define FHE_ERR_HIP (-2)    /* HIP runtime error (no device, launch failure)  */
#define FHE_ERR_NOMEM (-3)

#define FHE_MAX_K 8
#define FHE_MAX_POLYS 64      /* polynomials per ciphertext (the deepest reference circuit reaches 22, homo/fhe_decode.h:239) */

typedef struct fhe_ctx fhe_ctx;
typedef struct fhe_dct_plan fhe_dct_plan;
typedef void *fhe_stream;

const char *fhe_last_error(void);
/* ABI version of this header; bumped on any signature change and whenever entry points are added or a contract changes.
 *   1: rounds 1-3.
 *   2: + fhe_gather, fhe_host_alloc / fhe_host_free, fhe_stream_create / destroy, fhe_ctx_bind_thread / fhe_ctx_device,
 *      fhe_count_unreduced, fhe_add_sizes, the *_range / *_rows circuit shards, fhe_relinearize_to, the relinearised mode of the circuits
 *      (fhe_circuits_create_relin, fhe_circuits_out_size); fhe_ctx_create no longer builds or validates the ct x ct tables
 *      (auxiliary-prime failures surface at the first multiply or at fhe_circuits_create), fhe_arith_path builds them as
 *      a side effect, and the context's second stream exists only with FHE_DCT_PIPELINE=1.
 *   3: + fhe_encrypt_batch / fhe_encrypt_scratch_bytes / fhe_encrypt_draws / fhe_noise_cdt, fhe_frac_encode_batch,
 *      fhe_decrypt_batch / fhe_decrypt_scratch_bytes / fhe_ctx_modulus_bits.
 *   4: + fhe_relinearize_poly / fhe_relinearize_n (key switches for s^3 ..: a size-4 Cubic result goes to size 2 in one
 *      evaluator.relinearize, as SEAL's does), fhe_circuits_create_relin_at (include/fhe_circuits.h: where the relinearised mode
 *      relinearises); fhe_relinearize_to rejects partially overlapping input / output ranges.
 * A host compiled against this header compares fhe_abi_version() with FHE_ABI_VERSION before anything else (the Python
 * binding and seal/seal.h do). */
#define FHE_ABI_VERSION 4
uint32_t fhe_abi_version(void);

/* ---- context: replaces seal::EncryptionParameters + seal::SEALContext -------------------------
 * (homo/server_jpeg.cpp:74-80: set_poly_modulus("1x^n + 1"), set_coeff_modulus(coeff_modulus_128(n)),
 *  set_plain_modulus(t)).  q_i must be distinct primes < 2^61 with q_i = 1 (mod 2n); n a power of
 * two in [1024, 16384].  Tables (twiddles, Shoup companions, BEHZ base-conversion constants) are
 * built on the host and uploaded to `device`. */
int fhe_ctx_create(uint32_t n, const uint64_t *q, uint32_t k, uint64_t t, int device, fhe_ctx **out);
int fhe_ctx_destroy(fhe_ctx *ctx);
/* 1 once the lazily built ct x ct tables exist (diagnostic: a context that only ran linear circuits reports 0) */
int fhe_ctx_has_ctct_tables(const fhe_ctx *ctx);
/* the device the context was created on; fhe_ctx_bind_thread makes it the calling thread's current device (hipSetDevice) */
int fhe_ctx_device(const fhe_ctx *ctx);
int fhe_ctx_bind_thread(const fhe_ctx *ctx);
/* a non-blocking stream on the calling thread's current device, for hosts without HIP headers (pass it as `stream`) */
int fhe_stream_create(fhe_stream *out);
int fhe_stream_destroy(fhe_stream stream);
uint32_t fhe_ctx_n(const fhe_ctx *ctx);
uint32_t fhe_ctx_k(const fhe_ctx *ctx);
uint64_t fhe_ctx_t(const fhe_ctx *ctx);
uint64_t fhe_ctx_q(const fhe_ctx *ctx, uint32_t i);
/* SEAL 2.3 `coeff_modulus_128(n)` replacement (homo/server_jpeg.cpp:78).  preset 0 = the
 * 3x36/37-bit set BASELINE.json names for n=4096 (and the matching sets for other n),
 * preset 1 = SEAL 2.3.1 defaults.  Returns the number of primes written (<= FHE_MAX_K) or <0. */
int fhe_default_coeff_modulus(uint32_t n, int preset, uint64_t *q_out);

/* ---- device memory helpers for hosts without their own allocator (the C++ facade) ------------ */
int fhe_dev_alloc(size_t bytes, void **dptr);
int fhe_dev_free(void *dptr);
int fhe_upload(void *dst_dev, const void *src_host, size_t bytes, fhe_stream stream);
int fhe_download(void *dst_host, const void *src_dev, size_t bytes, fhe_stream stream);
int fhe_copy(void *dst_dev, const void *src_dev, size_t bytes, fhe_stream stream);
int fhe_stream_sync(fhe_stream stream);
/* page-locked host memory for staging buffers of hosts without their own allocator (the facade's Ciphertext::load / save go
 * through one per thread: a transfer from or to pageable memory is staged a second time inside the runtime) */
int fhe_host_alloc(size_t bytes, void **hptr);
int fhe_host_free(void *hptr);
/* `count` scattered device buffers (addresses in HOST memory, consumed before the call returns) of words_each u64 ->
 * dst[i * dst_stride_words ...], one launch per 256 sources, no staging copy.  The SEAL facade's lazy mode uses it to
 * run the reference's one-ciphertext-at-a-time Evaluator calls (homo/fhe_image.h:206-284) as batched launches.  16-byte
 * units: even word counts, 16-byte aligned buffers. */
int fhe_gather(const uint64_t *const *src_host, uint64_t count, uint64_t words_each, uint64_t *dst,
               uint64_t dst_stride_words, fhe_stream stream);

/* ---- seal::FractionalEncoder(t, poly_modulus, int_coeffs, frac_coeffs, base=2) ------------------
 * (ctor homo/server_jpeg.cpp:100; encode() call sites homo/fhe_image.h:221-236,259,301,317-319).
 * Host-side, no device work.  plain_out must hold n coefficients; returns the significant
 * coefficient count (0 for the zero plaintext) or <0. */
int fhe_frac_encode(uint32_t n, uint64_t t, double value, int int_coeffs, int frac_coeffs,
                    uint64_t *plain_out);
double fhe_frac_decode(uint32_t n, uint64_t t, const uint64_t *plain, int int_coeffs, int frac_coeffs);

/* ---- seal::Evaluator::add / sub / negate (homo/fhe_image.h:207-220, fhe_resize.h:151-169,
 * fhe_decode.h:219).  Element-wise over n_polys RNS polynomials; out may alias a or b.
 * Unequal ciphertext sizes are handled by the caller (facade) by running the common prefix
 * through add/sub and the tail through copy/negate, as SEAL does. */
int fhe_add(const fhe_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n_polys,
            fhe_stream stream);
int fhe_sub(const fhe_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n_polys,
            fhe_stream stream);
int fhe_negate(const fhe_ctx *ctx, const uint64_t *a, uint64_t *out, uint64_t n_polys, fhe_stream stream);
/* The same add / sub for `count` pairs of ciphertexts of UNEQUAL sizes, batched (a: [count][size_a][k][n], b: [count][size_b][k][n],
 * out: [count][max(size_a, size_b)][k][n]): the destination grows and the polynomials the shorter operand lacks count as zero
 * (sub: the tail of b is negated), as seal::Evaluator does at homo/fhe_resize.h:181-184 and homo/fhe_decode.h:114-118,237 where a
 * size-3 product meets a size-2 or a size-5 ciphertext.  out may alias the longer operand.  subtract: 0 = a + b, 1 = a - b. */
int fhe_add_sizes(const fhe_ctx *ctx, const uint64_t *a, uint32_t size_a, const uint64_t *b, uint32_t size_b, uint64_t *out,
                  uint64_t count, int subtract, fhe_stream stream);

/* ---- plaintext operands ---------------------------------------------------------------------------
 * fhe_plain_prepare: centred lift of a seal::Plaintext (coefficients in [0,t), host memory) to the
 * q-base followed by a forward NTT; writes 2*k*n u64 to d_plain_ntt (value + Shoup companion per
 * slot).  Cache the result: the reference re-encodes and re-transforms the same 13 constants on
 * every call (homo/fhe_image.h:221-236). */
size_t fhe_plain_ntt_words(const fhe_ctx *ctx);
int fhe_plain_prepare(const fhe_ctx *ctx, const uint64_t *plain_host, uint32_t plain_len,
                      uint64_t *d_plain_ntt, fhe_stream stream);
/* product of two prepared plaintexts as ring elements (used to fold encode(0.125)*encode(1/q)) */
int fhe_plain_ntt_mul(const fhe_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out,
                      fhe_stream stream);

/* seal::Evaluator::multiply_plain (homo/fhe_image.h:221 etc.): every polynomial of the n_polys
 * inputs is multiplied in R_q by the prepared plaintext (fused NTT -> dyadic -> inverse NTT).
 * n_polys must be a multiple of k (whole RNS polynomials).  out may alias in. */
int fhe_multiply_plain(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n_polys,
                       const uint64_t *d_plain_ntt, fhe_stream stream);
/* multiply_plain for a plaintext with at most FHE_SPARSE_MAX_TERMS non-zero coefficients (host
 * memory, coefficients in [0,t)): the ring product is formed directly as a sum of signed rotations
 * in coefficient form -- no transform.  The integer and power-of-two constants of Cubic
 * (encode(3) = x+1, encode(5) = x^2+1, encode(0.5) = -x^(n-1); homo/fhe_resize.h:150-163,186) are of
 * this kind.  Same result as fhe_multiply_plain (exact ring arithmetic, canonical residues).
 * out may alias in.  n <= 8192.  Returns FHE_ERR_PARAM if the plaintext has more terms or is zero. */
#define FHE_SPARSE_MAX_TERMS 8
int fhe_multiply_plain_sparse(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n_polys,
                              const uint64_t *plain_host, uint32_t plain_len, fhe_stream stream);

/* Cubic (homo/fhe_resize.h:143-189) outside its four ciphertext products, for FractionalEncoder base 2
 * where encode(3) = x+1, encode(2) = x, encode(5) = x^2+1, encode(4) = x^2, encode(0.5) = -x^(n-1):
 *   fhe_cubic_coeffs:  a = 3B - A - 3C + D,  b = 2A - 5B + 4C - D,  c = C - A      (:150-172; one pass
 *                      over A..D instead of six multiply_plain and eight add/sub calls)
 *   fhe_cubic_combine: out = 0.5 (a + b + c) + B                                   (:181-188)
 * exactly as the Evaluator calls compose (same ring elements, canonical residues).  Operands are
 * batches of `count` ciphertexts; A..D, a, b, c have `size` polynomials each; in combine a, b, c have
 * `size_abc` polynomials, B has `size_b` <= size_abc, out has size_abc.  Outputs must not alias inputs. */
int fhe_cubic_coeffs(const fhe_ctx *ctx, const uint64_t *A, const uint64_t *B, const uint64_t *C, const uint64_t *D,
                     uint64_t *a, uint64_t *b, uint64_t *c, uint32_t size, uint64_t count, fhe_stream stream);
int fhe_cubic_combine(const fhe_ctx *ctx, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint32_t size_abc,
                      const uint64_t *B, uint32_t size_b, uint64_t *out, uint64_t count, fhe_stream stream);

/* seal::Evaluator::add_plain / sub_plain (homo/fhe_image.h:317 sub_plain(128.0); fhe_resize.h:196;
 * fhe_decode.h:57,113,218,220,229): c_0 += sign * Delta * m' for `count` ciphertexts whose first
 * polynomial starts every ct_stride_words u64. sign = +1 / -1. */
int fhe_add_plain(const fhe_ctx *ctx, uint64_t *ct, uint64_t ct_stride_words, uint64_t count,
                  const uint64_t *plain_host, uint32_t plain_len, int sign, fhe_stream stream);

/* ---- negacyclic NTT over the q-base (north_star primitive; SEAL-internal in the reference) -----
 * in: [n_polys][k][n] coefficient form; out: NTT form (internal slot order), values in [0,q_i). */
int fhe_ntt_forward(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n_polys,
                    fhe_stream stream);
int fhe_ntt_inverse(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n_polys,
                    fhe_stream stream);
/* coefficient-wise (dyadic) product of two NTT-form operands with Barrett reduction */
int fhe_dyadic_multiply(const fhe_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out,
                        uint64_t n_polys, fhe_stream stream);

/* ---- seal::Evaluator::multiply / square (homo/fhe_resize.h:174-179,197-198; fhe_decode.h:67-97,
 * 235,239): full-RNS BEHZ product of `count` pairs of ciphertexts of sizes size_a, size_b
 * (each operand contiguous, [count][size][k][n]); out has size_a+size_b-1 polys per pair.
 * scratch: device memory of at least fhe_multiply_scratch_bytes(). */
size_t fhe_multiply_scratch_bytes(const fhe_ctx *ctx, uint32_t size_a, uint32_t size_b, uint64_t count);
int fhe_multiply(const fhe_ctx *ctx, const uint64_t *a, uint32_t size_a, const uint64_t *b,
                 uint32_t size_b, uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes,
                 fhe_stream stream);
int fhe_square(const fhe_ctx *ctx, const uint64_t *a, uint32_t size_a, uint64_t *out, uint64_t count,
               void *scratch, size_t scratch_bytes, fhe_stream stream);
/* An operand of fhe_multiply extended to the auxiliary base and transformed ONCE, for circuits that
 * multiply several ciphertexts by the same one (Cubic: t and t^2 enter every row of a pixel,
 * homo/fhe_resize.h:176-179,296-303).  fhe_multiply(a, b) == fhe_multiply_prepared(prepare(a), prepare(b))
 * bit for bit.  A prepared operand holds fhe_multiply_operand_words() u64 words in device memory and is
 * opaque.  Pass NULL for a_prepared (b_prepared) together with the plain ciphertext a (b) to prepare that
 * side inside the call. */
size_t fhe_multiply_operand_words(const fhe_ctx *ctx, uint32_t size, uint64_t count);
int fhe_multiply_prepare(const fhe_ctx *ctx, const uint64_t *a, uint32_t size, uint64_t count, uint64_t *prepared,
                         fhe_stream stream);
int fhe_multiply_prepared(const fhe_ctx *ctx, const uint64_t *a, const uint64_t *a_prepared, uint32_t size_a,
                          const uint64_t *b, const uint64_t *b_prepared, uint32_t size_b, uint64_t *out, uint64_t count,
                          void *scratch, size_t scratch_bytes, fhe_stream stream);
/* The same with a prepared operand batch SHARED between the pairs: b_prepared holds b_count entries and pair c
 * multiplies entry (b_first + c / b_div) % b_count -- ResizeImage's offsets (homo/fhe_resize.h:351,382): frac(u) depends
 * on the output column only and frac(v) on the output row only, so a batch of whole pixel rows in row-major order
 * multiplies xfract[c % width] (b_div = 1, b_count = width) and yfract[first_row + c / width] (b_div = width).
 * Bit-identical to fhe_multiply_prepared on the gathered operand; no copy of the prepared words is made. */
int fhe_multiply_prepared_shared(const fhe_ctx *ctx, const uint64_t *a, const uint64_t *a_prepared, uint32_t size_a,
                                 const uint64_t *b_prepared, uint32_t size_b, uint64_t b_count, uint64_t b_div,
                                 uint64_t b_first, uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes,
                                 fhe_stream stream);

/* ---- seal::Evaluator::relinearize (north_star API surface; the reference never calls it, only
 * tests/parameters.cpp:112 touches evaluation keys).  Key-switch inner product for `count`
 * size-3 ciphertexts -> size 2, in place on the first two polys.  evk (device):
 * [k][n_digits][2][k][n] in NTT form (internal order) as produced by fhe_evk_to_ntt. */
uint32_t fhe_evk_digits(const fhe_ctx *ctx, uint32_t dbc);
int fhe_relinearize(const fhe_ctx *ctx, uint64_t *ct3, uint64_t ct_stride_words, uint64_t count,
                    const uint64_t *d_evk_ntt, uint32_t dbc, void *scratch, size_t scratch_bytes,
                    fhe_stream stream);
size_t fhe_relinearize_scratch_bytes(const fhe_ctx *ctx, uint32_t dbc, uint64_t count);
/* The same with the result written elsewhere: out2[c * out_stride_words ...] = the relinearised size-2 ciphertext of
 * ct3[c * ct_stride_words ...] (a batch of products becomes a compact [count][2][k][n] batch without a copy of its own;
 * the relinearised mode of the circuits, include/fhe_circuits.h, is built on it).  out2 == ct3 with equal strides is
 * fhe_relinearize.  Same scratch.  Aliasing rule: the output range [out2, out2 + (count-1) out_stride + 2kn) either IS the
 * input (same pointer, same stride) or does not overlap [ct3, ct3 + (count-1) stride + 3kn) at all; any other overlap
 * returns FHE_ERR_PARAM (a compacting in-place relinearisation would let one ciphertext's output land on another's input). */
int fhe_relinearize_to(const fhe_ctx *ctx, const uint64_t *ct3, uint64_t ct_stride_words, uint64_t *out2,
                       uint64_t out_stride_words, uint64_t count, const uint64_t *d_evk_ntt, uint32_t dbc, void *scratch,
                       size_t scratch_bytes, fhe_stream stream);

/* One key-switch step (what SEAL 2.3's relinearize repeats until size 2): polynomial `src_poly` >= 2 -- the LAST one of a
 * ciphertext of src_poly + 1 polynomials -- is decomposed into digits and folded into c0 / c1 with the keys for
 * s^src_poly (d_evk_ntt: [k][n_digits][2][k][n], same form as above, made from s^src_poly instead of s^2).  Only c0' and c1'
 * are written (out2, same aliasing rule as fhe_relinearize_to); in place (out2 == ct, equal strides) polynomials
 * 2 .. src_poly - 1 simply stay where they are, so the ciphertext has become one polynomial shorter.  src_poly == 2 is
 * fhe_relinearize_to.  Same scratch. */
int fhe_relinearize_poly(const fhe_ctx *ctx, const uint64_t *ct, uint64_t ct_stride_words, uint32_t src_poly, uint64_t *out2,
                         uint64_t out_stride_words, uint64_t count, const uint64_t *d_evk_ntt, uint32_t dbc, void *scratch,
                         size_t scratch_bytes, fhe_stream stream);
/* evaluator.relinearize(ct, evk) for ciphertexts of `size` >= 3 polynomials down to 2 (SEAL 2.3: size - 2 steps, the top
 * polynomial first): d_evk_ntt holds the keys for s^2, s^3, .. s^(size-1) one after the other (fhe_evk_words(ctx, dbc) words
 * each; KeyGenerator::generate_evaluation_keys(dbc, size - 2, keys)).  The steps above the last one run IN PLACE: ct's first two
 * polynomials are overwritten with partial sums when size > 3 (ct is scratch after the call).  out2 as in fhe_relinearize_to. */
size_t fhe_evk_words(const fhe_ctx *ctx, uint32_t dbc);
/* scratch for fhe_relinearize_n: the digits of all size - 2 source polynomials at once.  With at least this much the size - 2 key
 * switches run in as few PASSES as the lazy sums of the pseudo-Mersenne kernels allow (k * digits * powers <= 20 terms per pass: one
 * pass for a size-4 ciphertext at k = 4, dbc 30, and for a size-6 one at dbc 60; two passes of two powers for size 6 at dbc 30) --
 * every step's source polynomial is the caller's own, so the result is the ciphertext plus the sum of the steps' terms, the same bits
 * in any order -- with one inverse transform pair per pass instead of one per step; with only fhe_relinearize_scratch_bytes the steps
 * run one after the other. */
size_t fhe_relinearize_n_scratch_bytes(const fhe_ctx *ctx, uint32_t size, uint32_t dbc, uint64_t count);
int fhe_relinearize_n(const fhe_ctx *ctx, uint64_t *ct, uint32_t size, uint64_t ct_stride_words, uint64_t *out2,
                      uint64_t out_stride_words, uint64_t count, const uint64_t *d_evk_ntt, uint32_t dbc, void *scratch,
                      size_t scratch_bytes, fhe_stream stream);

/* ---- fused block circuit: encrypted_dct (homo/fhe_image.h:196-288) followed by quantize_fhe
 * (homo/fhe_image.h:294-305) on n_blocks independent 8x8 blocks.  in/out: [n_blocks][64][2][k][n].
 * The 832 Evaluator calls per block are exact operations in R_q, so the kernels transform each
 * input polynomial once, evaluate the whole linear circuit per NTT slot, and transform back:
 * the ciphertexts are bit-identical to the op-at-a-time evaluation.
 * quant64 == NULL builds a plan for encrypted_dct alone. */
int fhe_dct_plan_create(const fhe_ctx *ctx, const double *quant64, int int_coeffs, int frac_coeffs,
                        fhe_stream stream, fhe_dct_plan **out);
int fhe_dct_plan_destroy(fhe_dct_plan *plan);
size_t fhe_dct8x8_scratch_bytes(const fhe_ctx *ctx, uint64_t n_blocks);
/* which kernels fhe_dct8x8_quant launches for this context (for labels in measurements): 1 = the fused exact-FP64
 * pair k_dct_rows + k_dct_cols (primes < 2^47, n <= 8192), 2 = the fused u64 pair k_dct_rows_u64 + k_dct_cols_u64
 * (primes <= 57 bits, n in 2048..8192), 0 = the general path k_ntt_fwd + k_dct_slots + k_ntt_inv. */
int fhe_dct_path(const fhe_ctx *ctx);
/* which arithmetic the u64 kernels of this context run on (for labels in measurements and for tests that must know which
 * kernels they exercised): bits 0-1 = class of the q-base, bits 2-3 = class of the auxiliary ct x ct base -- 0 Shoup / Harvey
 * kernels, 1 pseudo-Mersenne kernels for primes <= 55 bits, 2 pseudo-Mersenne kernels for primes <= 58 bits
 * (csrc/ntt_core.h) --, bit 4 = the ct x ct base conversions run as two-column sums (k_behz_to_bsk_pm,
 * k_behz_floor_back_pm).  0 when FHE_NTT_NOPM=1 was set at fhe_ctx_create or no prime of a base qualifies. */
int fhe_arith_path(const fhe_ctx *ctx);
int fhe_dct8x8_quant(const fhe_ctx *ctx, const fhe_dct_plan *plan, const uint64_t *in, uint64_t *out,
                     uint64_t n_blocks, void *scratch, size_t scratch_bytes, fhe_stream stream);

/* rgb_to_ycc_fhe (homo/fhe_image.h:310-325) on `count` pixels; r,g,b: [count][2][k][n], in place. */
int fhe_rgb_to_ycc(const fhe_ctx *ctx, uint64_t *r, uint64_t *g, uint64_t *b, uint64_t count,
                   int int_coeffs, int frac_coeffs, fhe_stream stream);
/* The same on the layout of the reference's ciphertext streams (homo/server_jpeg.cpp:115-124: per 8x8 block 64 R, 64 G,
 * 64 B ciphertexts): blocks [n_blocks][3][64][2][k][n], in place -> [n_blocks][Y, Cb, Cr][64][2][k][n], which read as
 * [3 n_blocks][64][2][k][n] is the input layout of fhe_dct8x8_quant and the order homo/server_jpeg.cpp:146-153 saves. */
int fhe_rgb_to_ycc_blocks(const fhe_ctx *ctx, uint64_t *blocks, uint64_t n_blocks, int int_coeffs, int frac_coeffs,
                          fhe_stream stream);

/* ---- server-side encryptions (round 5) ------------------------------------------------------------
 * The reference's servers ENCRYPT inside their loops: SampleLinear / SampleBicubic encrypt frac(x) and frac(y) for every output
 * pixel (homo/fhe_resize.h:230,234,262,266: `encryptor.encrypt(encoder.encode(x - floor(x)), xfract)`), homomorphic_sin / cos an
 * encode(0) per call (homo/fhe_decode.h:54,134), server_decode the index and the accumulators (homo/server_decode.cpp:121,126).
 * With the circuits batched these encryptions were what a server spent its time on (host sampling, three uploads and five
 * launches per ciphertext); the entry points below form a whole batch on the device:
 *     Enc(m) = (Delta m' + pk0 u + e1, pk1 u + e2)        u ternary, e1, e2 rounded normals (sigma 3.19, redrawn beyond 19)
 * (textbook BFV, SURVEY.md App. A.7 -- what seal::Encryptor::encrypt computes; like every ciphertext of this build the bits are
 * the library's own: SEAL's sampler is not available here).
 *
 * Randomness: the ChaCha20 stream cipher (D. J. Bernstein's original layout: 256-bit key, 64-bit block counter, 64-bit nonce)
 * under a caller-supplied key.  Encryption number e = first_index + i of a key uses nonce e; 64-bit draw d of it is bytes
 * [8 d, 8 d + 8) of that stream (little endian): d = j for u_j, n + j for e1_j, 2 n + j for e2_j.  u_j = floor(3 r / 2^64) - 1;
 * the noise takes x = r >> 1, |e| = #{i : x >= cdt[i]} with cdt[i] = floor(2^63 P(|e| <= i)) (fhe_noise_cdt; integer work only,
 * so every implementation draws the same values), sign = low bit of r.  A (key, index) pair must never be used twice: draw the
 * key from the operating system's generator (getrandom) once per Encryptor and count.  tests/ pin the stream against the
 * published ChaCha20 vector, the table against a 90-digit evaluation, and the ciphertexts against the oracle's restatement. */
#define FHE_NOISE_CDT_LEN 19
void fhe_noise_cdt(uint64_t out[FHE_NOISE_CDT_LEN]);
/* FractionalEncoder::encode of `count` host doubles into d_plain [count][n] (device, coefficients below t): bit for bit
 * fhe_frac_encode of each value.  Asynchronous (the values travel through the staging ring). */
int fhe_frac_encode_batch(const fhe_ctx *ctx, const double *values, uint64_t count, int int_coeffs, int frac_coeffs,
                          uint64_t *d_plain, fhe_stream stream);
/* d_pk_ntt: the public key [2][k][n] in NTT form (fhe_ntt_forward of (pk0, pk1)); d_plain: [count][n] plaintext coefficients
 * below t, or NULL for encryptions of the zero plaintext; d_out: [count][2][k][n].  scratch: fhe_encrypt_scratch_bytes. */
size_t fhe_encrypt_scratch_bytes(const fhe_ctx *ctx, uint64_t count);
int fhe_encrypt_batch(const fhe_ctx *ctx, const uint64_t *d_pk_ntt, const uint64_t *d_plain, uint64_t count,
                      const uint8_t key[32], uint64_t first_index, uint64_t *d_out, void *scratch, size_t scratch_bytes,
                      fhe_stream stream);
/* the draws alone, for tests and for anyone who wants to check a ciphertext: d_draws [count][3][n] int8 (u, e1, e2) */
int fhe_encrypt_draws(const fhe_ctx *ctx, const uint8_t key[32], uint64_t first_index, uint64_t count, int8_t *d_draws,
                      fhe_stream stream);

/* ---- decryption in batches (the clients' half: homo/client_jpeg.cpp:266-280, homo/client_resize.cpp:190-210) -----------------------
 * seal::Decryptor::decrypt of `count` ciphertexts of `size` polynomials: phase = sum_j c_j s^j (Horner per NTT slot), then
 * m = floor((t x + floor(q/2)) / q) mod t EXACTLY for x = the CRT value of the phase -- formed per coefficient from the residues with
 * multi-word integers on the device (no big-integer loop on the host).  d_sk_ntt: the secret key [k][n] in NTT form; d_ct:
 * [count][size][k][n]; d_plain: [count][n] coefficients below t; d_noise_bits (or NULL): [count] u32, the bit length of the largest
 * |t x - m q| of each ciphertext -- seal::Decryptor::invariant_noise_budget = max(0, fhe_ctx_modulus_bits - that - 1).
 * Bit for bit the oracle's big-integer decryption (tests/test_gpu_encrypt.py), also beyond the noise budget. */
uint32_t fhe_ctx_modulus_bits(const fhe_ctx *ctx);
size_t fhe_decrypt_scratch_bytes(const fhe_ctx *ctx, uint32_t size, uint64_t count);
int fhe_decrypt_batch(const fhe_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_ct, uint32_t size, uint64_t count,
                      uint64_t *d_plain, uint32_t *d_noise_bits, void *scratch, size_t scratch_bytes, fhe_stream stream);

/* ---- synthetic inputs and digests (bench / parity harness) --------------------------------------
 * fill: value = splitmix64(seed ^ (first_linear_index + linear index)) mod q_i */
int fhe_fill_random(const fhe_ctx *ctx, uint64_t *ct, uint64_t n_polys, uint64_t seed,
                    uint64_t first_linear_index, fhe_stream stream);
/* order-independent 64-bit digest: sum over elements of splitmix64(value ^ splitmix64(index0+i))
 * mod 2^64, written to *d_out (device u64). */
int fhe_digest(const fhe_ctx *ctx, const uint64_t *data, uint64_t count, uint64_t index0,
               uint64_t *d_out, fhe_stream stream);

/* Input validation for ciphertext STREAMS (include/fhe_stream.h checks record headers only): ADDS to *d_count (device u64,
 * zeroed by the caller) the number of residues of the n_polys RNS polynomials at ct that are not below their modulus.
 * Every kernel assumes canonical residues -- seal::Ciphertext::load rejects anything else, and so does the facade's load --
 * so a server that takes streams from clients runs this on every wave it uploads (one HBM-bound pass, < 1 % of the PCIe
 * time of that wave) and refuses the job when the count is not zero.  homo/server_jpeg.cpp:117-123 is the load it guards. */
int fhe_count_unreduced(const fhe_ctx *ctx, const uint64_t *ct, uint64_t n_polys, uint64_t *d_count, fhe_stream stream);

#ifdef __cplusplus
}
#endif
#endif
