/*
 * fhe_circuits.h -- C ABI of libfhe_hip.so, part 2: the reference's ciphertext x ciphertext CIRCUITS on whole
 * batches (the resize path of homo/fhe_resize.h and the decode path of homo/fhe_decode.h, plus the driver
 * loop of homo/server_decode.cpp), so that a C / C++ host gets the batched evaluation without any tensor
 * library in between.  Part 1 (include/fhe_hip.h) holds the context, the Evaluator primitives and the
 * fused JPEG circuit; everything here composes those primitives inside the library:
 * tap gathers, the t^2 reuse of Cubic, prepared (extended + transformed) operands, the growth of
 * ciphertext sizes, unequal-size additions and all temporaries live in caller-supplied scratch.
 *
 * Conventions: as in fhe_hip.h (plain pointers and sizes, `stream` = hipStream_t as void*, status codes,
 * fhe_last_error()).  Ciphertext batches are device memory, u64 [count][size][k][n], residues fully
 * reduced.  Index arrays (`taps`) and scalars are HOST memory; they are consumed before the call returns.
 * Every circuit has a `*_scratch_bytes` query that runs the same host logic without launching anything, so
 * the figure is exact for the arguments given (it does not depend on the circuit's scalar arguments -- order,
 * delta -- nor on whether a constant fits the encoder: those are checked by the call itself, before anything
 * is enqueued, together with the taps); scratch may be reused by the next call on the same stream.
 * Results are bit-identical to the reference's op-by-op evaluation through seal::Evaluator with the same
 * server-side encryptions (which the reference draws inside the circuits and which are INPUTS here:
 * the fractional offsets of SampleLinear / SampleBicubic, the Enc(0) accumulators of homomorphic_sin/cos).
 */
#ifndef FHE_CIRCUITS_H
#define FHE_CIRCUITS_H

#include "fhe_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Constants of the circuits for one context and one seal::FractionalEncoder(t, poly, int_coeffs,
 * frac_coeffs, 2) (homo/server_resize.cpp:110, homo/server_decode.cpp:117): each distinct plaintext is encoded,
 * lifted and transformed once and kept on the device (the reference re-encodes on every call).  Thread-safe;
 * must outlive every call that uses it; the context must outlive it. */
typedef struct fhe_circuits fhe_circuits;
int fhe_circuits_create(const fhe_ctx *ctx, int int_coeffs, int frac_coeffs, fhe_circuits **out);
int fhe_circuits_destroy(fhe_circuits *circ);

/* ---- the relinearised mode ---------------------------------------------------
 * The reference multiplies ciphertexts without ever relinearising (homo/fhe_resize.h:174-179, homo/fhe_decode.h:67-98,
 * 235,239): sizes reach 6 in SampleBicubic and 22 in approximated_step.  It does carry the decomposition bit count it
 * would need (`dbc`, homo/client_resize.cpp:26,47,72, homo/client_decode.cpp:26, DBC = 30 homo/fhe_image.h:28) and never
 * uses it.  A handle made by fhe_circuits_create_relin evaluates the SAME Evaluator call sequences with
 * evaluator.relinearize(x, evk) after every multiply / square -- each product 2 x 2 -> 3 -> 2 -- so every ciphertext
 * of every circuit has TWO polynomials: all `out` arrays below then hold [..][2][k][n] (fhe_circuits_out_size tells),
 * and the `size` argument of fhe_cubic / fhe_linear must be 2.  These are NOT the reference's ciphertext bits (key
 * switching adds its own noise term); they decrypt to the same values with a larger remaining noise budget, and they are
 * bit-identical to the oracle's op-by-op composition multiply -> relinearize (oracle/oracle.py RelinOracle).
 * d_evk_ntt: evaluation keys for s^2 as fhe_relinearize takes them ([k][fhe_evk_digits(dbc)][2][k][n], device memory,
 * NTT form); they must outlive the handle.  Everything else is as for fhe_circuits_create. */
int fhe_circuits_create_relin(const fhe_ctx *ctx, int int_coeffs, int frac_coeffs, const uint64_t *d_evk_ntt, uint32_t dbc,
                              fhe_circuits **out);
/* WHERE the relinearised mode relinearises.
 *   FHE_RELIN_EVERY_PRODUCT  (fhe_circuits_create_relin) after every multiply / square, as described above: five key switches per
 *                            Cubic (t^2, t * t and the three products), two per Linear; every circuit of this header.
 *   FHE_RELIN_PER_CUBIC      the reference's call sequence of Cubic / Linear UNCHANGED (homo/fhe_resize.h:150-184,196-199: products
 *                            grow to 3 and 4 polynomials, the fused tail of fhe_cubic included) and ONE evaluator.relinearize(result, evk)
 *                            at the end of each, taking the size-4 (Cubic) / size-3 (Linear) result to 2 the way SEAL's relinearize
 *                            does: one key switch per polynomial above the second, the top one first -- TWO per Cubic, one per
 *                            Linear.  d_evk_ntt then holds the keys for s^2 followed by the keys for s^3 (fhe_evk_words(ctx, dbc)
 *                            words each; KeyGenerator::generate_evaluation_keys(dbc, 2, keys)).  Operands and results of fhe_cubic /
 *                            fhe_linear / the samplers / the shared resize have two polynomials, as in the other placement; bit-identical
 *                            to the oracle's composition `reference sequence -> relinearize` (oracle/oracle.py TailRelinOracle).  The
 *                            decode circuits (sin / cos, approximated_step, decode_channel) have no Cubic to end: they return
 *                            FHE_ERR_PARAM for such a handle.
 *   FHE_RELIN_PER_SAMPLE     the SAMPLERS' call sequences unchanged -- SampleBicubic: five Cubics in the reference's mode, sizes 2 -> 4 -> 6
 *                            (homo/fhe_resize.h:293-303); SampleLinear: three Linears, 2 -> 3 -> 4 (:237-248); the shared-offset resize the same --
 *                            and ONE evaluator.relinearize of every OUTPUT pixel (6 -> 2: four key switches, keys for s^2 .. s^5; 4 -> 2 for
 *                            bilinear), which fhe_relinearize_n runs as one pass at dbc 60 and as two at dbc 30.  d_evk_ntt holds FOUR key sets
 *                            (generate_evaluation_keys(dbc, 4, keys)).  Costs one pass of key switches per output pixel where PER_CUBIC costs
 *                            five: the device time of the reference's mode + 2-6 %, with records of two polynomials.  fhe_cubic / fhe_linear on
 *                            their own relinearise their result (any operand size the keys reach: size + 2 <= 6).  Checker:
 *                            oracle/oracle.py SampleRelinOracle.  Resize circuits only, like PER_CUBIC. */
#define FHE_RELIN_EVERY_PRODUCT 0
#define FHE_RELIN_PER_CUBIC 1
#define FHE_RELIN_PER_SAMPLE 2
int fhe_circuits_create_relin_at(const fhe_ctx *ctx, int int_coeffs, int frac_coeffs, const uint64_t *d_evk_ntt, uint32_t dbc,
                                 uint32_t placement, fhe_circuits **out);
uint32_t fhe_circuits_relin_placement(const fhe_circuits *circ);
/* the decomposition bit count of a relinearising handle, 0 for a handle in the reference's mode */
uint32_t fhe_circuits_relin_dbc(const fhe_circuits *circ);
/* polynomials per output ciphertext of a circuit for this handle.  `arg`: FHE_CIRC_CUBIC / FHE_CIRC_LINEAR the operand
 * size, FHE_CIRC_STEP / FHE_CIRC_DECODE the degree (FHE_CIRC_DECODE: of a channel with at least one run), else ignored.
 * Reference mode: size + 2, size + 1, 6, 4, 11, 22 (3 for degree 0), the same; relinearised mode: 2 throughout. */
#define FHE_CIRC_CUBIC 0
#define FHE_CIRC_LINEAR 1
#define FHE_CIRC_SAMPLE_BICUBIC 2
#define FHE_CIRC_SAMPLE_LINEAR 3
#define FHE_CIRC_SINCOS 4
#define FHE_CIRC_STEP 5
#define FHE_CIRC_DECODE 6
uint32_t fhe_circuits_out_size(const fhe_circuits *circ, int circuit, uint32_t arg);

/* ---- resize path ---------------------------------------------------------------------------------------
 * Index arithmetic of ResizeImage / SampleBicubic / SampleLinear / GetPixelClamped in the reference's float
 * arithmetic (homo/fhe_resize.h:350-351,381-382,260-290,215-220): per output pixel (row-major) the clamped
 * source pixel indices y * src_w + x -- 16 for bicubic (row-major 4 x 4: x-1..x+2 fastest, y-1..y+2), 4 for
 * bilinear (p00, p10, p01, p11) -- and the fractional offsets frac(u), frac(v).  Host only.  Any output
 * pointer may be NULL. */
int fhe_resize_sample_plan(uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, int bicubic,
                           uint32_t *taps, double *xfract, double *yfract);

/* Cubic(result, A, B, C, D, t) (homo/fhe_resize.h:143-189) for `count` independent 5-tuples: A..D of `size`
 * polynomials, t of size 2, out of size + 2 (relinearised mode: size must be 2, out has 2).  t2 = square(t) and
 * t3 = multiply(t, t) (:174-175, the reference's t^3 IS t^2) are one ring element, formed once. */
size_t fhe_cubic_scratch_bytes(const fhe_circuits *circ, uint32_t size, uint64_t count);
int fhe_cubic(const fhe_circuits *circ, const uint64_t *A, const uint64_t *B, const uint64_t *C, const uint64_t *D,
              uint32_t size, const uint64_t *t, uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes,
              fhe_stream stream);
/* Linear(result, A, B, t) = (1 - t) A + t B (homo/fhe_resize.h:191-204); out has size + 1 polynomials (relinearised
 * mode: size must be 2, out has 2). */
size_t fhe_linear_scratch_bytes(const fhe_circuits *circ, uint32_t size, uint64_t count);
int fhe_linear(const fhe_circuits *circ, const uint64_t *A, const uint64_t *B, uint32_t size, const uint64_t *t,
               uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream stream);

/* SampleBicubic (homo/fhe_resize.h:254-305) for `count` output pixels of one colour channel.
 * pixels: [n_pixels][2][k][n] (the resident source window); taps: host, [count][16] indices into `pixels`
 * (fhe_resize_sample_plan order); xfract, yfract: [count][2][k][n] encryptions of the offsets (:262,266);
 * out: [count][6][k][n] ([count][2][k][n] in the relinearised mode).  Five Cubic evaluations per pixel; xfract^2 and the prepared forms of xfract,
 * xfract^2 are shared by the four row Cubics. */
size_t fhe_sample_bicubic_scratch_bytes(const fhe_circuits *circ, uint64_t count);
int fhe_sample_bicubic(const fhe_circuits *circ, const uint64_t *pixels, uint64_t n_pixels, const uint32_t *taps,
                       const uint64_t *xfract, const uint64_t *yfract, uint64_t *out, uint64_t count, void *scratch,
                       size_t scratch_bytes, fhe_stream stream);
/* SampleLinear (homo/fhe_resize.h:222-252): taps [count][4]; out [count][4][k][n] (relinearised mode: [count][2][k][n]). */
size_t fhe_sample_linear_scratch_bytes(const fhe_circuits *circ, uint64_t count);
int fhe_sample_linear(const fhe_circuits *circ, const uint64_t *pixels, uint64_t n_pixels, const uint32_t *taps,
                      const uint64_t *xfract, const uint64_t *yfract, uint64_t *out, uint64_t count, void *scratch,
                      size_t scratch_bytes, fhe_stream stream);

/* ResizeImage with SampleBicubic (homo/fhe_resize.h:308-392) for one colour channel of a resident source
 * image when the offsets arrive as ONE ciphertext per output column (xfract [dst_w][2][k][n]) and ONE per
 * output row (yfract [dst_h][2][k][n]) -- SURVEY.md section 8(d), configs[2]: frac(u) depends on the column
 * only and frac(v) on the row only (:351,382).  Every repeated ring element is formed once (row Cubics of
 * overlapping 4-row windows, squares and prepared operands per column / row); each output equals
 * fhe_sample_bicubic with xfract[x], yfract[y] bit for bit.  Output pixels are produced in bands of
 * `band_rows` destination rows, row-major: written to out ([dst_w * dst_h][S][k][n], S = fhe_circuits_out_size(circ,
 * FHE_CIRC_SAMPLE_BICUBIC, 0) = 6, or 2 in the relinearised mode) when out != NULL, and /
 * or handed to `consume` (may be NULL) as consume(user, first_pixel, d_band, n_pixels, stream), which must
 * enqueue its reads of d_band on `stream` (the band buffer is reused).  `batch` bounds the Cubics per launch
 * sequence (rounded to whole rows of dst_w). */
typedef int (*fhe_band_consumer)(void *user, uint64_t first_pixel, const uint64_t *d_band, uint64_t n_pixels,
                                 fhe_stream stream);
size_t fhe_resize_bicubic_shared_scratch_bytes(const fhe_circuits *circ, uint32_t src_w, uint32_t src_h,
                                               uint32_t dst_w, uint32_t dst_h, uint32_t batch, uint32_t band_rows,
                                               int has_out);
int fhe_resize_bicubic_shared(const fhe_circuits *circ, const uint64_t *pixels, uint32_t src_w, uint32_t src_h,
                              uint32_t dst_w, uint32_t dst_h, const uint64_t *xfract, const uint64_t *yfract,
                              uint64_t *out, uint32_t batch, uint32_t band_rows, fhe_band_consumer consume,
                              void *user, void *scratch, size_t scratch_bytes, fhe_stream stream);

/* A SHARD of the destination rows: the multi-GPU partition of ResizeImage's outer loop (homo/fhe_resize.h:350 `for y`;
 * BASELINE.json north_star: "the per-output-pixel bicubic-weight circuit [is] embarrassingly parallel ... partition ...
 * across the 8 GPUs").  Destination rows [row0, row1) are independent of all others; they read the source rows
 * fhe_resize_source_rows reports (the shard's rows plus the sampler's halo: yi-1 .. yi+2 for bicubic, yi .. yi+1 for
 * bilinear, clamped; :264-290,229-240) and nothing else, so every GPU loads its rows +- the halo and no exchange is
 * needed.  Host only. */
int fhe_resize_source_rows(uint32_t src_h, uint32_t dst_h, uint32_t row0, uint32_t row1, int bicubic, uint32_t *first,
                           uint32_t *count);
/* fhe_resize_bicubic_shared for destination rows [row0, row1) only.  pixels: the source rows
 * [src_row0, src_row0 + n_src_rows) ([n_src_rows * src_w][2][k][n]; must cover what fhe_resize_source_rows reports, else
 * FHE_ERR_PARAM before anything is enqueued); xfract: [dst_w][2][k][n] as before; yfract: the offsets of rows
 * [row0, row1) only ([row1 - row0][2][k][n]); out: [(row1 - row0) * dst_w][6][k][n].  The consumer's first_pixel stays the
 * GLOBAL pixel index row * dst_w.  Every output equals the whole-image call's bit for bit. */
size_t fhe_resize_bicubic_shared_rows_scratch_bytes(const fhe_circuits *circ, uint32_t src_w, uint32_t src_h, uint32_t dst_w,
                                                    uint32_t dst_h, uint32_t row0, uint32_t row1, uint32_t src_row0,
                                                    uint32_t n_src_rows, uint32_t batch, uint32_t band_rows, int has_out);
int fhe_resize_bicubic_shared_rows(const fhe_circuits *circ, const uint64_t *pixels, uint32_t src_w, uint32_t src_h,
                                   uint32_t dst_w, uint32_t dst_h, uint32_t row0, uint32_t row1, uint32_t src_row0,
                                   uint32_t n_src_rows, const uint64_t *xfract, const uint64_t *yfract, uint64_t *out,
                                   uint32_t batch, uint32_t band_rows, fhe_band_consumer consume, void *user, void *scratch,
                                   size_t scratch_bytes, fhe_stream stream);

/* ---- decode path ---------------------------------------------------------------------------------------
 * homomorphic_sin (cosine = 0, homo/fhe_decode.h:48-120) / homomorphic_cos (cosine = 1, :128-200; the value
 * the reference leaves in `res` before falling off the end without a return statement) for `count`
 * arguments: x, zero (the Enc(0) of :54 / :134): [count][2][k][n]; out: [count][11][k][n] (relinearised mode: [count][2][k][n]). */
size_t fhe_homomorphic_sincos_scratch_bytes(const fhe_circuits *circ, uint64_t count);
int fhe_homomorphic_sincos(const fhe_circuits *circ, int cosine, const uint64_t *x, const uint64_t *zero,
                           uint64_t *out, uint64_t count, void *scratch, size_t scratch_bytes, fhe_stream stream);

/* The homomorphic overload of approximated_step (homo/fhe_decode.h:202-242) for ONE run (amplitude, index,
 * count_ct: [2][k][n]) over npos = width * height positions.  zeros: [npos][degree][2][2][k][n], the Enc(0)
 * accumulators in the reference's call order (position i, harmonic j = 1..degree, sin then cos).
 * out: [npos][fhe_approximated_step_out_size(degree)][k][n] (22 polynomials for degree >= 1; a relinearising handle
 * writes [npos][2][k][n]: fhe_circuits_out_size(circ, FHE_CIRC_STEP, degree)).
 * `offset` advances by add_plain(offset, encode(i)) inside the harmonic loop (:229), as the reference does. */
uint32_t fhe_approximated_step_out_size(int degree);
size_t fhe_approximated_step_scratch_bytes(const fhe_circuits *circ, int degree, uint32_t npos);
int fhe_approximated_step(const fhe_circuits *circ, const uint64_t *amplitude, const uint64_t *index,
                          const uint64_t *count_ct, int order, int degree, double delta, uint32_t width,
                          uint32_t height, const uint64_t *zeros, uint64_t *out, void *scratch,
                          size_t scratch_bytes, fhe_stream stream);

/* Output positions [pos0, pos1) of the same run: the multi-GPU partition of the position loop (homo/fhe_decode.h:224;
 * "decode shards by (run, output position)").  zeros and out hold the shard's positions only
 * ([pos1 - pos0][degree][2][2][k][n], [pos1 - pos0][S][k][n]).  The serial part of the loop -- `offset` advancing by
 * add_plain inside the harmonic loop (:229) -- is replayed from position 0 by every shard (pos0 * degree polynomial
 * additions, no products), so each output equals the whole-run call's bit for bit. */
size_t fhe_approximated_step_range_scratch_bytes(const fhe_circuits *circ, int degree, uint32_t npos, uint32_t pos0,
                                                 uint32_t pos1);
int fhe_approximated_step_range(const fhe_circuits *circ, const uint64_t *amplitude, const uint64_t *index,
                                const uint64_t *count_ct, int order, int degree, double delta, uint32_t width,
                                uint32_t height, uint32_t pos0, uint32_t pos1, const uint64_t *zeros, uint64_t *out,
                                void *scratch, size_t scratch_bytes, fhe_stream stream);

/* One colour channel of the server_decode driver loop (homo/server_decode.cpp:120-137): the channel's
 * accumulators start as acc0 ([npos][2][k][n], the Enc(0) of :126); for each of `pairs` runs
 * (runs: [pairs][2][2][k][n] = elem, count as loaded at :131-132) approximated_step is evaluated with the
 * running `index` ([2][k][n], in/out: :121 and :137 `index += count`), and its npos results are added to the
 * accumulators (:134-136).  zeros: [pairs][npos][degree][2][2][k][n].
 * out: [npos][S][k][n] with S = fhe_circuits_out_size(circ, FHE_CIRC_DECODE, degree) (= fhe_approximated_step_out_size(degree)
 * in the reference's mode) if pairs > 0, else 2. */
size_t fhe_decode_channel_scratch_bytes(const fhe_circuits *circ, int degree, uint32_t npos, uint32_t pairs);
int fhe_decode_channel(const fhe_circuits *circ, const uint64_t *runs, uint32_t pairs, uint64_t *index,
                       const uint64_t *acc0, const uint64_t *zeros, int order, int degree, double delta,
                       uint32_t width, uint32_t height, uint64_t *out, void *scratch, size_t scratch_bytes,
                       fhe_stream stream);

/* Positions [pos0, pos1) of one channel: acc0 ([pos1 - pos0][2][k][n]), zeros ([pairs][pos1 - pos0][degree][2][2][k][n])
 * and out hold the shard's positions only; every shard owns a copy of `index` and advances it through all runs
 * (`pairs` polynomial additions -- the replicated prefix chain). */
size_t fhe_decode_channel_range_scratch_bytes(const fhe_circuits *circ, int degree, uint32_t npos, uint32_t pos0,
                                              uint32_t pos1, uint32_t pairs);
int fhe_decode_channel_range(const fhe_circuits *circ, const uint64_t *runs, uint32_t pairs, uint64_t *index,
                             const uint64_t *acc0, const uint64_t *zeros, int order, int degree, double delta,
                             uint32_t width, uint32_t height, uint32_t pos0, uint32_t pos1, uint64_t *out, void *scratch,
                             size_t scratch_bytes, fhe_stream stream);

#ifdef __cplusplus
}
#endif
#endif
