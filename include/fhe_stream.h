/*
 * fhe_stream.h -- C ABI of libfhe_hip.so, part 3: ciphertext STREAM records <-> host staging buffers.
 *
 * The reference's servers read and write their ciphertexts one `Ciphertext::load` / `save` at a time from a
 * std::fstream (homo/server_jpeg.cpp:115-124,150-152; homo/fhe_resize.h:335-341; homo/server_decode.cpp:131-143).
 * A stream is a concatenation of records of FIXED size -- "FHEHIP1\0", u32 polys, u32 k, u32 n, u32 reserved, then
 * polys * k * n little-endian u64 (seal/seal.h save_words) -- so a batch of them can be moved with positional scatter /
 * gather I/O by several threads at once, payloads landing contiguously in (page-locked) staging memory that the GPU
 * copies from.  Host-only code (no device work): `threads` POSIX threads each issue preadv / pwritev calls that cover
 * many records per system call.  Blocking; callable from any thread (the Python host calls it with the GIL released).
 *
 * What is checked: every record HEADER (magic, polys, k, n) and the file bounds.  What is NOT: the payload -- residues that
 * are not reduced modulo the coefficient moduli would be computed on silently (the kernels assume canonical inputs), so a
 * server that reads streams it did not write validates each uploaded wave with fhe_count_unreduced (include/fhe_hip.h);
 * server.server_jpeg / server_resize / server_decode and seal/server_jpeg_hip.cpp do.  A mapped stream that another process
 * truncates while it is being read raises SIGBUS, like any mapping: spool files belong to the server.
 */
#ifndef FHE_STREAM_H
#define FHE_STREAM_H

#include "fhe_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of one record holding a ciphertext of `polys` polynomials */
size_t fhe_io_record_bytes(uint32_t polys, uint32_t k, uint32_t n);
/* Read records [first_record, first_record + count) of the open file `fd` into dst (count * polys * k * n u64, payloads
 * only, in order); every header is checked against (polys, k, n).  FHE_ERR_PARAM on a foreign / mismatching record or a
 * short file. */
int fhe_io_read_records(int fd, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n, void *dst,
                        uint32_t threads);
/* Write `count` records at record index first_record (headers generated, payloads taken from src in order). */
int fhe_io_write_records(int fd, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n,
                         const void *src, uint32_t threads);

/* The same transfers through ONE long-lived shared mapping of the whole file -- the form the streaming servers use.
 * pwritev holds the file's inode lock exclusively, so any number of writer threads move data at the rate of one; stores
 * into a mapping scale with the threads.  fhe_io_open(path, write = 0) maps an existing stream for reading;
 * write = 1 creates the file if needed and gives it size_bytes -- an existing file of exactly that size keeps its pages
 * (a reused spool file: overwriting allocated page-cache pages is memcpy-bound, while first-touch allocation of fresh
 * pages is serialised inside the kernel whatever the method).  fhe_io_transfer reads (handle opened for reading) or
 * writes (opened for writing) `count` records from / to buf with `threads` threads. */
typedef struct fhe_io_file fhe_io_file;
int fhe_io_open(const char *path, int write, uint64_t size_bytes, fhe_io_file **out);
int fhe_io_close(fhe_io_file *file);
uint64_t fhe_io_size(const fhe_io_file *file);
int fhe_io_transfer(fhe_io_file *file, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n,
                    void *buf, uint32_t threads);

#ifdef __cplusplus
}
#endif
#endif
