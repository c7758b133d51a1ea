// stream_io.hip -- include/fhe_stream.h: fixed-size ciphertext records between a file and contiguous staging memory,
// by positional scatter / gather I/O from several threads (host code only; built with the library's other units).
// Nothing here needs the HIP headers: the unit also compiles as plain C++ (g++ -x c++), which is how the sanitizer / fuzz
// build of oracle/Makefile (target `asan`, oracle/stream_fuzz_main.cpp) takes it.
#include "../../include/fhe_stream.h"

#include <string>
#include <vector>
typedef unsigned long long u64;      // as in modarith.h
typedef unsigned int u32;
int fhe_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));      // fhe_hip.hip: sets fhe_last_error()
#define fail fhe_fail

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
#include <memory>
#include <thread>

namespace {
constexpr size_t kHeader = 24;
constexpr int kRecordsPerCall = 256;          // 2 iovecs each: below IOV_MAX (1024)

struct Header { char magic[8]; u32 polys, k, n, reserved; };
static_assert(sizeof(Header) == kHeader, "record header layout");

void make_header(Header &h, u32 polys, u32 k, u32 n) {
    memcpy(h.magic, "FHEHIP1\0", 8);
    h.polys = polys; h.k = k; h.n = n; h.reserved = 0;
}

// one worker: records [lo, hi) of the batch
int worker(bool write, int fd, u64 first_record, u64 lo, u64 hi, u32 polys, u32 k, u32 n, unsigned char *mem, std::string *err) {
    const size_t payload = (size_t)polys * k * n * 8, rec = kHeader + payload;
    Header expect;
    make_header(expect, polys, k, n);
    std::vector<Header> hdr(kRecordsPerCall);
    std::vector<iovec> iov(2 * kRecordsPerCall);
    for (u64 r0 = lo; r0 < hi; r0 += kRecordsPerCall) {
        const int cnt = (int)(hi - r0 < (u64)kRecordsPerCall ? hi - r0 : kRecordsPerCall);
        for (int i = 0; i < cnt; ++i) {
            if (write) hdr[i] = expect;
            iov[2 * i] = iovec{&hdr[i], kHeader};
            iov[2 * i + 1] = iovec{mem + (r0 + i) * payload, payload};
        }
        // a positional call may transfer less than asked (2 GiB limit per call, signals): continue where it stopped
        size_t done = 0;
        const size_t total = (size_t)cnt * rec;
        int first_iov = 0;
        while (done < total) {
            const off_t at = (off_t)((first_record + r0) * rec + done);
            const ssize_t got = write ? pwritev(fd, iov.data() + first_iov, 2 * cnt - first_iov, at) : preadv(fd, iov.data() + first_iov, 2 * cnt - first_iov, at);
            if (got < 0) {
                if (errno == EINTR) continue;
                *err = std::string(write ? "pwritev: " : "preadv: ") + strerror(errno);
                return FHE_ERR_PARAM;
            }
            if (got == 0) { *err = "ciphertext stream ended"; return FHE_ERR_PARAM; }
            done += (size_t)got;
            size_t left = (size_t)got;                     // advance the iovec cursor past what was transferred
            while (left && first_iov < 2 * cnt) {
                if (left >= iov[first_iov].iov_len) { left -= iov[first_iov].iov_len; ++first_iov; }
                else { iov[first_iov].iov_base = (char *)iov[first_iov].iov_base + left; iov[first_iov].iov_len -= left; left = 0; }
            }
        }
        if (!write)
            for (int i = 0; i < cnt; ++i)
                if (memcmp(&hdr[i], &expect, 20) != 0) {     // magic, polys, k, n (the reserved word is ignored)
                    *err = memcmp(hdr[i].magic, expect.magic, 8) ? "not a ciphertext record" : "ciphertext record does not match the context (polys, k, n)";
                    return FHE_ERR_PARAM;
                }
    }
    return FHE_OK;
}

int run(bool write, int fd, u64 first_record, u64 count, u32 polys, u32 k, u32 n, void *mem, u32 threads) {
    if (fd < 0 || !mem) return fail(FHE_ERR_PARAM, "bad file descriptor or null buffer");
    if (!polys || !k || !n || k > FHE_MAX_K) return fail(FHE_ERR_PARAM, "bad record shape");
    if (!count) return FHE_OK;
    if (!threads) threads = 1;
    if (threads > 64) threads = 64;
    if ((u64)threads > count) threads = (u32)count;
    std::vector<std::thread> pool;
    std::vector<int> rc(threads, FHE_OK);
    std::vector<std::string> err(threads);
    for (u32 t = 0; t < threads; ++t) {
        const u64 lo = count * t / threads, hi = count * (t + 1) / threads;
        pool.emplace_back([=, &rc, &err] { rc[t] = worker(write, fd, first_record, lo, hi, polys, k, n, (unsigned char *)mem, &err[t]); });
    }
    for (auto &th : pool) th.join();
    for (u32 t = 0; t < threads; ++t)
        if (rc[t]) return fail(rc[t], "%s", err[t].c_str());
    return FHE_OK;
}
}  // namespace

// ---- whole-file mappings ---------------------------------------------------------------------------------------------
// pwritev takes the file's inode lock exclusively (tmpfs and most file systems): any number of writer threads move data at
// the rate of one.  Stores into ONE long-lived MAP_SHARED mapping scale with the threads -- provided the pages exist:
// first-touch allocation of page-cache pages is serialised inside the kernel (measured on tmpfs: 3.2 GB/s with 16 threads
// through a mapping, less with more threads, ~4.5 GB/s through pwritev), so a fresh output file is allocation-bound
// whatever the method, and an output file whose pages already exist (a reused spool file) is memcpy-bound.
struct fhe_io_file {
    int fd = -1;
    bool write = false;
    unsigned char *map = nullptr;
    size_t size = 0;
};
extern "C" int fhe_io_open(const char *path, int write, uint64_t size_bytes, fhe_io_file **out) {
    if (!path || !out) return fail(FHE_ERR_PARAM, "null argument");
    *out = nullptr;
    std::unique_ptr<fhe_io_file> f(new fhe_io_file);
    f->write = write != 0;
    f->fd = open(path, write ? (O_RDWR | O_CREAT) : O_RDONLY, 0644);
    if (f->fd < 0) return fail(FHE_ERR_PARAM, "open %s: %s", path, strerror(errno));
    struct stat stb;
    if (fstat(f->fd, &stb) != 0) { close(f->fd); return fail(FHE_ERR_PARAM, "fstat: %s", strerror(errno)); }
    if (write) {
        // an existing file of exactly this size keeps its pages (no truncation): overwriting allocated pages is memcpy-bound
        if ((uint64_t)stb.st_size != size_bytes && ftruncate(f->fd, (off_t)size_bytes) != 0) { close(f->fd); return fail(FHE_ERR_PARAM, "ftruncate: %s", strerror(errno)); }
        f->size = (size_t)size_bytes;
    } else {
        f->size = (size_t)stb.st_size;
    }
    if (f->size) {
        void *m = mmap(nullptr, f->size, write ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, f->fd, 0);
        if (m == MAP_FAILED) { close(f->fd); return fail(FHE_ERR_PARAM, "mmap: %s", strerror(errno)); }
        f->map = (unsigned char *)m;
    }
    *out = f.release();
    return FHE_OK;
}
extern "C" int fhe_io_close(fhe_io_file *f) {
    if (!f) return FHE_OK;
    if (f->map) munmap(f->map, f->size);
    if (f->fd >= 0) close(f->fd);
    delete f;
    return FHE_OK;
}
extern "C" uint64_t fhe_io_size(const fhe_io_file *f) { return f ? f->size : 0; }
extern "C" int fhe_io_transfer(fhe_io_file *f, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n, void *buf, uint32_t threads) {
    if (!f || !buf) return fail(FHE_ERR_PARAM, "null argument");
    if (!polys || !k || !n || k > FHE_MAX_K) return fail(FHE_ERR_PARAM, "bad record shape");
    if (!count) return FHE_OK;
    const size_t payload = (size_t)polys * k * n * 8, rec = kHeader + payload;
    if (first_record > f->size / rec || count > f->size / rec - first_record) return fail(FHE_ERR_PARAM, "ciphertext stream ended");     // no u64 wrap for large record numbers
    if (!threads) threads = 1;
    if (threads > 64) threads = 64;
    if ((u64)threads > count) threads = (u32)count;
    Header expect;
    make_header(expect, polys, k, n);
    unsigned char *base = f->map + first_record * rec, *mem = (unsigned char *)buf;
    const bool write = f->write;
    std::atomic<int> bad{0};
    std::vector<std::thread> pool;
    for (u32 t = 0; t < threads; ++t) {
        const u64 a = count * t / threads, z = count * (t + 1) / threads;
        pool.emplace_back([=, &bad] {
            for (u64 r = a; r < z; ++r) {
                unsigned char *p = base + r * rec;
                if (write) {
                    memcpy(p, &expect, kHeader);
                    memcpy(p + kHeader, mem + r * payload, payload);
                } else {
                    if (memcmp(p, &expect, 20) != 0) { bad.store(memcmp(p, expect.magic, 8) ? 1 : 2); return; }
                    memcpy(mem + r * payload, p + kHeader, payload);
                }
            }
        });
    }
    for (auto &th : pool) th.join();
    if (bad.load()) return fail(FHE_ERR_PARAM, bad.load() == 1 ? "not a ciphertext record" : "ciphertext record does not match the context (polys, k, n)");
    return FHE_OK;
}

extern "C" size_t fhe_io_record_bytes(uint32_t polys, uint32_t k, uint32_t n) { return kHeader + (size_t)polys * k * n * 8; }
extern "C" int fhe_io_read_records(int fd, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n, void *dst, uint32_t threads) {
    return run(false, fd, first_record, count, polys, k, n, dst, threads);
}
extern "C" int fhe_io_write_records(int fd, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n, const void *src, uint32_t threads) {
    return run(true, fd, first_record, count, polys, k, n, const_cast<void *>(src), threads);
}
