// stream_io.hip -- include/fhe_stream.h: fixed-size ciphertext records between a file and contiguous staging memory,
// by positional scatter / gather I/O from several threads (host code only; built with the library's other units).
#include "internal.h"
#include "../../include/fhe_stream.h"

#include <errno.h>
#include <string.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
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

extern "C" size_t fhe_io_record_bytes(uint32_t polys, uint32_t k, uint32_t n) { return kHeader + (size_t)polys * k * n * 8; }
extern "C" int fhe_io_read_records(int fd, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n, void *dst, uint32_t threads) {
    return run(false, fd, first_record, count, polys, k, n, dst, threads);
}
extern "C" int fhe_io_write_records(int fd, uint64_t first_record, uint64_t count, uint32_t polys, uint32_t k, uint32_t n, const void *src, uint32_t threads) {
    return run(true, fd, first_record, count, polys, k, n, const_cast<void *>(src), threads);
}
