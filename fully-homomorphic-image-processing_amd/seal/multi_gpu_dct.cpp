// multi_gpu_dct.cpp -- the N > 1 path from a C++ host, no Python and no torch: the block loop of homo/server_jpeg.cpp:113-138
// (encrypted_dct + quantize_fhe on independent 8x8 blocks) partitioned over the GPUs of one node through the C ABI
// (include/fhe_hip.h) -- one host thread, one fhe_ctx, one stream per rank -- with the optional final ciphertext gather
// as RCCL point-to-point transfers over xGMI (BASELINE.json north_star: "RCCL over xGMI only for the final ciphertext gather").
//
//   rank r of R owns the contiguous block range [r N / R, (r + 1) N / R) (the same split as parallel.block_range);
//   its synthetic inputs are seeded by the GLOBAL block index, so any R produces the same bytes;
//   every rank digests its own outputs with global indices: the sum of the rank digests is the digest of the whole output
//   and must equal what ONE rank computes alone over all N blocks (checked here, wave by wave, on rank 0's device);
//   gather (ranks on distinct devices only -- RCCL refuses two ranks on one device): every rank r > 0 sends each finished
//   wave of output ciphertexts to rank 0 with ncclSend while it computes the next wave; rank 0 posts the matching ncclRecv
//   of a wave from all peers as one group (each peer arrives over its own xGMI link) and digests what it received: that
//   digest must equal the sum of the senders' own digests.
//
// Ranks may outnumber devices (rank r runs on device r % devices): with one GPU the partition logic, the threading
// contract of the C ABI (contexts shared by nothing, fhe_ctx_bind_thread, per-rank streams) and the digests are still
// exercised; only the RCCL transfers need two devices.
//
// Two timing modes:
//   verify   (default) every wave generates its inputs and digests its outputs INSIDE the loop: two wave buffers, any job size;
//            a verifier -- its rate is about half of what the kernels deliver (fill + digest are two more passes over 12 MiB per block)
//   resident the shape of bench.py's timed region: the rank's inputs are generated BEFORE the clock starts and stay in HBM, the
//            outputs of every wave stay in HBM too and are digested AFTER the last synchronisation, `reps` passes over the shard are
//            timed: nothing but fhe_dct8x8_quant (and, with gather, the transfers and the root's drain) is inside the clock.  The
//            single-rank figure next to it is ONE rank over the per-rank share (weak scaling: the same work per GPU).
//
// usage: multi_gpu_dct [total_blocks=512] [ranks=<device count>] [wave_blocks=64] [gather=1] [mode=verify|resident] [reps=1]
// prints one JSON line; exit code 0 iff every check held.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "fhe_hip.h"

namespace {
const double YQT[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                        18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};   // homo/fhe_image.h:99
constexpr uint64_t SEED = 0x5EA12026ULL;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Fail {
    std::string what;
};
void check(int rc, const char *what) {
    if (rc < 0) throw Fail{std::string(what) + ": " + fhe_last_error()};
}
void hcheck(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Fail{std::string(what) + ": " + hipGetErrorString(e)};
}
void ncheck(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw Fail{std::string(what) + ": " + ncclGetErrorString(r)};
}

// [start, end) of rank r: sizes differ by at most one (parallel.block_range)
void block_range(uint64_t r, uint64_t world, uint64_t n, uint64_t &start, uint64_t &end) {
    const uint64_t base = n / world, rem = n % world;
    start = r * base + (r < rem ? r : rem);
    end = start + base + (r < rem ? 1 : 0);
}

struct Rank {
    int rank = 0, device = 0;
    uint64_t start = 0, end = 0, digest = 0, received_digest = 0;
    double seconds = 0;
    std::string error;
};

struct Job {
    uint32_t n = 4096, k = 3;
    uint64_t q[FHE_MAX_K] = {0}, t = 1 << 14;
    uint64_t total = 512, wave = 64, words_per_block = 0;
    int world = 1, devices = 1, reps = 1;
    bool gather = false, resident = false;
    std::vector<ncclComm_t> comms;
};

// one device buffer of `words` u64 through the ABI's allocator (the calling thread's current device)
uint64_t *dalloc(uint64_t words) {
    void *p = nullptr;
    check(fhe_dev_alloc(words * 8, &p), "fhe_dev_alloc");
    return (uint64_t *)p;
}

// digest of one wave into ITS OWN device slot, asynchronously on `st`: no host round trip inside the wave loop (the slots are
// brought back and summed once, after the loop).  Round 4 synchronised the stream here -- once per peer per wave on the root,
// so the receives of wave w + 1 could not be posted before seven digests of wave w had each gone to the host and back: fine
// for a verifier, wrong for the thing the first real multi-GPU run is going to time.
void digest_into(const fhe_ctx *ctx, const uint64_t *d, uint64_t words, uint64_t index0, uint64_t *d_slot, fhe_stream st) {
    check(fhe_digest(ctx, d, words, index0, d_slot, st), "fhe_digest");
}
uint64_t sum_slots(const uint64_t *d_slots, uint64_t count, fhe_stream st) {
    std::vector<uint64_t> h(count, 0);
    if (count) {
        check(fhe_download(h.data(), d_slots, count * 8, st), "fhe_download");
        check(fhe_stream_sync(st), "fhe_stream_sync");
    }
    uint64_t s = 0;
    for (uint64_t v : h) s += v;
    return s;
}

void run_rank(const Job &J, Rank &R, std::atomic<int> &arrived) {
    fhe_ctx *ctx = nullptr;
    fhe_dct_plan *plan = nullptr;
    fhe_stream st = nullptr, st_comm = nullptr, st_dig = nullptr;
    try {
        hcheck(hipSetDevice(R.device), "hipSetDevice");
        check(fhe_ctx_create(J.n, J.q, J.k, J.t, R.device, &ctx), "fhe_ctx_create");     // one context per rank: nothing is shared between ranks
        check(fhe_ctx_bind_thread(ctx), "fhe_ctx_bind_thread");
        check(fhe_stream_create(&st), "fhe_stream_create");
        check(fhe_stream_create(&st_comm), "fhe_stream_create");
        check(fhe_stream_create(&st_dig), "fhe_stream_create");                            // the root digests received waves here, beside the next wave's receives
        check(fhe_dct_plan_create(ctx, YQT, 100, 100, st, &plan), "fhe_dct_plan_create");
        const uint64_t wpb = J.words_per_block, mine = R.end - R.start;
        const uint64_t n_waves = (mine + J.wave - 1) / J.wave;
        const bool res = J.resident;
        std::vector<void *> owned;                                                      // freed before the rank returns (a second run follows in this process)
        auto take = [&](uint64_t words) { uint64_t *p = dalloc(words); owned.push_back(p); return p; };
        // verify: one input wave and two output waves; resident: the whole shard in and out
        uint64_t *in = take((res ? (mine ? mine : 1) : J.wave) * wpb);
        uint64_t *out_all = res ? take((mine ? mine : 1) * wpb) : nullptr;
        uint64_t *out[2] = {res ? nullptr : take(J.wave * wpb), res ? nullptr : take(J.wave * wpb)};
        const size_t scr_bytes = fhe_dct8x8_scratch_bytes(ctx, J.wave);
        void *scr = nullptr;
        check(fhe_dev_alloc(scr_bytes, &scr), "fhe_dev_alloc(scratch)");
        owned.push_back(scr);
        hipEvent_t computed[2], sent[2];
        for (int i = 0; i < 2; ++i) {
            hcheck(hipEventCreateWithFlags(&computed[i], hipEventDisableTiming), "event");
            hcheck(hipEventCreateWithFlags(&sent[i], hipEventDisableTiming), "event");
        }
        // the root's receive buffers: TWO waves per peer, so that wave w + 1 arrives while wave w is being digested
        std::vector<uint64_t *> rx[2];
        hipEvent_t received[2], digested[2];
        for (int i = 0; i < 2; ++i) {
            hcheck(hipEventCreateWithFlags(&received[i], hipEventDisableTiming), "event");
            hcheck(hipEventCreateWithFlags(&digested[i], hipEventDisableTiming), "event");
        }
        // every rank has the same number of waves up to one: the root posts receives for the longest peer shard, peers send
        // zero-length nothing for a missing last wave (wave counts are computed from the same split on both sides)
        std::vector<uint64_t> peer_blocks(J.world, 0);
        uint64_t max_waves = n_waves;
        for (int r = 0; r < J.world; ++r) {
            uint64_t s, e;
            block_range(r, J.world, J.total, s, e);
            peer_blocks[r] = e - s;
            const uint64_t w = (e - s + J.wave - 1) / J.wave;
            if (w > max_waves) max_waves = w;
        }
        if (J.gather && R.rank == 0)
            for (int b = 0; b < 2; ++b)
                for (int r = 1; r < J.world; ++r) rx[b].push_back(take(J.wave * wpb));
        // resident + gather: a wave's region of out_all is rewritten by the next pass only after its transfer has left
        std::vector<hipEvent_t> sent_wave(res && J.gather && R.rank != 0 ? max_waves : 0);
        for (auto &e : sent_wave) hcheck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
        // digest slots: one per own wave, one per (wave, peer) on the root; zeroed once, summed once after the loop
        const uint64_t n_own = max_waves, n_rx = (J.gather && R.rank == 0) ? max_waves * (uint64_t)(J.world - 1) : 0;
        uint64_t *d_own = take(n_own + 1), *d_rx = take(n_rx + 1);
        hcheck(hipMemsetAsync(d_own, 0, (n_own + 1) * 8, (hipStream_t)st), "memset");
        hcheck(hipMemsetAsync(d_rx, 0, (n_rx + 1) * 8, (hipStream_t)st), "memset");
        if (res && mine) check(fhe_fill_random(ctx, in, mine * 64 * 2, SEED, R.start * wpb, st), "fhe_fill_random");   // block g = splitmix64(seed ^ global index)
        if (res && mine) {                                                               // one untimed pass: first-use costs (plan constants, code objects) stay outside the clock
            for (uint64_t w = 0; w < n_waves; ++w) {
                const uint64_t b0 = w * J.wave, nb = b0 + J.wave <= mine ? J.wave : mine - b0;
                check(fhe_dct8x8_quant(ctx, plan, in + b0 * wpb, out_all + b0 * wpb, nb, scr, scr_bytes, st), "fhe_dct8x8_quant");
            }
        }
        check(fhe_stream_sync(st), "sync");
        arrived.fetch_add(1);
        while (arrived.load() < J.world) std::this_thread::yield();                      // every rank is set up: start the clock together
        const double t0 = now();
        for (int rep = 0; rep < J.reps; ++rep)
        for (uint64_t w = 0; w < max_waves; ++w) {
            const uint64_t it = (uint64_t)rep * max_waves + w;
            const int slot = (int)(it & 1);
            const uint64_t b0 = R.start + w * J.wave, nb = w < n_waves ? (b0 + J.wave <= R.end ? J.wave : R.end - b0) : 0;
            uint64_t *dst = res ? out_all + w * J.wave * wpb : out[slot];
            if (nb) {
                if (!res) {
                    if (it >= 2) hcheck(hipStreamWaitEvent((hipStream_t)st, sent[slot], 0), "wait(sent)");      // out[slot] has left for the root
                    check(fhe_fill_random(ctx, in, nb * 64 * 2, SEED, b0 * wpb, st), "fhe_fill_random");
                } else if (rep && !sent_wave.empty()) {
                    hcheck(hipStreamWaitEvent((hipStream_t)st, sent_wave[w], 0), "wait(sent)");
                }
                check(fhe_dct8x8_quant(ctx, plan, res ? in + w * J.wave * wpb : in, dst, nb, scr, scr_bytes, st), "fhe_dct8x8_quant");
                if (!res) digest_into(ctx, dst, nb * wpb, b0 * wpb, d_own + w, st);
                hcheck(hipEventRecord(computed[slot], (hipStream_t)st), "record");
            }
            if (!J.gather) continue;
            if (R.rank != 0) {
                if (nb) {                                                                // the transfer overlaps the next wave's compute
                    hcheck(hipStreamWaitEvent((hipStream_t)st_comm, computed[slot], 0), "wait(computed)");
                    ncheck(ncclSend(dst, nb * wpb, ncclUint64, 0, J.comms[R.rank], (hipStream_t)st_comm), "ncclSend");
                    hcheck(hipEventRecord(res ? sent_wave[w] : sent[slot], (hipStream_t)st_comm), "record");
                }
            } else {
                // wave w of every peer that has one, as ONE group: the transfers arrive concurrently, each over its peer's own link
                std::vector<std::pair<int, uint64_t>> got;
                if (it >= 2) hcheck(hipStreamWaitEvent((hipStream_t)st_comm, digested[slot], 0), "wait(digested)");     // rx[slot] has been read
                ncheck(ncclGroupStart(), "ncclGroupStart");
                for (int r = 1; r < J.world; ++r) {
                    const uint64_t done = w * J.wave;
                    if (done >= peer_blocks[r]) continue;
                    const uint64_t cnt = peer_blocks[r] - done < J.wave ? peer_blocks[r] - done : J.wave;
                    ncheck(ncclRecv(rx[slot][r - 1], cnt * wpb, ncclUint64, r, J.comms[0], (hipStream_t)st_comm), "ncclRecv");
                    got.push_back({r, cnt});
                }
                ncheck(ncclGroupEnd(), "ncclGroupEnd");
                hcheck(hipEventRecord(received[slot], (hipStream_t)st_comm), "record");
                hcheck(hipStreamWaitEvent((hipStream_t)st_dig, received[slot], 0), "wait(received)");
                for (auto &g : got) {
                    uint64_t s, e;
                    block_range(g.first, J.world, J.total, s, e);
                    digest_into(ctx, rx[slot][g.first - 1], g.second * wpb, (s + w * J.wave) * wpb, d_rx + w * (uint64_t)(J.world - 1) + (g.first - 1), st_dig);
                }
                hcheck(hipEventRecord(digested[slot], (hipStream_t)st_dig), "record");
            }
        }
        check(fhe_stream_sync(st), "sync");
        check(fhe_stream_sync(st_comm), "sync");
        check(fhe_stream_sync(st_dig), "sync");
        R.seconds = now() - t0;                                                           // the ONE host synchronisation of the loop
        if (res)                                                                          // resident: the outputs are still there -- digested after the clock stopped
            for (uint64_t w = 0; w < n_waves; ++w) {
                const uint64_t b0 = w * J.wave, nb = b0 + J.wave <= mine ? J.wave : mine - b0;
                digest_into(ctx, out_all + b0 * wpb, nb * wpb, (R.start + b0) * wpb, d_own + w, st);
            }
        R.digest = sum_slots(d_own, n_own, st);
        R.received_digest = sum_slots(d_rx, n_rx, st);
        for (void *p : owned) (void)fhe_dev_free(p);
    } catch (const Fail &f) {
        R.error = f.what;
        arrived.fetch_add(J.world);                                                       // release the others' start barrier
    }
    if (plan) fhe_dct_plan_destroy(plan);
    if (st) fhe_stream_destroy(st);
    if (st_comm) fhe_stream_destroy(st_comm);
    if (st_dig) fhe_stream_destroy(st_dig);
    if (ctx) fhe_ctx_destroy(ctx);                                                        // device buffers die with the process
}
}  // namespace

int main(int argc, char **argv) {
    Job J;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        std::fprintf(stderr, "multi_gpu_dct needs a HIP device (no CPU path exists)\n");
        return 2;
    }
    J.devices = ndev;
    J.total = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 512;
    J.world = argc > 2 ? std::atoi(argv[2]) : ndev;
    J.wave = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 64;
    const bool want_gather = argc > 4 ? std::atoi(argv[4]) != 0 : true;
    J.resident = argc > 5 && std::string(argv[5]) == "resident";
    J.reps = argc > 6 ? std::atoi(argv[6]) : 1;
    if (J.world < 1 || !J.wave || !J.total || J.reps < 1) return 2;
    J.gather = want_gather && J.world > 1 && J.world <= ndev;                             // RCCL: one rank per device
    const int kk = fhe_default_coeff_modulus(J.n, 0, J.q);
    if (kk < 1) return 2;
    J.k = (uint32_t)kk;
    J.words_per_block = (uint64_t)64 * 2 * J.k * J.n;
    int rc = 0;
    try {
        if (J.gather) {
            std::vector<int> devs(J.world);
            for (int r = 0; r < J.world; ++r) devs[r] = r;
            J.comms.resize(J.world);
            ncheck(ncclCommInitAll(J.comms.data(), J.world, devs.data()), "ncclCommInitAll");
        }
        std::vector<Rank> ranks(J.world);
        std::atomic<int> arrived{0};
        std::vector<std::thread> pool;
        for (int r = 0; r < J.world; ++r) {
            ranks[r].rank = r;
            ranks[r].device = r % ndev;
            block_range(r, J.world, J.total, ranks[r].start, ranks[r].end);
            pool.emplace_back(run_rank, std::cref(J), std::ref(ranks[r]), std::ref(arrived));
        }
        for (auto &t : pool) t.join();
        for (auto &c : J.comms) ncclCommDestroy(c);
        double slowest = 0;
        uint64_t sum = 0, peers = 0;
        for (auto &R : ranks) {
            if (!R.error.empty()) throw Fail{"rank " + std::to_string(R.rank) + ": " + R.error};
            if (R.seconds > slowest) slowest = R.seconds;
            sum += R.digest;
            if (R.rank) peers += R.digest;
        }
        // the reference point: ONE rank over all N blocks (rank 0's device, the verifying wave loop: two wave buffers whatever N is)
        Job one = J;
        one.world = 1;
        one.gather = false;
        one.resident = false;
        one.reps = 1;
        Rank solo;
        block_range(0, 1, J.total, solo.start, solo.end);
        std::atomic<int> a1{0};
        run_rank(one, solo, a1);
        if (!solo.error.empty()) throw Fail{"single-rank run: " + solo.error};
        double solo_rate = J.total / solo.seconds;
        if (J.resident) {                                  // the N = 1 figure of the weak-scaling pair: one rank, the per-rank share, the same resident loop
            Job share = J;
            share.world = 1;
            share.gather = false;
            share.total = ranks[0].end - ranks[0].start;
            Rank alone;
            alone.start = 0;
            alone.end = share.total;
            std::atomic<int> a2{0};
            run_rank(share, alone, a2);
            if (!alone.error.empty()) throw Fail{"single-rank resident run: " + alone.error};
            solo_rate = share.total * (double)J.reps / alone.seconds;
        }
        const double rate = J.total * (double)J.reps / slowest;
        const bool digests_ok = solo.digest == sum, gather_ok = !J.gather || ranks[0].received_digest == peers;
        std::printf("{\"workload\": \"homomorphic 8x8 DCT+quant, %llu blocks over %d ranks on %d device(s), C++ host over include/fhe_hip.h (n=%u, k=%u)\", "
                    "\"mode\": \"%s\", \"reps\": %d, "
                    "\"ranks\": %d, \"devices\": %d, \"wave_blocks\": %llu, \"seconds\": %.4f, \"blocks_per_s\": %.1f, \"single_rank_blocks_per_s\": %.1f, "
                    "\"single_rank_is\": \"%s\", \"efficiency_vs_single_rank\": %.4f, "
                    "\"output_digest\": \"%016llx\", \"single_rank_digest\": \"%016llx\", \"digests_equal\": %s, "
                    "\"gather\": \"%s\", \"gathered_digest_equals_senders\": %s}\n",
                    (unsigned long long)J.total, J.world, ndev, J.n, J.k,
                    J.resident ? "resident: inputs generated before the clock, outputs digested after it, only fhe_dct8x8_quant (+ transfers) timed" : "verify: fill + digest inside the loop",
                    J.reps, J.world, ndev, (unsigned long long)J.wave, slowest, rate, solo_rate,
                    J.resident ? "one rank over the per-rank share, same resident loop (weak scaling)" : "one rank over all blocks, verifying loop",
                    rate / (solo_rate * (J.resident ? J.world : 1)),
                    (unsigned long long)sum, (unsigned long long)solo.digest, digests_ok ? "true" : "false",
                    J.gather ? "rccl send/recv per wave to rank 0" : (want_gather && J.world > 1 ? "skipped: ranks share a device" : "none"),
                    J.gather ? (gather_ok ? "true" : "false") : "null");
        rc = digests_ok && gather_ok ? 0 : 1;
    } catch (const Fail &f) {
        std::fprintf(stderr, "multi_gpu_dct: %s\n", f.what.c_str());
        rc = 1;
    }
    return rc;
}
