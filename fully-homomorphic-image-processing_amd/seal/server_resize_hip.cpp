// server_resize_hip.cpp -- the batched STREAMING server_resize as a C++ host (no Python, no torch): homo/server_resize.cpp:127-146 +
// ResizeImage (homo/fhe_resize.h:308-392) over seal/hip_circuits.h, include/fhe_stream.h and the HIP runtime's streams / events.
// The reference keeps a sliding window of 2 / 4 source rows resident (:324-379), walks the destination rows serially and, per output
// pixel, encrypts frac(x) and frac(y) (:230,234,262,266) and evaluates SampleLinear / SampleBicubic on the three channels (:381-388).
// Here (the pipeline of fully-homomorphic-image-processing_amd/server.py server_resize, same window logic, same stream formats,
// byte-identical output for the same sampler key -- tests/test_gpu_server.py):
//
//   file -> pinned      reader thread: fhe_io_transfer of the next step's NEW source rows into a page-locked slot
//   pinned -> HBM       upload stream: the rows land in a ring of resident rows (row r in slot r % R); residues validated behind the copy
//   compute             per step of up to `rows_per_step` destination rows: ONE fhe_encrypt_batch for all fractions of the step
//                       (seal::hip::DeviceEncryptor), then per channel ONE fhe_sample_bicubic / fhe_sample_linear whose taps index the
//                       interleaved R, G, B records of the ring directly, and one gather into the output record order
//   HBM -> pinned       download stream
//   pinned -> file      writer thread: fhe_io_transfer into the mapped output stream
//
// The NEXT step's upload is enqueued before THIS step's download (both copy streams can share a hardware queue: DESIGN.md section 5).
//
// usage: server_resize_hip <in.ct> <out.ct> <public key file> <src_w> <src_h> <dst_w> <dst_h> <bicubic 0|1>
//                          [rows_per_step=4] [io_threads=16] [n=8192] [plain_modulus=16384] [sampler key: 64 hex digits, or -] [passes=1]
//                          [shared offsets 0|1 = 0]
//   shared = 1 (bicubic only; server.server_resize(shared_offsets=True)): ONE offset ciphertext per output column (once per job) and per
//   output row instead of two per output pixel, and the shared-offset circuit (fhe_resize_bicubic_shared_rows) per channel and step --
//   not the reference's ciphertexts, the same decrypted image.
//   The sampler key is for reproducible tests only (default: getrandom()).  passes > 1 repeats the job with the stream files left mapped
//   and the staging buffers locked (a long-lived server's steady state); the JSON line reports the last pass.  FHE_SEAL23_MODULI=1 selects SEAL 2.3.1's coefficient moduli.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <deque>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "fhe_stream.h"
#include "seal/hip_circuits.h"

using namespace seal;

namespace {
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Fail { std::string what; };
void check(int rc, const char *what) { if (rc < 0) throw Fail{std::string(what) + ": " + fhe_last_error()}; }
void hcheck(hipError_t e, const char *what) { if (e != hipSuccess) throw Fail{std::string(what) + ": " + hipGetErrorString(e)}; }

template <typename T> class Queue {
public:
    void put(T v) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(v); } cv_.notify_one(); }
    T get() { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return !q_.empty(); }); T v = q_.front(); q_.pop_front(); return v; }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
};
struct InSlot { int slot; bool used; };
struct Ready { long step; int slot; };
struct Write { long step; int slot; };
struct Step { uint32_t y0, y1; uint32_t lo, hi; uint32_t first, cnt; };      // destination rows [y0, y1), source rows needed [lo, hi), rows to bring in [first, first + cnt)
}  // namespace

int main(int argc, char **argv) {
    if (argc < 9) {
        std::fprintf(stderr, "usage: %s in.ct out.ct pubkey src_w src_h dst_w dst_h bicubic [rows_per_step=4] [io_threads=16] [n=8192] [plain_modulus=16384] [key hex | -] [passes=1] [shared=0] [evaluation keys | -] [relin placement 0|1|2]\n", argv[0]);
        return 2;
    }
    const char *in_path = argv[1], *out_path = argv[2], *pk_path = argv[3];
    const uint32_t W = (uint32_t)std::atoi(argv[4]), H = (uint32_t)std::atoi(argv[5]), w = (uint32_t)std::atoi(argv[6]), h = (uint32_t)std::atoi(argv[7]);
    const bool bicubic = std::atoi(argv[8]) != 0;
    const uint32_t rows_per_step = argc > 9 ? (uint32_t)std::atoi(argv[9]) : 4;
    const int io_threads = argc > 10 ? std::atoi(argv[10]) : 16;
    const int n_arg = argc > 11 ? std::atoi(argv[11]) : 8192;
    const uint64_t t = argc > 12 ? std::strtoull(argv[12], nullptr, 10) : 16384;
    const int passes = argc > 14 ? std::atoi(argv[14]) : 1;
    const bool shared = argc > 15 && std::atoi(argv[15]) != 0;
    // the relinearised modes (include/fhe_circuits.h): an evaluation-key file (seal::EvaluationKeys::save: what a client that honoured the reference's
    // parsed-and-unused --dbc would send) and where to relinearise: 0 after every product, 1 once per Cubic / Linear (keys s^2, s^3), 2 once per pixel (s^2 .. s^5)
    const char *evk_path = argc > 16 && std::strcmp(argv[16], "-") != 0 ? argv[16] : nullptr;
    const int placement = argc > 17 ? std::atoi(argv[17]) : 0;
    uint8_t key[32];
    const bool have_key = argc > 13 && std::strlen(argv[13]) == 64;
    for (int i = 0; have_key && i < 32; ++i) { unsigned v = 0; std::sscanf(argv[13] + 2 * i, "%2x", &v); key[i] = (uint8_t)v; }
    const uint32_t init_rows = bicubic ? 4 : 2;
    if (shared && !bicubic) { std::fprintf(stderr, "server_resize_hip: shared offsets exist for the bicubic sampler only\n"); return 2; }
    if (w < 2 || h < 2 || H < init_rows || W < 1 || !rows_per_step || io_threads < 1 || passes < 1) { std::fprintf(stderr, "server_resize_hip: image too small for the sampler\n"); return 2; }
    fhe_io_file *fin = nullptr, *fout = nullptr;
    int rc = 0;
    try {
        EncryptionParameters params;
        char poly_mod[32];
        std::snprintf(poly_mod, sizeof poly_mod, "1x^%i + 1", n_arg);
        params.set_poly_modulus(poly_mod);
        params.set_coeff_modulus(coeff_modulus_128(n_arg));
        params.set_plain_modulus(t);
        SEALContext context(params);
        const detail::CtxState &st = *context.state();
        const uint32_t k = st.k, n = st.n;
        PublicKey pk;
        {
            std::ifstream kf(pk_path, std::ios::binary);
            if (!kf) throw Fail{"cannot open the public key file"};
            pk.load(kf);
        }
        EvaluationKeys evk;
        if (evk_path) {
            std::ifstream ef(evk_path, std::ios::binary);
            if (!ef) throw Fail{"cannot open the evaluation key file"};
            evk.load(ef);
        }
        std::unique_ptr<hip::Circuits> circ_p(evk_path ? new hip::Circuits(context, evk, 100, 100, placement) : new hip::Circuits(context, 100, 100));
        hip::Circuits &circ = *circ_p;
        hip::DeviceEncryptor enc(context, pk, 100, 100, have_key ? key : nullptr, 0);
        const uint32_t so = circ.out_size(bicubic ? FHE_CIRC_SAMPLE_BICUBIC : FHE_CIRC_SAMPLE_LINEAR);
        const size_t ct_in = (size_t)2 * k * n, ct_out = (size_t)so * k * n;
        // ---- the reference's window arithmetic, in float like homo/fhe_resize.h:351-352,382 ----------------------------------
        std::vector<float> vs(h), us(w);
        std::vector<uint32_t> wstart(h);
        {
            int start = 0;
            for (uint32_t y = 0; y < h; ++y) {
                const float v = float(y) / float(h - 1) * float(H) - 0.5f;
                int ns = (int)v - (int)init_rows / 2 + 1;
                if (ns > (int)(H - init_rows)) ns = (int)(H - init_rows);
                if (ns > start) start = ns;                                // the window never moves backwards
                vs[y] = v;
                wstart[y] = (uint32_t)start;
            }
            for (uint32_t x = 0; x < w; ++x) us[x] = float(x) / float(w - 1) * float(W) - 0.5f;
        }
        std::vector<Step> steps;
        for (uint32_t y = 0; y < h;) {
            uint32_t e = y + 1;
            while (e < h && e - y < rows_per_step && wstart[e] + init_rows - wstart[y] <= init_rows + rows_per_step) ++e;
            steps.push_back(Step{y, e, wstart[y], wstart[e - 1] + init_rows, 0, 0});
            y = e;
        }
        uint32_t next_row = steps[0].lo, max_rows = 0, max_new = 0, max_px = 0;
        for (Step &s : steps) {
            s.first = std::max(next_row, s.lo);
            s.cnt = s.hi > s.first ? s.hi - s.first : 0;
            next_row = std::max(next_row, s.hi);
            max_rows = std::max(max_rows, s.hi - s.lo);
            max_new = std::max(max_new, s.cnt);
            max_px = std::max(max_px, (s.y1 - s.y0) * w);
        }
        const uint32_t R = 2 * max_rows + max_new + 1;                        // ring of resident source rows: row r lives in slot r % R
        const size_t row_words = (size_t)W * 3 * ct_in;
        hip::CiphertextBatch ring(context, (size_t)R * W * 3, 2);            // record (slot, x, channel) = pixel index (slot * W + x) * 3 + channel
        const int slots = 3;
        std::vector<uint64_t *> hin(slots), hout(slots);
        std::vector<hipEvent_t> ev_copied(slots), ev_landed(slots);
        for (int i = 0; i < slots; ++i) {
            hcheck(hipHostMalloc((void **)&hin[i], (size_t)std::max(max_new, 1u) * row_words * 8, hipHostMallocDefault), "hipHostMalloc");
            hcheck(hipHostMalloc((void **)&hout[i], (size_t)max_px * 3 * ct_out * 8, hipHostMallocDefault), "hipHostMalloc");
            hcheck(hipEventCreateWithFlags(&ev_copied[i], hipEventDisableTiming), "event");
            hcheck(hipEventCreateWithFlags(&ev_landed[i], hipEventDisableTiming), "event");
        }
        uint64_t *dout[2], *d_bad = nullptr;
        hipEvent_t ev_drained[2];
        bool drained_used[2] = {false, false};
        for (int d = 0; d < 2; ++d) {
            check(fhe_dev_alloc((size_t)max_px * 3 * ct_out * 8, (void **)&dout[d]), "fhe_dev_alloc");
            hcheck(hipEventCreateWithFlags(&ev_drained[d], hipEventDisableTiming), "event");
        }
        check(fhe_dev_alloc(8, (void **)&d_bad), "fhe_dev_alloc");
        std::vector<hipEvent_t> ev_computed(steps.size());
        for (auto &e : ev_computed) hcheck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
        hipStream_t main = nullptr, h2d = nullptr, d2h = nullptr;             // the circuits run on the default stream (seal/hip_circuits.h)
        hcheck(hipStreamCreateWithFlags(&h2d, hipStreamNonBlocking), "stream");
        hcheck(hipStreamCreateWithFlags(&d2h, hipStreamNonBlocking), "stream");
        const size_t rec_in = fhe_io_record_bytes(2, k, n), rec_out = fhe_io_record_bytes(so, k, n);
        check(fhe_io_open(in_path, 0, 0, &fin), "open input stream");
        if (fhe_io_size(fin) < (uint64_t)W * H * 3 * rec_in) throw Fail{"ciphertext stream ended"};
        check(fhe_io_open(out_path, 1, (uint64_t)w * h * 3 * rec_out, &fout), "open output stream");

        std::vector<double> secs;
        double read_s = 0, write_s = 0;
        auto run_once = [&]() {
        Queue<InSlot> free_in;
        Queue<Ready> ready_in;
        Queue<int> free_out;
        Queue<Write> to_write;
        for (int i = 0; i < slots; ++i) { free_in.put(InSlot{i, false}); free_out.put(i); }
        std::string reader_err, writer_err, err;
        read_s = write_s = 0;
        drained_used[0] = drained_used[1] = false;
        std::thread reader([&] {
            for (size_t si = 0; si < steps.size(); ++si) {
                InSlot s = free_in.get();
                if (s.slot < 0) return;
                if (s.used && hipEventSynchronize(ev_copied[s.slot]) != hipSuccess) { reader_err = "event sync"; break; }
                const double t0 = now();
                if (steps[si].cnt && fhe_io_transfer(fin, (uint64_t)steps[si].first * W * 3, (uint64_t)steps[si].cnt * W * 3, 2, k, n, hin[s.slot], (uint32_t)io_threads) < 0) {
                    reader_err = fhe_last_error();
                    break;
                }
                read_s += now() - t0;
                ready_in.put(Ready{(long)si, s.slot});
            }
            if (!reader_err.empty()) ready_in.put(Ready{-1, -1});
        });
        std::thread writer([&] {
            for (;;) {
                Write it = to_write.get();
                if (it.step < 0) return;
                if (hipEventSynchronize(ev_landed[it.slot]) != hipSuccess) { writer_err = "event sync"; free_out.put(-1); return; }
                const Step &s = steps[it.step];
                const double t0 = now();
                if (fhe_io_transfer(fout, (uint64_t)s.y0 * w * 3, (uint64_t)(s.y1 - s.y0) * w * 3, so, k, n, hout[it.slot], (uint32_t)io_threads) < 0) {
                    writer_err = fhe_last_error();
                    free_out.put(-1);
                    return;
                }
                write_s += now() - t0;
                free_out.put(it.slot);
            }
        });
        hcheck(hipMemset(d_bad, 0, 8), "memset");                                // synchronous: the first validation kernel runs on the upload stream
        const double t_start = now();
        try {
            auto overlaps = [&](const Step &a, const Step &prev) {           // do the ring slots of a's new rows hold rows the previous step reads?
                for (uint32_t i = 0; i < a.cnt; ++i)
                    for (uint32_t r = prev.lo; r < prev.hi; ++r)
                        if ((a.first + i) % R == r % R) return true;
                return false;
            };
            auto upload = [&](size_t si) {
                const Ready r = ready_in.get();
                if (r.step < 0) throw Fail{"reader: " + reader_err};
                const Step &s = steps[si];
                if (si >= 2) hcheck(hipStreamWaitEvent(h2d, ev_computed[si - 2], 0), "wait");     // the slots these rows overwrite were last read two steps ago ...
                if (si >= 1 && s.cnt && overlaps(s, steps[si - 1])) hcheck(hipStreamWaitEvent(h2d, ev_computed[si - 1], 0), "wait");   // ... unless the window jumped
                for (uint32_t done = 0; done < s.cnt;) {                                          // consecutive rows sit in consecutive slots modulo R
                    const uint32_t s0 = (s.first + done) % R, part = std::min(s.cnt - done, R - s0);
                    uint64_t *dst = ring.ptr() + (size_t)s0 * row_words;
                    hcheck(hipMemcpyAsync(dst, hin[r.slot] + (size_t)done * row_words, (size_t)part * row_words * 8, hipMemcpyHostToDevice, h2d), "h2d");
                    check(fhe_count_unreduced(st.h, dst, (uint64_t)part * W * 3 * 2, d_bad, h2d), "fhe_count_unreduced");   // a client's payload: what Ciphertext::load would reject
                    done += part;
                }
                hcheck(hipEventRecord(ev_copied[r.slot], h2d), "record");
                free_in.put(InSlot{r.slot, true});
                return r.slot;
            };
            const int offs_n = bicubic ? 16 : 4;
            hip::CiphertextBatch xcols;
            if (shared) {                                                                         // one offset ciphertext per output column, once per job
                std::vector<double> fx(w);
                for (uint32_t x = 0; x < w; ++x) fx[x] = (double)(us[x] - std::floor(us[x]));
                xcols = enc.encrypt_values(fx);
            }
            int next_slot = upload(0);
            std::vector<uint32_t> taps, taps_ch;
            std::vector<double> fracs;
            std::vector<const uint64_t *> src;
            for (size_t si = 0; si < steps.size(); ++si) {
                const Step &s = steps[si];
                const uint32_t npx = (s.y1 - s.y0) * w;
                const int d = (int)(si & 1);
                hcheck(hipStreamWaitEvent(main, ev_copied[next_slot], 0), "wait");
                if (shared) {
                    std::vector<double> fy(s.y1 - s.y0);
                    for (uint32_t yy = s.y0; yy < s.y1; ++yy) fy[yy - s.y0] = (double)(vs[yy] - std::floor(vs[yy]));
                    hip::CiphertextBatch yrows = enc.encrypt_values(fy);                          // row y of the job at position w + y of the key's stream
                    uint32_t s0 = 0, sc = 0;
                    check(fhe_resize_source_rows(H, h, s.y0, s.y1, 1, &s0, &sc), "fhe_resize_source_rows");
                    if (s0 < s.lo || s0 + sc > s.hi) throw Fail{"the rows the taps touch are not resident"};
                    if (drained_used[d]) hcheck(hipStreamWaitEvent(main, ev_drained[d], 0), "wait");
                    hip::CiphertextBatch chan(context, (size_t)sc * W, 2);
                    src.resize((size_t)sc * W);
                    for (uint32_t ch = 0; ch < 3; ++ch) {
                        for (uint32_t r = 0; r < sc; ++r)
                            for (uint32_t x = 0; x < W; ++x) src[(size_t)r * W + x] = ring.at(((size_t)((s0 + r) % R) * W + x) * 3 + ch);
                        check(fhe_gather(src.data(), (uint64_t)sc * W, ct_in, chan.ptr(), ct_in, main), "fhe_gather");         // this channel's resident rows, contiguous
                        hip::CiphertextBatch out = circ.resize_bicubic_rows(chan, W, H, w, h, s.y0, s.y1, s0, sc, xcols, yrows);
                        src.resize(std::max(src.size(), (size_t)npx));
                        for (uint32_t i = 0; i < npx; ++i) src[i] = out.at(i);
                        check(fhe_gather(src.data(), npx, ct_out, dout[d] + (size_t)ch * ct_out, 3 * ct_out, main), "fhe_gather");
                        src.resize((size_t)sc * W);
                    }
                } else {
                // sample plan of these destination rows in terms of ring slots; fractions in the reference's order (per pixel: x, then y)
                taps.assign((size_t)npx * offs_n, 0);
                fracs.assign((size_t)npx * 2, 0.0);
                size_t p = 0;
                for (uint32_t yy = s.y0; yy < s.y1; ++yy) {
                    const float v = vs[yy];
                    const int yi = (int)v;
                    for (uint32_t xx = 0; xx < w; ++xx, ++p) {
                        const float u = us[xx];
                        const int xi = (int)u;
                        int o = 0;
                        for (int dy = bicubic ? -1 : 0; dy <= (bicubic ? 2 : 1); ++dy)
                            for (int dx = bicubic ? -1 : 0; dx <= (bicubic ? 2 : 1); ++dx, ++o) {
                                const int cy = std::min(std::max(yi + dy, 0), (int)H - 1), cx = std::min(std::max(xi + dx, 0), (int)W - 1);
                                taps[p * offs_n + o] = (uint32_t)((((uint32_t)cy % R) * W + (uint32_t)cx) * 3);
                            }
                        fracs[2 * p] = (double)(u - std::floor(u));
                        fracs[2 * p + 1] = (double)(v - std::floor(v));
                    }
                }
                hip::CiphertextBatch fr = enc.encrypt_values(fracs);                              // ONE batch: 2 npx fresh encryptions
                hip::CiphertextBatch xf(context, npx, 2), yf(context, npx, 2);
                src.resize(npx);
                for (uint32_t i = 0; i < npx; ++i) src[i] = fr.at(2 * (size_t)i);
                check(fhe_gather(src.data(), npx, ct_in, xf.ptr(), ct_in, main), "fhe_gather");
                for (uint32_t i = 0; i < npx; ++i) src[i] = fr.at(2 * (size_t)i + 1);
                check(fhe_gather(src.data(), npx, ct_in, yf.ptr(), ct_in, main), "fhe_gather");
                if (drained_used[d]) hcheck(hipStreamWaitEvent(main, ev_drained[d], 0), "wait");  // dout[d] has left for the host
                taps_ch.resize(taps.size());
                for (uint32_t ch = 0; ch < 3; ++ch) {
                    for (size_t i = 0; i < taps.size(); ++i) taps_ch[i] = taps[i] + ch;
                    hip::CiphertextBatch out = bicubic ? circ.sample_bicubic(ring, taps_ch.data(), xf, yf) : circ.sample_linear(ring, taps_ch.data(), xf, yf);
                    for (uint32_t i = 0; i < npx; ++i) src[i] = out.at(i);
                    check(fhe_gather(src.data(), npx, ct_out, dout[d] + (size_t)ch * ct_out, 3 * ct_out, main), "fhe_gather");   // into the interleaved record order
                }
                }
                hcheck(hipEventRecord(ev_computed[si], main), "record");
                if (si + 1 < steps.size()) next_slot = upload(si + 1);                            // before this step's download
                const int oslot = free_out.get();
                if (oslot < 0) throw Fail{"writer: " + writer_err};
                hcheck(hipStreamWaitEvent(d2h, ev_computed[si], 0), "wait");
                hcheck(hipMemcpyAsync(hout[oslot], dout[d], (size_t)npx * 3 * ct_out * 8, hipMemcpyDeviceToHost, d2h), "d2h");
                hcheck(hipEventRecord(ev_landed[oslot], d2h), "record");
                hcheck(hipEventRecord(ev_drained[d], d2h), "record");
                drained_used[d] = true;
                to_write.put(Write{(long)si, oslot});
            }
        } catch (const Fail &f) {
            err = f.what;
        } catch (const std::exception &e) {
            err = e.what();
        }
        free_in.put(InSlot{-1, false});
        to_write.put(Write{-1, -1});
        reader.join();
        writer.join();
        (void)hipDeviceSynchronize();
        const double dt = now() - t_start;
        if (err.empty() && !writer_err.empty()) err = "writer: " + writer_err;
        if (err.empty() && !reader_err.empty()) err = "reader: " + reader_err;
        if (!err.empty()) throw Fail{err};
        uint64_t bad = 0;
        check(fhe_download(&bad, d_bad, 8, main), "fhe_download");
        check(fhe_stream_sync(main), "fhe_stream_sync");
        if (bad) throw Fail{"the input stream holds " + std::to_string(bad) + " residues that are not reduced modulo the coefficient moduli"};
        return dt;
        };
        for (int pass = 0; pass < passes; ++pass) secs.push_back(run_once());
        const double dt = secs.back();
        std::printf("{\"workload\": \"server_resize stream %ux%u -> %ux%u %s, three channels, C++ host over seal/hip_circuits.h + fhe_stream.h (n=%u, k=%u)\", "
                    "\"output_pixels\": %u, \"rows_per_step\": %u, \"steps\": %zu, \"passes\": %d, \"server_side_encryptions\": %llu, \"seconds\": %.4f, \"pixels_per_s\": %.1f, "
                    "\"stream_GB_per_s_in_plus_out\": %.2f, \"file_read_seconds\": %.4f, \"file_write_seconds\": %.4f}\n",
                    W, H, w, h, bicubic ? "bicubic" : "bilinear", n, k, w * h, rows_per_step, steps.size(), passes, (unsigned long long)enc.next_index() / (unsigned long long)passes, dt, w * h / dt,
                    ((double)W * H * 3 * rec_in + (double)w * h * 3 * rec_out) / dt / 1e9, read_s, write_s);
    } catch (const Fail &f) {
        std::fprintf(stderr, "server_resize_hip: %s\n", f.what.c_str());
        rc = 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "server_resize_hip: %s\n", e.what());
        rc = 1;
    }
    if (fin) fhe_io_close(fin);
    if (fout) fhe_io_close(fout);
    // a failed job must not leave a complete-looking output stream behind
    if (rc && fout && truncate(out_path, 0) != 0) std::fprintf(stderr, "server_resize_hip: could not truncate %s\n", out_path);
    return rc;
}
