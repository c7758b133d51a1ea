"""CPU: the HOST code under AddressSanitizer + UndefinedBehaviorSanitizer and ThreadSanitizer (SURVEY.md section 5 "Race
detection / sanitizers": "ASan/UBSan on CPU oracle tests"), plus a hypothesis-driven fuzz of the untrusted-input surface.

What is instrumented (oracle/Makefile targets `asan`, `tsan`; binaries in oracle/_san/): the oracle and the oracle-backed C ABI,
the header-only SEAL facade seal/seal.h (lazy expression graph, aliasing handles, auto-flush, stream records, failure paths),
the product's stream I/O unit csrc/stream_io.hip (host code, compiled as plain C++), the facade's self tests
(seal/facade_test.cpp, seal/facade_threads.cpp) and the reference's UNMODIFIED server mains on top of the facade
(homo/server_jpeg.cpp in both facade modes, homo/server_resize.cpp, homo/server_decode.cpp).  Any sanitizer report fails
the test: UBSan is built with -fno-sanitize-recover, ASan aborts on the first error, LeakSanitizer runs with the suppressions
of oracle/lsan.supp (the facade's device-buffer pool keeps its buffers until process exit on purpose), TSan with
halt_on_error.  The one disabled check (-fno-sanitize=return for ref_server_decode) is the reference's own undefined behaviour at
homo/fhe_decode.h:128-200; oracle/Makefile cites it.
"""
import os
import struct
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")
HAVE_REF = os.path.exists("/root/reference/homo/fhe_image.h")
ENV = {"LSAN_OPTIONS": "suppressions=%s:print_suppressions=0" % os.path.join(ROOT, "oracle", "lsan.supp"),
       "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1", "ASAN_OPTIONS": "abort_on_error=0:detect_leaks=1",
       "TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"}


@pytest.fixture(scope="module")
def san_bins():
    """build what is missing or stale (about a minute from scratch with -j8; seconds afterwards)"""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-j8", "asan", "tsan"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return SAN


def _run(argv, extra_env=None, timeout=900):
    r = subprocess.run(argv, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **ENV, **(extra_env or {})))
    report = [ln for ln in (r.stdout + r.stderr).splitlines() if "Sanitizer" in ln or "runtime error" in ln]
    assert r.returncode == 0 and not report, "%s\n%s\n%s" % (" ".join(argv), "\n".join(report[:20]), (r.stdout + r.stderr)[-3000:])
    return r.stdout


def test_facade_self_tests_under_asan_ubsan_and_tsan(san_bins):
    """every Evaluator operation, relinearize, save / load incl. the malformed-record rejections, the fused DCT helper
    (facade_test); shared Evaluator across threads, copies of pending values across threads, a flush that throws, a pending value
    that outlives its context (facade_threads) -- lazy and eager facade modes"""
    jobs = [([os.path.join(san_bins, "facade_test_cpu_asan")], {}),
            ([os.path.join(san_bins, "facade_test_cpu_asan")], {"FHE_FACADE_EAGER": "1"}),
            ([os.path.join(san_bins, "facade_threads_cpu_asan"), "4096", "4", "4"], {}),
            ([os.path.join(san_bins, "facade_threads_cpu_asan"), "4096", "4", "4"], {"FHE_FACADE_EAGER": "1"}),
            ([os.path.join(san_bins, "facade_threads_cpu_tsan"), "4096", "4", "4"], {}),
            ([os.path.join(san_bins, "facade_threads_cpu_tsan"), "4096", "4", "4"], {"FHE_FACADE_EAGER": "1"})]
    with ThreadPoolExecutor(3) as ex:
        outs = list(ex.map(lambda j: _run(j[0], j[1]), jobs))
    assert all("TEST OK" in o for o in outs)


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference at build time (oracle/_san/ref_server_*_cpu_asan)")
def test_reference_server_mains_under_asan_ubsan(san_bins, oracle_mod, tmp_path):
    """the reference's unmodified servers through the instrumented facade: server_jpeg in both facade modes and server_resize
    (bilinear) reproduce the published RMSError values, server_decode writes the oracle composition's bytes -- and no report"""
    sys.path.insert(0, ROOT)
    from oracle.pin_against_reference import PUBLISHED, PUBLISHED_RESIZE, run_resize_set, run_set
    from refrun import oracle_server_decode, parse_stream, run_server_decode

    def decode_job():
        n, t = 1024, 1 << 14
        orc = oracle_mod.Oracle(n, [0x3FFFFFFF000001], t)
        pairs, w, h, order, degree, delta = (1, 1, 0), 2, 1, 64, 1, 0.5
        runs = orc.random_ct(2 * sum(pairs), seed=31).reshape(sum(pairs), 2, 2, orc.k, orc.n)
        hook = orc.random_ct(sum(1 + w * h + p * w * h * degree * 2 for p in pairs), seed=32)
        os.environ["FHE_REF_VARIANT"] = "asan"
        try:
            raw = run_server_decode(str(tmp_path), orc, runs, pairs, w, h, hook, gpu=False, n_arg=n, order=order, degree=degree, delta=delta, env_extra=ENV)
        finally:
            os.environ.pop("FHE_REF_VARIANT", None)
        want = oracle_server_decode(orc, oracle_mod, runs, pairs, w, h, hook, order, degree, delta)
        got = parse_stream(raw, orc.k, orc.n)
        assert len(got) == 3 * w * h
        for i in range(w * h):
            for ch in range(3):                                            # interleaved save, homo/server_decode.cpp:139-143
                assert np.array_equal(got[i * 3 + ch], want[ch][i]), (i, ch)
        return "decode ok"

    jobs = [lambda: run_set(2048, 3001, variant="asan", env=ENV)[0],
            lambda: run_set(2048, 3001, variant="asan", env=dict(ENV, FHE_FACADE_EAGER="1"))[0],
            lambda: run_resize_set("bilinear", 2048, 11, variant="asan", env=ENV)[0],
            decode_job]
    with ThreadPoolExecutor(4) as ex:
        got = [f.result() for f in [ex.submit(j) for j in jobs]]
    assert got == [PUBLISHED[3001], PUBLISHED[3001], PUBLISHED_RESIZE[("bilinear", 2048, 11)], "decode ok"]


# ---- fuzz: record streams are untrusted input ------------------------------------------------------------------------------
Q54 = 0x3FFFFFFF000001           # coeff_modulus_128(1024): the context the harness creates


def _record(polys, k, n, rng, magic=b"FHEHIP1\0", reduce=True):
    body = rng.integers(0, Q54 if reduce else 1 << 63, size=polys * k * n, dtype=np.uint64)
    return struct.pack("<8sIIII", magic, polys, k, n, 0) + body.tobytes()


def _evk_record(dbc, digits, count, k, n, rng, magic=b"FHEHIPK\0", reduce=True, words=None):
    """seal::EvaluationKeys::save's record (seal/seal.h): magic, u32 dbc, digits, count, k, n, reserved, then count*k*digits*2*k*n words"""
    w = count * k * digits * 2 * k * n if words is None else words
    body = rng.integers(0, Q54 if reduce else 1 << 63, size=w, dtype=np.uint64)
    return struct.pack("<8sIIIIII", magic, dbc, digits, count, k, n, 0) + body.tobytes()


def _bundle(cases):
    out = [struct.pack("<I", len(cases))]
    for (polys, k, n, first, count, threads, data) in cases:
        out.append(struct.pack("<IIIIIII", polys, k, n, first, count, threads, len(data)))
        out.append(data)
    return b"".join(out)


def test_stream_loaders_fuzz_under_asan(san_bins, tmp_path):
    """hypothesis mutates well-formed record streams (bit flips, truncations, header fields, splices) and draws the shape /
    range arguments of the transfers independently of the bytes; seal::Ciphertext / PublicKey / SecretKey::load and
    fhe_io_open + fhe_io_transfer + fhe_io_read_records must accept or reject every case without a sanitizer report, accept the
    untouched stream, and reject every truncation"""
    from hypothesis import HealthCheck, given, seed, settings, strategies as st
    exe = os.path.join(san_bins, "stream_fuzz_cpu_asan")
    rng = np.random.default_rng(7)
    good2 = _record(2, 1, 1024, rng) + _record(2, 1, 1024, rng) + _record(2, 1, 1024, rng)
    good1 = _record(1, 1, 1024, rng)
    rec2 = len(good2) // 3
    goodk = _evk_record(16, 4, 1, 1, 1024, rng)               # one 54-bit prime at dbc 16: four digits
    bundles = []

    mutation = st.one_of(
        st.tuples(st.just("flip"), st.integers(0, 10 ** 9), st.integers(0, 7)),
        st.tuples(st.just("trunc"), st.integers(0, 10 ** 9), st.just(0)),
        st.tuples(st.just("field"), st.integers(0, 4), st.integers(0, 2 ** 32 - 1)),
        st.tuples(st.just("splice"), st.integers(0, 10 ** 9), st.integers(0, 64)),
        st.tuples(st.just("none"), st.just(0), st.just(0)))
    case = st.tuples(st.sampled_from([0, 1, 2]), st.lists(mutation, min_size=0, max_size=3),
                     st.integers(0, 5), st.integers(0, 3), st.sampled_from([0, 1, 512, 1024, 1000, 2048, 4096]),      # polys, k, n of the transfer
                     st.integers(0, 6), st.integers(0, 6), st.integers(0, 5))                                       # first record, count, threads

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), database=None, derandomize=True)
    @given(st.lists(case, min_size=8, max_size=24))
    def collect(cases):
        out = []
        for which, muts, polys, k, n, first, count, threads in cases:
            data = bytearray((good2, good1, goodk)[which])
            for kind, a, b in muts:
                if kind == "flip" and data:
                    data[a % len(data)] ^= 1 << b
                elif kind == "trunc":
                    data = data[:a % (len(data) + 1)]
                elif kind == "field" and len(data) >= 24:
                    struct.pack_into("<I", data, 8 + 4 * (a % (6 if which == 2 else 4)), b)
                elif kind == "splice" and data:
                    at = a % len(data)
                    data[at:at] = bytes(b)
            out.append((polys, k, n, first, count, threads, bytes(data)))
        bundles.append(out)
    collect()
    # fixed cases with known verdicts, first in every run
    fixed = [(2, 1, 1024, 0, 3, 2, good2), (2, 1, 1024, 1, 2, 1, good2), (1, 1, 1024, 0, 1, 1, good1),
             (2, 1, 1024, 0, 3, 2, good2[:-8]), (2, 1, 1024, 0, 4, 1, good2), (2, 1, 1024, 0xFFFFFFFF, 2, 1, good2),
             (2, 1, 1024, 0, 3, 3, good2[:rec2] + _record(2, 1, 1024, rng, magic=b"NOTHIP1\0") + good2[2 * rec2:]),
             (2, 1, 1024, 0, 1, 1, _record(2, 1, 1024, rng, reduce=False)), (2, 1, 1024, 0, 1, 1, b""),
             # evaluation-key streams (seal::EvaluationKeys::load + Evaluator::relinearize of a size-3 ciphertext with what loaded)
             (0, 0, 0, 0, 0, 0, goodk), (0, 0, 0, 0, 0, 0, goodk[:-8]), (0, 0, 0, 0, 0, 0, _evk_record(16, 4, 1, 1, 1024, rng, reduce=False)),
             (0, 0, 0, 0, 0, 0, _evk_record(16, 1, 1, 1, 1024, rng)),               # self-consistent, but ONE digit cannot hold 54 bits at dbc 16
             (0, 0, 0, 0, 0, 0, _evk_record(60, 1, 1, 1, 2048, rng)),               # another degree than the Evaluator's context
             (0, 0, 0, 0, 0, 0, _evk_record(1, 61, 62, 1, 1024, rng, words=4096)),  # a header worth 62 MB on 32 KB of payload
             (0, 0, 0, 0, 0, 0, _evk_record(16, 4, 1, 16, 16384, rng, words=16)),   # (k, n) no context of the process has
             (0, 0, 0, 0, 0, 0, _evk_record(0, 4, 1, 1, 1024, rng)), (0, 0, 0, 0, 0, 0, _evk_record(16, 4, 0, 1, 1024, rng, words=8))]
    total = 0
    for i, cases in enumerate([fixed] + bundles):
        bpath, scratch = str(tmp_path / ("bundle%d.bin" % i)), str(tmp_path / "scratch.bin")
        open(bpath, "wb").write(_bundle(cases))
        out = _run([exe, bpath, scratch], timeout=600)
        lines = [ln for ln in out.splitlines() if ln.startswith("case ")]
        assert len(lines) == len(cases) and "FUZZ BUNDLE DONE" in out
        total += len(lines)
        if i == 0:
            v = [dict(kv.split("=") for kv in ln.split()[2:]) for ln in lines]
            assert v[0] == {"load": "1", "pk": "1", "sk": "0", "evk": "0", "transfer": "0", "read": "0"}      # three size-2 records: the first loads as a ciphertext or a public key
            assert v[1]["transfer"] == "0" and v[1]["read"] == "0"
            assert v[2] == {"load": "1", "pk": "0", "sk": "1", "evk": "0", "transfer": "0", "read": "0"}
            assert v[3]["transfer"] == "-1" and v[3]["read"] == "-1" and v[3]["load"] == "1"      # the LAST record is truncated: the first still loads
            assert v[4]["transfer"] == "-1" and v[4]["read"] == "-1"                              # more records than the stream holds
            assert v[5]["transfer"] == "-1" and v[5]["read"] == "-1"                              # a record number that would wrap the byte offset
            assert v[6]["transfer"] == "-1" and v[6]["read"] == "-1"                              # a foreign record in the middle
            assert v[7]["load"] == "0" and v[7]["transfer"] == "0"                                # unreduced residues: load() checks payloads, the raw transfer does not (fhe_count_unreduced does)
            assert v[8] == {"load": "0", "pk": "0", "sk": "0", "evk": "0", "transfer": "-1", "read": "-1"}
            assert [x["evk"] for x in v[:8]] == ["0"] * 8                                         # no ciphertext / key record passes for evaluation keys
            assert v[9]["evk"] == "2" and v[9]["load"] == "0"                                     # loads AND relinearises
            assert v[10]["evk"] == "0" and v[11]["evk"] == "0"                                    # truncated; unreduced residues
            assert v[12]["evk"] == "1" and v[13]["evk"] == "1"                                    # load, but relinearize refuses them on this context
            assert [x["evk"] for x in v[14:18]] == ["0"] * 4
    assert total >= 300
