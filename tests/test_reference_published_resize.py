"""The reference's own golden outputs for the RESIZE pipeline: the `RMSError` lines of
benchmark/results.txt (benchmark/benchmark.py:18-29 on image/boazbarak.jpg 48x48 -> 17x17, bilinear and
bicubic, four poly degrees x nine plain moduli).  These are the only reference-held numbers that pass
through Evaluator::multiply / square (BEHZ): two ciphertext products per Linear, five per Cubic.

The reference's UNMODIFIED mains (homo/client_resize.cpp, homo/server_resize.cpp; oracle/Makefile target
`ref`, binaries in oracle/_ref/) run
  * on the CPU against the oracle (oracle/libfhe_cabi_oracle.so)   -> pins the ORACLE   (not gpu)
  * on the MI355X against libfhe_hip.so                            -> pins the PRODUCT  (gpu)
and must print exactly the published value.  The compare step needs cv::imread / cv::resize; OpenCV is not
in this image, so the validated stand-in tests/stubs/opencv2/opencv.hpp is used (tests/test_opencv_standin.py).
Entries whose value depends on the random noise (runs at the edge of the noise budget) are excluded, see
oracle/pin_against_reference.py.  The 113.692 entries (budget exhausted in the reference's run: every pixel
decoded to garbage and clamped to 0) are kept in the table but not run as pins: which pixels fail once the
budget is gone depends on an implementation's noise constants, not on its arithmetic (here the last row and
column of the bilinear image stay decodable at n = 2048, because Linear's cross terms cancel where both
taps clamp to the same pixel); the value itself is reproduced without any FHE in tests/test_opencv_standin.py."""
import os
import re
from concurrent.futures import ThreadPoolExecutor

import pytest

from oracle.pin_against_reference import PUBLISHED_RESIZE, ROOT, run_resize_set

NOISE_EDGE = {("bilinear", 2048, 307), ("bicubic", 4096, 3001), ("bicubic", 4096, 10007), ("bicubic", 4096, 30011)}


def _have(sfx):
    return all(os.path.exists(os.path.join(ROOT, "oracle", "_ref", b + sfx)) for b in ("ref_client_resize", "ref_server_resize"))


def test_published_resize_table_is_the_references():
    """guard the transcription when the reference tree is at hand: every deterministic entry equals the
    file, and the only entries left out are the four noise-edge ones"""
    path = "/root/reference/benchmark/results.txt"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    lines = open(path).read().splitlines()
    seen = {}
    for i, ln in enumerate(lines):
        m = re.match(r"\.\./logs/resize_boaz_(\w+)_17_17_(\d+)_(\d+)\.txt", ln)
        if m:
            rms = [x for x in lines[i + 1:i + 5] if x.startswith("RMSError,")][0].split(",")[1]
            seen[(m.group(1), int(m.group(2)), int(m.group(3)))] = rms
    assert len(seen) == 72
    assert {k: v for k, v in seen.items() if k not in NOISE_EDGE} == PUBLISHED_RESIZE
    assert set(seen) - set(PUBLISHED_RESIZE) == NOISE_EDGE


def test_oracle_reproduces_published_bilinear_rms_on_cpu():
    """n = 2048 (one 54-bit prime): t = 11, 31 and 101 -> 17.9597 through 5,202 BEHZ products each.
    About 40 s, three processes."""
    if not _have("_cpu"):
        pytest.skip("oracle/_ref/ref_*_resize_cpu not built (needs /root/reference at build time)")
    sets = [11, 31, 101]
    # + the RELINEARISED mode through the facade switch FHE_FACADE_RELIN=<dbc>: the reference's unchanged
    # server_resize, every product followed by a relinearisation with keys the facade derives from the secret key the server
    # loads (homo/server_resize.cpp:103-116) -- size-2 ciphertexts throughout, the same decrypted samples, the same RMSError
    # (dbc = 16: at n = 2048 the 54-bit modulus has no room for the key-switch noise of 30-bit digits)
    jobs = [(t, {}) for t in sets] + [(11, {"FHE_FACADE_RELIN": "16"})]
    with ThreadPoolExecutor(len(jobs)) as ex:
        got = list(ex.map(lambda j: run_resize_set("bilinear", 2048, j[0], env=j[1])[0], jobs))
    assert got == [PUBLISHED_RESIZE[("bilinear", 2048, t)] for t, _ in jobs]


@pytest.mark.gpu
@pytest.mark.parametrize("inter,n,sets", [
    ("bilinear", 2048, [11, 31, 101]),
    ("bilinear", 4096, [11, 31, 101, 307, 1009, 3001, 10007, 30011, 100003]),
    ("bicubic", 4096, [31, 101, 307, 1009]),
    ("bicubic", 8192, [31, 3001, 100003]),
    ("bilinear", 16384, [31, 100003]),               # the last column of the reference's grid (benchmark/benchmark.py:6): SEAL 2.3.1's eight
    ("bicubic", 16384, [31, 100003]),                # primes, 438 bits, through keygen / encrypt / BEHZ products / decrypt of the whole facade
])
def test_product_reproduces_published_resize_rms_on_gpu(inter, n, sets):
    if not _have(""):
        pytest.skip("oracle/_ref/ref_*_resize not built (needs /root/reference at build time)")
    with ThreadPoolExecutor(3) as ex:
        got = list(ex.map(lambda t: run_resize_set(inter, n, t, gpu=True)[0], sets))
    assert got == [PUBLISHED_RESIZE[(inter, n, t)] for t in sets]


@pytest.mark.gpu
@pytest.mark.parametrize("inter,n,t,dbc", [("bicubic", 4096, 101, 30), ("bilinear", 4096, 1009, 30), ("bicubic", 8192, 3001, 60)])
def test_reference_server_resize_in_the_relinearised_mode_on_gpu(inter, n, t, dbc, tmp_path):
    """FHE_FACADE_RELIN=<dbc>: the reference's UNCHANGED server_resize (homo/fhe_resize.h Cubic / Linear through seal::Evaluator)
    with every product relinearised by the facade (keys derived from the secret key the server loads); the client decrypts
    size-2 ciphertexts to the same samples: the published RMSError.  The facade's statistics show the relinearisations."""
    if not _have(""):
        pytest.skip("oracle/_ref/ref_*_resize not built (needs /root/reference at build time)")
    sf = str(tmp_path / "stats.txt")
    rms = run_resize_set(inter, n, t, gpu=True, env={"FHE_FACADE_RELIN": str(dbc), "FHE_FACADE_STATS": sf})[0]
    assert rms == PUBLISHED_RESIZE[(inter, n, t)]
    recorded = max(int(m) for m in re.findall(r"recorded=(\d+)", open(sf).read()))
    products = 867 * (25 if inter == "bicubic" else 6)                # 17 x 17 x 3 samples; Cubic: 5 products x 5 calls, Linear: 2 x 3
    plain = {"bicubic": 867 * 5 * 20, "bilinear": 867 * 3 * 5}[inter]  # the Evaluator calls of the reference's mode ...
    assert recorded >= plain + products                                # ... plus one relinearisation per product


# Bicubic at t = 11: plaintext coefficients wrap modulo t and 23 of the 867 decoded samples leave [0, 255]
# (-142 ... 397).  The committed client clamps them (`CLAMP(pixel, 0, 255)`, homo/client_resize.cpp:208) and prints
# 29.715; the SAME decoded samples cast to uint8_t without the clamp (i.e. modulo 256) give exactly the published
# 34.4 (n = 4096, 8192, 16384).  No other published resize entry has a sample outside [0, 255] -- the clamp is
# invisible in all of them -- so benchmark/results.txt was produced by a client without that line, and with the
# decoded samples in hand every deterministic entry of the table is reproduced.
BICUBIC_T11_CLAMPED = "29.715"


def _rms_of_decoded(decoded, conversion):
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import plain_ring_model as prm
    assert len(decoded) == 17 * 17 * 3
    img = np.array([prm.to_pixel(v, conversion) for v in decoded], dtype=np.int64).reshape(17, 17, 3)
    return prm.rms_string(img, prm.reference_image())


def test_plain_ring_model_reproduces_every_published_conversion():
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "plain_ring_model.py")
    run = lambda a: subprocess.run([sys.executable, tool] + list(a), capture_output=True, text=True, check=True).stdout.strip()
    with ThreadPoolExecutor(4) as ex:                                     # four independent processes
        r = list(ex.map(run, [("11",), ("11", "wrap"), ("11", "bicubic"), ("11", "bicubic", "wrap")]))
    assert r[0].endswith("RMSError 17.9597")
    assert r[1].endswith("RMSError 17.9597")                              # no sample leaves [0, 255]: the clamp is invisible
    assert r[2].endswith("RMSError " + BICUBIC_T11_CLAMPED)
    assert r[3].endswith("RMSError " + PUBLISHED_RESIZE[("bicubic", 4096, 11)])
    assert PUBLISHED_RESIZE[("bicubic", 4096, 11)] == "34.4"


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 8192])
def test_product_reproduces_published_bicubic_t11_on_gpu(n):
    """the reference's unmodified mains on the MI355X: the client prints the clamped figure, and the samples it
    decoded -- five BEHZ products per Cubic, size-6 ciphertexts, wrapped plaintext -- give the published 34.4
    under the unclamped cast"""
    if not _have(""):
        pytest.skip("oracle/_ref/ref_*_resize not built (needs /root/reference at build time)")
    decoded = []
    assert run_resize_set("bicubic", n, 11, gpu=True, decoded=decoded)[0] == BICUBIC_T11_CLAMPED
    assert _rms_of_decoded(decoded, "clamp") == BICUBIC_T11_CLAMPED
    assert _rms_of_decoded(decoded, "wrap") == PUBLISHED_RESIZE[("bicubic", n, 11)] == "34.4"


@pytest.mark.gpu
def test_unclamped_cast_changes_no_other_published_entry_on_gpu():
    """the decoded samples of a bilinear and a bicubic run with the noise budget intact stay inside [0, 255]"""
    if not _have(""):
        pytest.skip("oracle/_ref/ref_*_resize not built (needs /root/reference at build time)")
    for inter, n, t in (("bilinear", 4096, 11), ("bicubic", 4096, 101)):
        decoded = []
        assert run_resize_set(inter, n, t, gpu=True, decoded=decoded)[0] == PUBLISHED_RESIZE[(inter, n, t)]
        assert all(0 <= v < 256 for v in decoded)
        assert _rms_of_decoded(decoded, "wrap") == PUBLISHED_RESIZE[(inter, n, t)]


def test_oracle_reproduces_published_bicubic_rms_on_cpu():
    """n = 4096, t = 101 -> 19.8048 through 4,335 Cubic calls (21,675 BEHZ products, size-6 results) on the CPU ORACLE:
    about 7 minutes on one core, so it runs only when FHE_RUN_SLOW=1; the record of the last run is tracked
    (profiles/r03_oracle_pin_bicubic_n4096_t101.json) and checked against the published table here."""
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "r03_oracle_pin_bicubic_n4096_t101.json")))
    assert rec["match"] and rec["rms"] == PUBLISHED_RESIZE[("bicubic", rec["n"], rec["plain_modulus"])] == "19.8048" and rec["cubic_calls"] == 4335
    if os.environ.get("FHE_RUN_SLOW") != "1":
        pytest.skip("set FHE_RUN_SLOW=1 to re-run the 7-minute CPU pin (tracked record verified)")
    if not _have("_cpu"):
        pytest.skip("oracle/_ref/ref_*_resize_cpu not built (needs /root/reference at build time)")
    assert run_resize_set("bicubic", 4096, 101)[0] == "19.8048"
