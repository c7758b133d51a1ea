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
    with ThreadPoolExecutor(len(sets)) as ex:
        got = list(ex.map(lambda t: run_resize_set("bilinear", 2048, t)[0], sets))
    assert got == [PUBLISHED_RESIZE[("bilinear", 2048, t)] for t in sets]


@pytest.mark.gpu
@pytest.mark.parametrize("inter,n,sets", [
    ("bilinear", 2048, [11, 31, 101]),
    ("bilinear", 4096, [11, 31, 101, 307, 1009, 3001, 10007, 30011, 100003]),
    ("bicubic", 4096, [31, 101, 307, 1009]),
    ("bicubic", 8192, [31, 3001, 100003]),
])
def test_product_reproduces_published_resize_rms_on_gpu(inter, n, sets):
    if not _have(""):
        pytest.skip("oracle/_ref/ref_*_resize not built (needs /root/reference at build time)")
    with ThreadPoolExecutor(3) as ex:
        got = list(ex.map(lambda t: run_resize_set(inter, n, t, gpu=True)[0], sets))
    assert got == [PUBLISHED_RESIZE[(inter, n, t)] for t in sets]


# The one deterministic entry that is NOT reproduced: bicubic at t = 11, where plaintext coefficients wrap
# modulo t.  The reference recorded 34.4 (n = 4096, 8192, 16384); the oracle, the GPU and an independent
# exact model of the plaintext ring (tools/plain_ring_model.py: no ciphertexts at all) all give 29.715.
# A correct BFV evaluation of the committed homo/fhe_resize.h must decrypt to the ring model's
# polynomials, so the difference is not in this library's arithmetic; its cause in the reference's SEAL
# 2.3 run is not known.  The bilinear entry at the same t (17.9597) and the wrapped
# JPEG entries (72.7491, 77.6639, 114.663, 35.672) do reproduce.
BICUBIC_T11_HERE = "29.715"


def test_plain_ring_model_agrees_with_published_and_documents_t11():
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "plain_ring_model.py")
    out = subprocess.run([sys.executable, tool, "11"], capture_output=True, text=True, check=True).stdout
    assert out.strip().endswith("RMSError 17.9597"), out
    out = subprocess.run([sys.executable, tool, "11", "bicubic"], capture_output=True, text=True, check=True).stdout
    assert out.strip().endswith("RMSError " + BICUBIC_T11_HERE), out
    assert PUBLISHED_RESIZE[("bicubic", 4096, 11)] == "34.4"


@pytest.mark.gpu
def test_product_bicubic_t11_equals_the_exact_ring_model_on_gpu():
    if not _have(""):
        pytest.skip("oracle/_ref/ref_*_resize not built (needs /root/reference at build time)")
    assert run_resize_set("bicubic", 4096, 11, gpu=True)[0] == BICUBIC_T11_HERE
