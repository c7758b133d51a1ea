"""CPU: the SEAL-shaped facade (seal/seal.h) is a drop-in for the reference's circuit headers.

These tests only run where /root/reference exists (the build container); they never copy reference
text into the repository.
  * homo/fhe_image.h compiles AND links unchanged (that is how oracle/_ref/ref_jpeg_circuit is made);
  * homo/fhe_resize.h, homo/fhe_decode.h and all six mains pass `g++ -fsyntax-only` -- they also
    include OpenCV, which this image lacks, so a names-only stub (tests/stubs/) is on the include
    path for the syntax probe; that is not a build and nothing from it is executed.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PKG = os.path.join(ROOT, "fully-homomorphic-image-processing_amd")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "homo", "fhe_image.h")),
                                reason="reference tree not present (GPU box)")


def _syntax(src, extra=()):
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-w", "-Dlinux", "-I" + os.path.join(REF, "homo"),
           "-I" + os.path.join(REF, "include"), "-I" + PKG, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "tests", "stubs"), *extra, src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("hdr", ["fhe_image.h", "fhe_resize.h", "fhe_decode.h"])
def test_reference_circuit_headers_compile_unchanged(hdr, tmp_path):
    tu = tmp_path / "tu.cpp"
    tu.write_text('#include "%s"\nint main() { return 0; }\n' % hdr)
    _syntax(str(tu))


@pytest.mark.parametrize("src", ["server_jpeg.cpp", "server_resize.cpp", "server_decode.cpp",
                                 "client_jpeg.cpp", "client_resize.cpp", "client_decode.cpp"])
def test_reference_mains_compile_unchanged(src):
    _syntax(os.path.join(REF, "homo", src))


def test_reference_jpeg_circuit_links_against_facade():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_jpeg_circuit"))


@pytest.mark.parametrize("n", [4096, 8192])
def test_facade_self_test_on_oracle_backed_cabi(n):
    """seal/facade_test.cpp (keygen, encrypt, every Evaluator method, relinearize, save/load, fused
    DCT, decrypt) linked against oracle/cabi_on_oracle.c: the facade's host logic without a GPU."""
    exe = os.path.join(ROOT, "oracle", "facade_test_cpu")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "facade_test_cpu"])
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FACADE TEST OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
