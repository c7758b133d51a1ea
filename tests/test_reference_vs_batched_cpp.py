"""GPU: the batched C++ host API (seal::hip::Circuits over include/fhe_circuits.h) against the REFERENCE's own circuit
functions run one ciphertext at a time through the facade, in ONE C++ process (oracle/ref_vs_batched_main.cpp):
Cubic (sizes 2, 4), Linear (2, 3), SampleBicubic / SampleLinear over an image, the shared-offset resize,
homomorphic_sin / cos, approximated_step and the per-channel loop of homo/server_decode.cpp:120-137 -- bit for bit."""
import os
import subprocess

import pytest

from refrun import ref_bin

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,env", [(4096, {}), (8192, {"FHE_SEAL23_MODULI": "1"}),
                                   # the RELINEARISED mode on both sides: the reference's unchanged functions under FHE_FACADE_RELIN against
                                   # seal::hip::Circuits built with the same keys (fhe_circuits_create_relin) -- size 2 everywhere, same bits
                                   (8192, {"FHE_SEAL23_MODULI": "1", "FHE_FACADE_RELIN": "30"}), (4096, {"FHE_FACADE_RELIN": "16"}),
                                   # the per-Cubic placement: the reference's unchanged Cubic / Linear + ONE facade relinearize (size 4 / 3 -> 2, keys for
                                   # s^2 and s^3) against seal::hip::Circuits(context, keys, 100, 100, per_cubic = true): samplers and shared resize too
                                   (8192, {"FHE_SEAL23_MODULI": "1", "FHE_XCHECK_PER_CUBIC": "30"}), (4096, {"FHE_SEAL23_MODULI": "1", "FHE_XCHECK_PER_CUBIC": "60"}),
                                   # the per-sample placement: the reference's UNCHANGED SampleBicubic / SampleLinear / Cubic / Linear (sizes up to 6) + ONE facade
                                   # relinearize of each result (keys for s^2 .. s^5) against seal::hip::Circuits(context, keys, 100, 100, FHE_RELIN_PER_SAMPLE)
                                   (8192, {"FHE_SEAL23_MODULI": "1", "FHE_XCHECK_PER_SAMPLE": "60"}), (8192, {"FHE_SEAL23_MODULI": "1", "FHE_XCHECK_PER_SAMPLE": "30"})])
def test_reference_functions_equal_batched_cpp_api(n, env):
    exe = ref_bin("ref_vs_batched", True)
    if not exe:
        pytest.skip("oracle/_ref/ref_vs_batched not built (needs /root/reference at build time)")
    r = subprocess.run([exe, str(n), str(1 << 14)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1800)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0 and "OK: reference circuits == batched C++ API" in r.stdout and "MISMATCH" not in r.stdout, tail
