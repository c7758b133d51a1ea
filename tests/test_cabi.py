"""CPU: the C-ABI shared library loads, exports every symbol include/*.h declares, its
host-only entry points work, and compute entry points fail loudly without a HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest


def _declared_symbols(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(fhe):
    lib = fhe._lib.load()
    import glob
    headers = sorted(glob.glob(os.path.join(os.path.dirname(fhe.HEADER_PATH), "*.h")))
    assert headers == sorted(fhe.HEADER_PATHS)
    names = sorted({n for h in headers for n in _declared_symbols(h)} - {"fhe_band_consumer"})
    assert len(names) >= 60
    for name in names:
        assert hasattr(lib, name), "libfhe_hip.so does not export %s" % name
    # and the Python binding table covers the whole header
    assert set(names) == set(fhe._lib.SIGNATURES), set(names) ^ set(fhe._lib.SIGNATURES)
    assert lib.fhe_abi_version() == 1


def test_no_oracle_dependency_in_product(fhe):
    """the shipped library and package never reference oracle/"""
    pkg = os.path.dirname(fhe.LIB_PATH)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(root, f)).read()
                for line in txt.splitlines():
                    code = line.split("//")[0].split("#")[0]
                    assert "fhe_oracle" not in code and "from oracle" not in code and "import oracle" not in code, (f, line)
    out = os.popen("ldd %s" % fhe.LIB_PATH).read()
    assert "oracle" not in out


def test_default_coeff_modulus(fhe):
    lib = fhe._lib.load()
    buf = (C.c_uint64 * 8)()
    assert lib.fhe_default_coeff_modulus(4096, 0, buf) == 3
    assert list(buf[:3]) == [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001]
    assert lib.fhe_default_coeff_modulus(8192, 1, buf) == 4
    assert lib.fhe_default_coeff_modulus(3000, 0, buf) < 0
    assert b"no default" in lib.fhe_last_error()


def test_host_encoder_matches_oracle_and_model(fhe, oracle_mod):
    from oracle import bigint_model as bm
    lib = fhe._lib.load()
    n, t = 4096, 1 << 14
    orc = oracle_mod.Oracle(n, [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], t)
    out = np.zeros(n, dtype=np.uint64)
    for v in [0.541196100, -1.847759065, 0.125, 128.0, 1 / 16.0, 1 / 99.0, -4.71238898038469, 0.0, -1.0, 255.0, 1e-9, 12345.678]:
        ln = lib.fhe_frac_encode(n, t, v, 100, 100, out.ctypes.data_as(C.c_void_p))
        assert ln >= 0
        assert np.array_equal(out, orc.encode(v)), v
        assert list(out) == bm.frac_encode(v, n, t), v
        assert ln == (int(np.nonzero(out)[0].max()) + 1 if out.any() else 0)
        dec = lib.fhe_frac_decode(n, t, out.ctypes.data_as(C.c_void_p), 100, 100)
        assert dec == orc.decode(out) == bm.frac_decode(list(map(int, out)), t)
        assert abs(dec - v) < 1e-12 * max(1.0, abs(v)) + 2.0 ** -100


def test_host_encoder_errors(fhe):
    lib = fhe._lib.load()
    out = np.zeros(128, dtype=np.uint64)
    assert lib.fhe_frac_encode(128, 1 << 14, 1.0, 100, 100, out.ctypes.data_as(C.c_void_p)) < 0   # 200 coefficients do not fit
    out = np.zeros(4096, dtype=np.uint64)
    assert lib.fhe_frac_encode(4096, 1 << 14, float("nan"), 100, 100, out.ctypes.data_as(C.c_void_p)) < 0
    assert lib.fhe_frac_encode(4096, 1 << 14, 2.0 ** 40, 10, 10, out.ctypes.data_as(C.c_void_p)) < 0  # integer part too wide


def test_compute_fails_loudly_without_gpu(fhe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    lib = fhe._lib.load()
    q = (C.c_uint64 * 3)(0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001)
    h = C.c_void_p()
    rc = lib.fhe_ctx_create(4096, q, 3, 1 << 14, 0, C.byref(h))
    assert rc == -2 and not h.value                      # FHE_ERR_HIP, no context
    assert b"no CPU fallback" in lib.fhe_last_error()
    with pytest.raises(RuntimeError):
        fhe.SEALContext.preset("P4096")


def test_ctx_create_rejects_bad_parameters(fhe):
    lib = fhe._lib.load()
    h = C.c_void_p()
    q = (C.c_uint64 * 3)(0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001)
    assert lib.fhe_ctx_create(4095, q, 3, 1 << 14, 0, C.byref(h)) == -1      # not a power of two
    assert lib.fhe_ctx_create(4096, q, 0, 1 << 14, 0, C.byref(h)) == -1      # no moduli
    assert lib.fhe_ctx_create(4096, q, 9, 1 << 14, 0, C.byref(h)) == -1      # too many
