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
    assert lib.fhe_abi_version() == fhe._lib.ABI_VERSION == 4
    assert re.search(r"#define FHE_ABI_VERSION 4\b", open(fhe.HEADER_PATH).read())


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


def test_resize_sample_plan_matches_the_reference_index_arithmetic(fhe):
    """fhe_resize_sample_plan (host only) against an independent float32 restatement of homo/fhe_resize.h:350-351,381-382
    (tests/refrun.sample_origins) and of GetPixelClamped / the tap order of SampleBicubic and SampleLinear"""
    from refrun import sample_origins
    for (W, H, w, h) in ((48, 48, 17, 17), (128, 128, 64, 64), (9, 7, 17, 13), (6, 5, 2, 2)):
        origins = sample_origins(W, H, w, h)
        for bicubic in (True, False):
            taps, fx, fy = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=bicubic)
            assert taps.shape == (w * h, 16 if bicubic else 4) and len(fx) == len(fy) == w * h
            offs = [(dx, dy) for dy in (-1, 0, 1, 2) for dx in (-1, 0, 1, 2)] if bicubic else [(0, 0), (1, 0), (0, 1), (1, 1)]
            for o in range(w * h):
                xi, yi = origins[o]
                want = [min(max(yi + dy, 0), H - 1) * W + min(max(xi + dx, 0), W - 1) for dx, dy in offs]
                assert list(map(int, taps[o])) == want, (W, H, w, h, bicubic, o)
                assert 0.0 <= fx[o] < 1.0 and 0.0 <= fy[o] < 1.0
            f32 = np.float32
            u = f32(f32(w - 1) / f32(w - 1) * f32(W)) - f32(0.5)
            assert fx[w - 1] == float(u - np.floor(u))
    lib = fhe._lib.load()
    assert lib.fhe_resize_sample_plan(4, 4, 1, 4, 1, None, None, None) < 0        # the reference divides by width - 1
    assert lib.fhe_approximated_step_out_size(12) == 22 and lib.fhe_approximated_step_out_size(0) == 3


def test_stream_record_io(fhe, tmp_path):
    """include/fhe_stream.h on the CPU: positional record reads / writes from threads and the mapped-file transfers
    produce and consume exactly the records the Python writer makes; foreign, mismatching and short streams are refused"""
    lib = fhe._lib.load()
    k, n, polys, cnt = 3, 1024, 2, 300
    rng = np.random.default_rng(1)
    data = rng.integers(0, 1 << 60, size=(cnt, polys, k, n), dtype=np.uint64)
    p1, p2, p3 = (str(tmp_path / x) for x in "abc")
    with open(p1, "wb") as f:
        for i in range(cnt):
            fhe.server.write_ciphertext(f, data[i])
    assert lib.fhe_io_record_bytes(polys, k, n) == os.path.getsize(p1) // cnt
    fd = os.open(p1, os.O_RDONLY)
    out = np.zeros_like(data)
    fhe._lib.call("fhe_io_read_records", fd, 0, cnt, polys, k, n, out.ctypes.data_as(C.c_void_p), 7)
    assert np.array_equal(out, data)
    part = np.zeros((13, polys, k, n), dtype=np.uint64)
    fhe._lib.call("fhe_io_read_records", fd, 250, 13, polys, k, n, part.ctypes.data_as(C.c_void_p), 3)
    assert np.array_equal(part, data[250:263])
    with pytest.raises(fhe.FheError, match="stream ended"):
        fhe._lib.call("fhe_io_read_records", fd, 290, 13, polys, k, n, part.ctypes.data_as(C.c_void_p), 3)
    with pytest.raises(fhe.FheError, match="does not match"):
        fhe._lib.call("fhe_io_read_records", fd, 0, 2, 3, k, n, part.ctypes.data_as(C.c_void_p), 1)
    os.close(fd)
    fd = os.open(p2, os.O_WRONLY | os.O_CREAT, 0o644)
    fhe._lib.call("fhe_io_write_records", fd, 0, cnt, polys, k, n, data.ctypes.data_as(C.c_void_p), 5)
    os.close(fd)
    assert open(p1, "rb").read() == open(p2, "rb").read()

    class Shape:
        pass
    ctx = Shape()
    ctx.k, ctx.n = k, n
    import torch
    t = torch.from_numpy(data.view(np.int64))
    w = fhe.server.StreamFile(p3, write=True, size=cnt * lib.fhe_io_record_bytes(polys, k, n))
    w.transfer(0, 100, polys, ctx, t, 5)
    w.transfer(100, 200, polys, ctx, t[100:], 7)
    w.close()
    assert open(p1, "rb").read() == open(p3, "rb").read()
    r = fhe.server.StreamFile(p1)
    back = torch.zeros_like(t)
    r.transfer(0, cnt, polys, ctx, back, 6)
    assert torch.equal(back, t)
    with pytest.raises(fhe.FheError, match="stream ended"):
        r.transfer(290, 20, polys, ctx, back, 2)
    r.close()
    open(p2, "r+b").write(b"NOTACIPH")
    bad = fhe.server.StreamFile(p2)
    with pytest.raises(fhe.FheError, match="not a ciphertext record"):
        bad.transfer(0, 1, polys, ctx, back, 1)
    bad.close()
