"""The REFERENCE's own resize and decode circuit code -- homo/fhe_resize.h:143-392 (Cubic, Linear,
GetPixelClamped, SampleLinear, SampleBicubic, ResizeImage with its sliding row window) through the
unchanged homo/server_resize.cpp main, and homo/fhe_decode.h:48-242 (homomorphic_sin, homomorphic_cos,
approximated_step) through oracle/ref_decode_circuit_main.cpp -- compiled unchanged against seal/seal.h
and run
  * on the CPU against the oracle-backed C ABI   -> must equal the oracle's restatement  (not gpu)
  * on the MI355X against libfhe_hip.so          -> must equal the oracle AND circuits.py (gpu)
bit for bit.  The circuits' server-side encryptions are supplied through
oracle/ref_hook.cpp so that both sides see the same ciphertexts.  The binaries exist only when
/root/reference was present at build time (oracle/Makefile target `ref`); they travel to the GPU box."""
import numpy as np
import pytest

from refrun import oracle_sample, ref_bin, run_decode_circuit, run_server_resize, sample_origins


def _need(name, gpu):
    if not ref_bin(name, gpu):
        pytest.skip("oracle/_ref/%s not built (needs /root/reference at build time)" % name)


# ----------------------------------------------------------------------------------------------
# CPU: the oracle is what the reference's code computes through the facade
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bicubic", [False, True])
def test_reference_resize_code_on_oracle_backend(oracle_mod, tmp_path, bicubic):
    _need("ref_server_resize", False)
    orc = oracle_mod.Oracle.preset("SEAL23_2048")
    W, H, w, h = 6, 5, 4, 3
    pix = orc.random_ct(W * H * 3, seed=11).reshape(W * H, 3, 2, orc.k, orc.n)
    fr = orc.random_ct(w * h * 2, seed=12)
    out = run_server_resize(str(tmp_path), orc, pix, W, H, w, h, bicubic, fr, gpu=False, n_arg=2048)
    for o, (xi, yi) in enumerate(sample_origins(W, H, w, h)):
        for ch in range(3):
            assert np.array_equal(out[o * 3 + ch], oracle_sample(orc, pix, W, H, xi, yi, ch, fr[2 * o], fr[2 * o + 1], bicubic)), (o, ch)


def test_reference_decode_code_on_oracle_backend(oracle_mod, tmp_path):
    _need("ref_decode_circuit", False)
    orc = oracle_mod.Oracle.preset("SEAL23_2048")
    x, zero = orc.random_ct(1, seed=5)[0], orc.random_ct(1, seed=6)
    for mode, fn in (("sin", oracle_mod.oracle_homomorphic_sin), ("cos", oracle_mod.oracle_homomorphic_cos)):
        got, = run_decode_circuit(str(tmp_path), orc, mode, x, zero, gpu=False, n_arg=2048)
        assert np.array_equal(got, fn(orc, x, zero[0])), mode
    npos, deg = 3, 2
    run_in = orc.random_ct(3, seed=9)
    zs = orc.random_ct(npos * deg * 2, seed=77)          # call order: for i: for j: sin's Enc(0), cos's Enc(0)
    got = run_decode_circuit(str(tmp_path), orc, "step", run_in, zs, gpu=False, n_arg=2048, extra=(64, deg, 0.5, npos, 1), sizes=(22,) * npos)
    ref = oracle_mod.oracle_approximated_step(orc, run_in[0], run_in[1], run_in[2], 64, deg, 0.5, npos, 1,
                                              lambda i, j, which: zs[(i * deg + j - 1) * 2 + (which == "cos")])
    for i in range(npos):
        assert np.array_equal(got[i], ref[i]), i


# ----------------------------------------------------------------------------------------------
# GPU: the product under the reference's own code == oracle == circuits.py
# ----------------------------------------------------------------------------------------------
GPU_SETS = [("P4096", 4096, {}), ("P8192", 8192, {"FHE_SEAL23_MODULI": "1"})]


@pytest.mark.gpu
@pytest.mark.parametrize("preset,n_arg,env", GPU_SETS)
@pytest.mark.parametrize("bicubic", [False, True])
def test_reference_resize_code_on_gpu(fhe, oracle_mod, tmp_path, preset, n_arg, env, bicubic):
    _need("ref_server_resize", True)
    orc = oracle_mod.Oracle.preset(preset)
    ctx = fhe.SEALContext.preset(preset)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    W, H, w, h = 6, 5, 4, 3
    pix = orc.random_ct(W * H * 3, seed=11).reshape(W * H, 3, 2, orc.k, orc.n)
    fr = orc.random_ct(w * h * 2, seed=12)
    out = run_server_resize(str(tmp_path), orc, pix, W, H, w, h, bicubic, fr, gpu=True, n_arg=n_arg, env_extra=env)
    origins = sample_origins(W, H, w, h)
    for o in (0, 5, w * h - 1):                                              # vs the CPU oracle
        xi, yi = origins[o]
        for ch in range(3):
            assert np.array_equal(out[o * 3 + ch], oracle_sample(orc, pix, W, H, xi, yi, ch, fr[2 * o], fr[2 * o + 1], bicubic)), (o, ch)
    # vs the product's batched restatement (circuits.py), every output pixel
    taps, _, _ = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=bicubic)
    xf, yf = fhe.to_device(fr[0::2]), fhe.to_device(fr[1::2])
    sampler = fhe.circuits.sample_bicubic if bicubic else fhe.circuits.sample_linear
    for ch in range(3):
        mine = fhe.to_host(sampler(ev, pc, fhe.to_device(np.ascontiguousarray(pix[:, ch])), taps, xf, yf))
        assert np.array_equal(mine, out[ch::3]), ch


@pytest.mark.gpu
@pytest.mark.parametrize("preset,n_arg,env", GPU_SETS)
def test_reference_decode_code_on_gpu(fhe, oracle_mod, tmp_path, preset, n_arg, env):
    _need("ref_decode_circuit", True)
    orc = oracle_mod.Oracle.preset(preset)
    ctx = fhe.SEALContext.preset(preset)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    x, zero = orc.random_ct(1, seed=5), orc.random_ct(1, seed=6)
    for mode, fn, mine in (("sin", oracle_mod.oracle_homomorphic_sin, fhe.circuits.homomorphic_sin),
                           ("cos", oracle_mod.oracle_homomorphic_cos, fhe.circuits.homomorphic_cos)):
        got, = run_decode_circuit(str(tmp_path), orc, mode, x[0], zero, gpu=True, n_arg=n_arg, env_extra=env)
        assert np.array_equal(got, fn(orc, x[0], zero[0])), mode
        assert np.array_equal(got, fhe.to_host(mine(ev, pc, fhe.to_device(x), fhe.to_device(zero)))[0]), mode
    npos, deg = 3, 2
    run_in = orc.random_ct(3, seed=9)
    zs = orc.random_ct(npos * deg * 2, seed=77)
    got = run_decode_circuit(str(tmp_path), orc, "step", run_in, zs, gpu=True, n_arg=n_arg, extra=(64, deg, 0.5, npos, 1), env_extra=env,
                             sizes=(22,) * npos)
    pick = lambda i, j, which: zs[(i * deg + j - 1) * 2 + (which == "cos")]
    ref = oracle_mod.oracle_approximated_step(orc, run_in[0], run_in[1], run_in[2], 64, deg, 0.5, npos, 1, pick)
    mine = fhe.circuits.approximated_step(ev, pc, *(fhe.to_device(run_in[i:i + 1]) for i in range(3)), order=64, degree=deg, delta=0.5,
                                          width=npos, height=1, zeros=lambda i, j, which: fhe.to_device(pick(i, j, which)[None]))
    for i in range(npos):
        assert np.array_equal(got[i], ref[i]), i
        assert np.array_equal(got[i], fhe.to_host(mine[i])[0]), i
