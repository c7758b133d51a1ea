"""GPU: the C++ SEAL facade end to end, and the REFERENCE's own circuit code run through it."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACADE_TEST = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "seal", "facade_test")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_jpeg_circuit")


@pytest.mark.parametrize("n", [4096, 8192])
def test_facade_self_test(n):
    assert os.path.exists(FACADE_TEST), "build it with __graft_entry__.build()"
    r = subprocess.run([FACADE_TEST, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FACADE TEST OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("env", [{}, {"FHE_FACADE_EAGER": "1"}])
def test_facade_from_several_threads_on_the_device(env):
    """seal/facade_threads.cpp (the program the CPU suite runs under ASan / TSan on the oracle-backed ABI) on the MI355X: two contexts on
    two threads, one Evaluator and one Encryptor shared by four threads (every encryption its own index of the object's sampler key),
    copies of a pending value, a failing flush, a ciphertext that outlives its context"""
    exe = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "seal", "facade_threads")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    r = subprocess.run([exe, "4096", "4", "4"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    assert r.returncode == 0 and "FACADE THREADS TEST OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_reference_circuit_code_through_facade_equals_oracle(oracle_mod, tmp_path):
    """encrypted_dct, quantize_fhe and rgb_to_ycc_fhe exactly as written in the reference's
    homo/fhe_image.h (compiled unchanged against seal/seal.h in the build container) produce, on the
    GPU, the same bytes as the CPU oracle's restatement."""
    if not os.path.exists(REF_BIN):
        pytest.skip("oracle/_ref/ref_jpeg_circuit not built (needs /root/reference at build time)")
    orc = oracle_mod.Oracle.preset("P4096")
    cts = orc.random_ct(67, seed=oracle_mod.SEED)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    cts.tofile(str(fin))
    r = subprocess.run([REF_BIN, "4096", str(fin), str(fout)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = np.fromfile(str(fout), dtype=np.uint64).reshape(cts.shape)
    assert np.array_equal(out[:64], orc.dct_quant(cts[:64], oracle_mod.YQT))
    y, u, v = orc.rgb_to_ycc(cts[64], cts[65], cts[66])
    assert np.array_equal(out[64], y) and np.array_equal(out[65], u) and np.array_equal(out[66], v)
    assert "," in r.stdout          # the reference's own timing prints (homo/fhe_image.h:286)


def test_reference_cli_pipeline_unmodified(tmp_path):
    """BASELINE.json configs[0] end to end with the reference's OWN mains: homo/client_jpeg.cpp and
    homo/server_jpeg.cpp compiled unchanged against the facade (oracle/Makefile, target `ref`).
    client --send (keygen + encrypt every pixel) -> server (rgb_to_ycc_fhe + encrypted_dct on the GPU)
    -> client --recieve (decrypt, quantise, Huffman-code, write the JPEG), then the reference's own
    check (homo/fhe_image.h:508-521): RMS error of the FHE-produced JPEG against jo_jpeg's."""
    client = os.path.join(ROOT, "oracle", "_ref", "ref_client_jpeg")
    server = os.path.join(ROOT, "oracle", "_ref", "ref_server_jpeg")
    if not (os.path.exists(client) and os.path.exists(server)):
        pytest.skip("oracle/_ref/ref_client_jpeg / ref_server_jpeg not built (needs /root/reference at build time)")
    Image = pytest.importorskip("PIL.Image")
    (tmp_path / "keys").mkdir()
    (tmp_path / "image").mkdir()
    w, h = 16, 8                                       # 2 blocks x 3 channels x 64 ciphertexts
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([40 + 12 * xx, 200 - 20 * yy, 90 + 5 * xx + 7 * yy], axis=-1).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(str(tmp_path / "image" / "in.jpg"), quality=95, subsampling=0)

    def run(argv):
        r = subprocess.run(argv, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, " ".join(argv) + "\n" + r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout

    out = run([client, "--send", "-f", "image/in.jpg", "-c", "image/ct_in.txt", "--cmod", "4096"])
    assert out.startswith("Encryption,")
    out = run([server, "-f", "image/ct_in.txt", "-o", "image/ct_out.txt", "--cmod", "4096"])
    assert out.count("DCT,") == (w // 8) * (h // 8)
    out = run([client, "--recieve", "-f", "image/in.jpg", "-i", "image/ct_out.txt", "-o", "image/out.jpg",
               "--cmod", "4096"])
    line = [ln for ln in out.splitlines() if ln.startswith("RMSError,")]
    assert line, out[-2000:]
    rms = float(line[0].split(",")[1])
    # the FHE JPEG and jo_jpeg's differ only where a coefficient sits on a rounding boundary
    assert rms < 6.0, rms
    got = np.asarray(Image.open(str(tmp_path / "image" / "out.jpg")).convert("RGB"), dtype=np.int32)
    assert got.shape == (h, w, 3)
    assert np.sqrt(np.mean((got - rgb.astype(np.int32)) ** 2)) < 12.0


def test_reference_quantize_fhe_known_answer_through_decrypt(oracle_mod, tmp_path):
    """quantize_fhe sits outside every published pipeline (the reference's server never calls it).  Here the reference's OWN
    encrypted_dct + quantize_fhe code (homo/fhe_image.h:196-305, unchanged, through the facade on the GPU) runs on real
    encryptions of a pixel block; the decrypted, decoded outputs must be the plaintext DCT of homo/fhe_image.h:400-484 divided
    by the quantisation table -- a known answer for the product of the two circuits, independent of the oracle's restatement."""
    if not os.path.exists(REF_BIN):
        pytest.skip("oracle/_ref/ref_jpeg_circuit not built (needs /root/reference at build time)")
    from oracle import bigint_model as bm
    orc = oracle_mod.Oracle.preset("P4096")
    sk, pk = orc.keygen(42)
    vals = [float((37 * x + 101 * y + 13) % 256) - 128.0 for y in range(8) for x in range(8)]
    cts = np.stack([orc.encrypt(pk, orc.encode(v), seed=2000 + i) for i, v in enumerate(vals)] +
                   [orc.encrypt(pk, orc.encode(float(v)), seed=3000 + i) for i, v in enumerate((200, 100, 50))])
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    cts.tofile(str(fin))
    r = subprocess.run([REF_BIN, "4096", str(fin), str(fout)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = np.fromfile(str(fout), dtype=np.uint64).reshape(cts.shape)
    expect = bm.plain_dct(vals)
    for i in range(64):
        plain, budget = orc.decrypt(sk, out[i])
        assert budget > 0
        assert abs(orc.decode(plain) - expect[i] / oracle_mod.YQT[i]) < 1e-6, i
    # and rgb_to_ycc_fhe on the three extra ciphertexts (homo/fhe_image.h:310-325)
    R, G, B = 200.0, 100.0, 50.0
    want = (0.299 * R + 0.587 * G + 0.114 * B - 128.0, -0.168736 * R - 0.331264 * G + 0.5 * B, 0.5 * R - 0.418688 * G - 0.081312 * B)
    for j in range(3):
        plain, budget = orc.decrypt(sk, out[64 + j])
        assert budget > 0 and abs(orc.decode(plain) - want[j]) < 1e-6, j
