"""GPU: the client halves of the JPEG pipeline in the product (client.py) against the reference's own client
(homo/client_jpeg.cpp compiled unchanged, oracle/_ref/ref_client_jpeg)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "oracle", "_ref", "ref_client_jpeg")


def _image(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([40 + 12 * xx, 200 - 20 * yy, 90 + 5 * xx + 7 * yy], axis=-1).astype(np.uint8)


def _load_key(fhe, path, polys, ctx):
    buf = np.zeros((polys, ctx.k, ctx.n), dtype=np.uint64)
    with open(path, "rb") as f:
        fhe.server.read_ciphertext_into(f, buf)
    return fhe.to_device(buf)


def test_receive_half_writes_the_same_jpeg_as_the_reference_client(fhe, tmp_path):
    """reference client --send -> streaming server (GPU) -> BOTH receive halves on the same stream and key:
    client.receive_jpeg must write the file homo/client_jpeg.cpp:196-292 writes, byte for byte (decrypt, decode,
    rounding, zig-zag, Huffman coding, headers)."""
    if not os.path.exists(CLIENT):
        pytest.skip("oracle/_ref/ref_client_jpeg not built (needs /root/reference at build time)")
    Image = pytest.importorskip("PIL.Image")
    (tmp_path / "keys").mkdir()
    (tmp_path / "image").mkdir()
    w, h = 16, 8
    Image.fromarray(_image(w, h), "RGB").save(str(tmp_path / "image" / "in.jpg"), quality=95, subsampling=0)
    par = ["--cmod", "4096", "--pmod", "3001"]

    def run(argv):
        r = subprocess.run(argv, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, " ".join(argv) + "\n" + r.stdout[-2000:] + r.stderr[-2000:]

    run([CLIENT, "--send", "-f", "image/in.jpg", "-c", "image/ct_in.txt"] + par)
    ctx = fhe.SEALContext(4096, fhe.PRESETS["P4096"]["q"], 3001)
    fhe.server.server_jpeg(ctx, str(tmp_path / "image" / "ct_in.txt"), str(tmp_path / "image" / "ct_out.txt"), 2, wave_blocks=2)
    run([CLIENT, "--recieve", "-f", "image/in.jpg", "-i", "image/ct_out.txt", "-o", "image/ref_out.jpg"] + par)
    sk = _load_key(fhe, str(tmp_path / "keys" / "seckey.txt"), 1, ctx)[0]
    enc = fhe.FractionalEncoder(ctx)
    fhe.client.receive_jpeg(ctx, fhe.Decryptor(ctx, sk), enc, str(tmp_path / "image" / "ct_out.txt"), w, h, str(tmp_path / "image" / "mine.jpg"))
    assert open(tmp_path / "image" / "mine.jpg", "rb").read() == open(tmp_path / "image" / "ref_out.jpg", "rb").read()


def test_send_and_receive_round_trip_through_the_streaming_server(fhe, tmp_path):
    """client.send_jpeg (keygen, encode, encrypt with the OS CSPRNG) -> server_jpeg -> client.receive_jpeg: the JPEG
    decodes to the input image (8x8 DCT with unit quantisation: only rounding and colour-conversion error)."""
    Image = pytest.importorskip("PIL.Image")
    ctx = fhe.SEALContext.preset("P4096")
    kg = fhe.KeyGenerator(ctx)
    enc = fhe.FractionalEncoder(ctx)
    w, h = 8, 8
    rgb = _image(w, h)
    n_blocks = fhe.client.send_jpeg(ctx, fhe.Encryptor(ctx, kg.public_key()), enc, rgb, str(tmp_path / "in.ct"))
    assert n_blocks == 1
    fhe.server.server_jpeg(ctx, str(tmp_path / "in.ct"), str(tmp_path / "out.ct"), n_blocks, wave_blocks=1)
    fhe.client.receive_jpeg(ctx, fhe.Decryptor(ctx, kg.secret_key()), enc, str(tmp_path / "out.ct"), w, h, str(tmp_path / "out.jpg"))
    got = np.asarray(Image.open(str(tmp_path / "out.jpg")).convert("RGB"))
    assert got.shape == rgb.shape
    assert fhe.client.rms_error(got, rgb) < 3.0
