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


@pytest.mark.parametrize("device", [False, True])
def test_send_and_receive_round_trip_through_the_streaming_server(fhe, tmp_path, device):
    """client.send_jpeg (keygen, encode, encrypt with the OS CSPRNG) -> server_jpeg -> client.receive_jpeg: the JPEG
    decodes to the input image (8x8 DCT with unit quantisation: only rounding and colour-conversion error)."""
    Image = pytest.importorskip("PIL.Image")
    ctx = fhe.SEALContext.preset("P4096")
    kg = fhe.KeyGenerator(ctx)
    enc = fhe.FractionalEncoder(ctx)
    w, h = 8, 8
    rgb = _image(w, h)
    encryptor = (fhe.DeviceEncryptor if device else fhe.Encryptor)(ctx, kg.public_key())          # device batches / the host sampler
    n_blocks = fhe.client.send_jpeg(ctx, encryptor, enc, rgb, str(tmp_path / "in.ct"))
    assert n_blocks == 1
    fhe.server.server_jpeg(ctx, str(tmp_path / "in.ct"), str(tmp_path / "out.ct"), n_blocks, wave_blocks=1)
    fhe.client.receive_jpeg(ctx, fhe.Decryptor(ctx, kg.secret_key()), enc, str(tmp_path / "out.ct"), w, h, str(tmp_path / "out.jpg"))
    got = np.asarray(Image.open(str(tmp_path / "out.jpg")).convert("RGB"))
    assert got.shape == rgb.shape
    assert fhe.client.rms_error(got, rgb) < 3.0


# ---------------------------------------------------------------------------------------------
# client halves of the resize and decode pipelines (homo/client_resize.cpp, homo/client_decode.cpp)
# ---------------------------------------------------------------------------------------------
RCLIENT = os.path.join(ROOT, "oracle", "_ref", "ref_client_resize")


@pytest.mark.parametrize("bicubic", [False, True])
def test_resize_client_halves_equal_the_reference_client(fhe, tmp_path, bicubic):
    """reference client_resize --send -> product server_resize (GPU) -> BOTH receiving clients on the same stream and key:
    client.receive_resize must produce the samples homo/client_resize.cpp:197-211 produces (its FractionalEncoder::decode
    values are logged by the test hook, then `int`, CLAMP, uint8_t).  And the product's own send half feeds the same
    server: decrypting its stream gives the image back."""
    if not os.path.exists(RCLIENT):
        pytest.skip("oracle/_ref/ref_client_resize not built (needs /root/reference at build time)")
    import struct
    Image = pytest.importorskip("PIL.Image")
    (tmp_path / "keys").mkdir()
    (tmp_path / "image").mkdir()
    W, H, w, h = 8, 6, 5, 4
    Image.fromarray(_image(W, H), "RGB").save(str(tmp_path / "image" / "in.jpg"), quality=95, subsampling=0)
    par = ["--width", str(w), "--height", str(h), "--cmod", "4096", "--pmod", "3001"]

    def run(argv, env=None):
        r = subprocess.run(argv, cwd=str(tmp_path), capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, " ".join(argv) + "\n" + r.stdout[-2000:] + r.stderr[-2000:]

    run([RCLIENT, "--send", "-f", "image/in.jpg", "-o", "image/ct_in.txt"] + par)
    ctx = fhe.SEALContext(4096, fhe.PRESETS["P4096"]["q"], 3001)
    pk = _load_key(fhe, str(tmp_path / "keys" / "pubkey.txt"), 2, ctx)
    sk = _load_key(fhe, str(tmp_path / "keys" / "seckey.txt"), 1, ctx)[0]
    enc = fhe.FractionalEncoder(ctx)
    fhe.server.server_resize(ctx, str(tmp_path / "image" / "ct_in.txt"), str(tmp_path / "image" / "ct_out.txt"), W, H, w, h, bicubic,
                             fhe.server.make_fraction_encryptor(ctx, pk, enc))
    run([RCLIENT, "--recieve", "-f", "image/in.jpg", "-c", "image/ct_out.txt", "-o", "image/ref_out.png"] + par,
        env={"FHE_DECODE_LOG_FILE": str(tmp_path / "decoded.f64")})
    raw = open(tmp_path / "decoded.f64", "rb").read()
    ref_decoded = struct.unpack("<%dd" % (len(raw) // 8), raw)
    assert len(ref_decoded) == w * h * 3
    mine_decoded = []
    img = fhe.client.receive_resize(ctx, fhe.Decryptor(ctx, sk), enc, str(tmp_path / "image" / "ct_out.txt"), w, h, decoded=mine_decoded)
    assert tuple(mine_decoded) == ref_decoded
    assert img.shape == (h, w, 3) and [int(x) for x in img.reshape(-1)] == [fhe.client.to_pixel(v) for v in ref_decoded]
    # the product's own sending half: same record order (RGB per pixel, row by row), decrypts to the image
    rgb = _image(W, H)
    assert fhe.client.send_resize(ctx, fhe.Encryptor(ctx, pk), enc, rgb, str(tmp_path / "image" / "mine_in.txt")) == (W, H)
    assert os.path.getsize(tmp_path / "image" / "mine_in.txt") == os.path.getsize(tmp_path / "image" / "ct_in.txt")
    back = fhe.client.receive_pixels(ctx, fhe.Decryptor(ctx, sk), enc, str(tmp_path / "image" / "mine_in.txt"), W, H)
    assert np.array_equal(back, rgb)
    # and in device batches (keys.DeviceEncryptor: fhe_frac_encode_batch + fhe_encrypt_batch): same stream shape, same image, fresh randomness
    first = open(tmp_path / "image" / "mine_in.txt", "rb").read()
    assert fhe.client.send_resize(ctx, fhe.DeviceEncryptor(ctx, pk), enc, rgb, str(tmp_path / "image" / "mine_in.txt")) == (W, H)
    second = open(tmp_path / "image" / "mine_in.txt", "rb").read()
    assert len(second) == len(first) and second != first
    assert np.array_equal(fhe.client.receive_pixels(ctx, fhe.Decryptor(ctx, sk), enc, str(tmp_path / "image" / "mine_in.txt"), W, H), rgb)


def test_decode_client_halves_and_run_length_pairs(fhe, tmp_path):
    """client.send_decode writes the run-length stream of homo/client_decode.cpp:122-153 (value, count per run, channel after
    channel; pairs per channel as in keys/params.txt); decrypting it gives the pairs back; to_pixel follows
    `int pixel = decode; CLAMP; (uint8_t)` including the unclamped cast."""
    ctx = fhe.SEALContext.preset("P4096")
    kg = fhe.KeyGenerator(ctx)
    enc = fhe.FractionalEncoder(ctx)
    rgb = np.zeros((2, 4, 3), dtype=np.uint8)
    rgb[..., 0] = [[5, 5, 5, 9], [9, 9, 2, 2]]
    rgb[..., 1] = 7
    rgb[..., 2] = [[1, 2, 3, 4], [5, 6, 7, 8]]
    assert fhe.client.run_length_pairs(rgb[..., 0].reshape(-1)) == [(5, 3), (9, 3), (2, 2)]
    w, h, pairs = fhe.client.send_decode(ctx, fhe.Encryptor(ctx, kg.public_key()), enc, rgb, str(tmp_path / "runs.ct"))
    assert (w, h, pairs) == (4, 2, [3, 1, 8])
    vals = []
    fhe.client.receive_pixels(ctx, fhe.Decryptor(ctx, kg.secret_key()), enc, str(tmp_path / "runs.ct"), sum(pairs) * 2 // 3, 1, decoded=vals)   # 24 records
    assert [round(v) for v in vals] == [5, 3, 9, 3, 2, 2, 7, 8] + [x for i in range(1, 9) for x in (i, 1)]
    assert fhe.client.to_pixel(254.99) == 254 and fhe.client.to_pixel(-3.2) == 0 and fhe.client.to_pixel(397.0) == 255
    assert fhe.client.to_pixel(397.0, clamp=False) == 397 - 256 and fhe.client.to_pixel(-142.0, clamp=False) == 256 - 142
    assert fhe.client.to_pixel(1e300) == 0                      # cvttsd2si saturates to INT_MIN, the clamp gives 0
