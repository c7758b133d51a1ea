"""GPU: the streaming server_jpeg loop (homo/server_jpeg.cpp:109-153) over a ciphertext stream file,
and the bilinear sampler (homo/fhe_resize.h:222-252)."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SMALL = dict(n=1024, q=[0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], t=1 << 14)


def _ctx(fhe, om):
    return fhe.SEALContext(SMALL["n"], SMALL["q"], SMALL["t"]), om.Oracle(SMALL["n"], SMALL["q"], SMALL["t"])


@pytest.mark.parametrize("wave_blocks", [2, 8])
def test_server_jpeg_stream(fhe, oracle_mod, tmp_path, wave_blocks):
    ctx, orc = _ctx(fhe, oracle_mod)
    n_blocks = 3
    cts = orc.random_ct(n_blocks * 3 * 64, seed=4711).reshape(n_blocks, 3, 64, 2, orc.k, orc.n)
    fin, fout = tmp_path / "in.ct", tmp_path / "out.ct"
    with open(fin, "wb") as f:
        for b in range(n_blocks):
            for ch in range(3):
                for i in range(64):
                    fhe.server.write_ciphertext(f, cts[b, ch, i])
    assert fhe.server.server_jpeg(ctx, str(fin), str(fout), n_blocks, wave_blocks=wave_blocks) == n_blocks
    out = np.zeros((n_blocks, 3, 64, 2, orc.k, orc.n), dtype=np.uint64)
    with open(fout, "rb") as f:
        for b in range(n_blocks):
            for ch in range(3):              # the order homo/client_jpeg.cpp:266-271 reads: for k<3, for j<64
                for i in range(64):
                    fhe.server.read_ciphertext_into(f, out[b, ch, i])
        assert f.read(1) == b""
    for b in (0, n_blocks - 1):
        ycc = np.zeros((3, 64, 2, orc.k, orc.n), dtype=np.uint64)
        for i in range(64):
            y, u, v = orc.rgb_to_ycc(cts[b, 0, i], cts[b, 1, i], cts[b, 2, i])
            ycc[0, i], ycc[1, i], ycc[2, i] = y, u, v
        for ch in range(3):
            ref = orc.encrypted_dct(ycc[ch])
            assert np.array_equal(out[b, ch], ref), (b, ch)


def test_server_jpeg_stream_n4096_fused_fp64_kernels(fhe, oracle_mod, tmp_path):
    """one colour block at the BASELINE parameter set: the streaming loop over the FP64 kernels
    (k_rgb2ycc_f64, k_dct_rows/cols) with quantisation, against the oracle op by op"""
    ctx = fhe.SEALContext.preset("P4096")
    orc = oracle_mod.Oracle.preset("P4096")
    cts = orc.random_ct(3 * 64, seed=99).reshape(3, 64, 2, orc.k, orc.n)
    fin, fout = tmp_path / "in.ct", tmp_path / "out.ct"
    with open(fin, "wb") as f:
        for ch in range(3):
            for i in range(64):
                fhe.server.write_ciphertext(f, cts[ch, i])
    assert fhe.server.server_jpeg(ctx, str(fin), str(fout), 1, quant=list(fhe.YQT)) == 1
    out = np.zeros((3, 64, 2, orc.k, orc.n), dtype=np.uint64)
    with open(fout, "rb") as f:
        for ch in range(3):
            for i in range(64):
                fhe.server.read_ciphertext_into(f, out[ch, i])
    ycc = np.zeros((3, 64, 2, orc.k, orc.n), dtype=np.uint64)
    for i in range(64):
        ycc[0, i], ycc[1, i], ycc[2, i] = orc.rgb_to_ycc(cts[0, i], cts[1, i], cts[2, i])
    for ch in range(3):
        assert np.array_equal(out[ch], orc.dct_quant(ycc[ch], oracle_mod.YQT)), ch


def test_stream_rejects_foreign_data(fhe, tmp_path):
    buf = np.zeros((2, 3, 16), dtype=np.uint64)
    with pytest.raises(ValueError):
        fhe.server.read_ciphertext_into(io.BytesIO(b"NOTACIPHERTEXT" + bytes(64)), buf)
    with pytest.raises(EOFError):
        fhe.server.read_ciphertext_into(io.BytesIO(b""), buf)


def test_sample_linear_vs_oracle(fhe, oracle_mod):
    """bilinear: three Linear calls per pixel (size 2 -> 3 -> 4), batched, vs oracle Linear"""
    ctx, orc = _ctx(fhe, oracle_mod)
    ev = fhe.Evaluator(ctx)
    pc = fhe.circuits.PlainCache(ctx)
    W = H = 4
    pix = ctx.random_ct(W * H, size=2, seed=31337)
    taps, fx, fy = fhe.circuits.resize_sample_plan(W, H, 3, 3, bicubic=False)
    xf, yf = ctx.random_ct(9, size=2, seed=1), ctx.random_ct(9, size=2, seed=2)
    out = fhe.to_host(fhe.circuits.sample_linear(ev, pc, pix, taps, xf, yf))
    assert out.shape[1] == 4
    hp, hx, hy = fhe.to_host(pix), fhe.to_host(xf), fhe.to_host(yf)
    for o in (0, 4, 8):
        p00, p10, p01, p11 = (hp[i] for i in taps[o])
        col0, col1 = orc.linear(p00, p10, hx[o]), orc.linear(p01, p11, hx[o])
        assert np.array_equal(out[o], orc.linear(col0, col1, hy[o]))


def test_end_to_end_image_through_streaming_server(fhe, oracle_mod, tmp_path):
    """BASELINE.json configs[0] shape on a synthetic image: client encrypts a 16x16 RGB image pixel by
    pixel (homo/client_jpeg.cpp:129-165, block order of split_image_eight_block, homo/fhe_image.h:108-124),
    the GPU server runs rgb_to_ycc_fhe + encrypted_dct over the ciphertext stream, the client decrypts
    and decodes; the result must equal the plaintext pipeline (colour transform + dct() of
    homo/fhe_image.h:400-484) -- the quantity behind the reference's RMSError line."""
    from oracle import bigint_model as bm
    p = oracle_mod.PRESETS["P4096"]
    ctx, orc = fhe.SEALContext(p["n"], p["q"], p["t"]), oracle_mod.Oracle(p["n"], p["q"], p["t"])
    sk, pk = orc.keygen(2026)
    W = H = 16
    img = np.array([[[(31 * x + 17 * y + 101 * c) % 256 for c in range(3)] for x in range(W)] for y in range(H)], dtype=np.float64)
    blocks = []                                   # per 8x8 block: [channel][64] pixel values, row-major inside the block
    for j in range(0, H, 8):
        for i in range(0, W, 8):
            blocks.append([[img[j + k, i + l, c] for k in range(8) for l in range(8)] for c in range(3)])
    fin, fout = tmp_path / "image.ct", tmp_path / "result.ct"
    seed = 0
    with open(fin, "wb") as f:
        for b in blocks:
            for c in range(3):
                for v in b[c]:
                    seed += 1
                    fhe.server.write_ciphertext(f, orc.encrypt(pk, orc.encode(v), seed=seed))
    fhe.server.server_jpeg(ctx, str(fin), str(fout), len(blocks), wave_blocks=3)
    ct = np.zeros((2, orc.k, orc.n), dtype=np.uint64)
    worst, min_budget = 0.0, 1 << 30
    with open(fout, "rb") as f:
        for b in blocks:
            r, g, bl = (np.array(b[c]) for c in range(3))
            ycc = [0.299 * r + 0.587 * g + 0.114 * bl - 128.0, -0.168736 * r - 0.331264 * g + 0.5 * bl, 0.5 * r - 0.418688 * g - 0.081312 * bl]
            expect = [bm.plain_dct(list(ch)) for ch in ycc]
            for c in range(3):
                for i in range(64):
                    fhe.server.read_ciphertext_into(f, ct)
                    if i % 9 == 0:                              # decrypt a sample of the 768 outputs
                        plain, budget = orc.decrypt(sk, ct)
                        worst = max(worst, abs(orc.decode(plain) - expect[c][i]))
                        min_budget = min(min_budget, budget)
    assert min_budget > 0
    assert worst < 1e-6, worst      # decode is exact up to double rounding of the fractional digits


def test_streaming_server_output_feeds_the_reference_client(fhe, tmp_path):
    """The record order contract end to end: the reference's own client (homo/client_jpeg.cpp, compiled
    unchanged against the facade) encrypts the benchmark image, THIS package's streaming server
    (server.server_jpeg, fused kernels) processes the ciphertext stream, and the reference client's
    --recieve half (decrypt, zig-zag, Huffman, compare with jo_jpeg) must print the RMSError the
    reference recorded for that parameter set (benchmark/results.txt: 1.71767 at t = 3001).  A wrong
    channel/coefficient order in the output stream (homo/server_jpeg.cpp:146-153) gives a garbage JPEG."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    client = os.path.join(root, "oracle", "_ref", "ref_client_jpeg")
    if not os.path.exists(client):
        pytest.skip("oracle/_ref/ref_client_jpeg not built (needs /root/reference at build time)")
    (tmp_path / "keys").mkdir()
    (tmp_path / "image").mkdir()
    shutil.copy(os.path.join(root, "tests", "golden", "boazbarak.jpg"), str(tmp_path / "image" / "in.jpg"))
    par = ["--cmod", "4096", "--pmod", "3001"]

    def run(argv):
        r = subprocess.run(argv, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, " ".join(argv) + "\n" + r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout

    run([client, "--send", "-f", "image/in.jpg", "-c", "image/ct_in.txt"] + par)
    ctx = fhe.SEALContext(4096, fhe.PRESETS["P4096"]["q"], 3001)
    n_blocks = (48 // 8) * (48 // 8)
    assert fhe.server.server_jpeg(ctx, str(tmp_path / "image" / "ct_in.txt"), str(tmp_path / "image" / "ct_out.txt"), n_blocks, wave_blocks=8) == n_blocks
    out = run([client, "--recieve", "-f", "image/in.jpg", "-i", "image/ct_out.txt", "-o", "image/out.jpg"] + par)
    line = [ln for ln in out.splitlines() if ln.startswith("RMSError,")]
    assert line and line[0].split(",")[1] == "1.71767", out[-500:]
