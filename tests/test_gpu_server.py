"""GPU: the streaming server_jpeg loop (homo/server_jpeg.cpp:109-153) over a ciphertext stream file,
and the bilinear sampler (homo/fhe_resize.h:222-252)."""
import io
import os
import shutil
import subprocess

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


# ---------------------------------------------------------------------------------------------
# server_resize: ResizeImage's sliding row window over a ciphertext stream (homo/fhe_resize.h:308-392)
# ---------------------------------------------------------------------------------------------
def _fraction_bank(fhe, bank):
    """encrypt_fractions stand-in: hands out pre-made ciphertexts in call order (xfract, yfract per pixel)"""
    pos = [0]

    def encrypt(values):
        out = bank[pos[0]:pos[0] + len(values)]
        assert len(out) == len(values)
        pos[0] += len(values)
        return fhe.to_device(np.ascontiguousarray(out))
    return encrypt


@pytest.mark.parametrize("bicubic,rows_per_step", [(True, 3), (False, 1), (True, 8)])
def test_server_resize_stream_16x16_to_8x8_vs_oracle(fhe, oracle_mod, tmp_path, bicubic, rows_per_step):
    from refrun import oracle_sample, read_records, sample_origins, write_record
    ctx, orc = _ctx(fhe, oracle_mod)
    W = H = 16
    w = h = 8
    pix = orc.random_ct(W * H * 3, seed=21).reshape(W * H, 3, 2, orc.k, orc.n)
    fin, fout = tmp_path / "in.ct", tmp_path / "out.ct"
    with open(fin, "wb") as f:
        for p in range(W * H):
            for c in range(3):
                write_record(f, pix[p, c])
    bank = orc.random_ct(w * h * 2, seed=22)
    n = fhe.server.server_resize(ctx, str(fin), str(fout), W, H, w, h, bicubic, _fraction_bank(fhe, bank), rows_per_step=rows_per_step)
    assert n == w * h
    out = read_records(str(fout), 6 if bicubic else 4, orc.k, orc.n, w * h * 3)
    origins = sample_origins(W, H, w, h)
    for o in range(0, w * h, 5 if bicubic else 3):
        xi, yi = origins[o]
        for ch in range(3):
            assert np.array_equal(out[o * 3 + ch], oracle_sample(orc, pix, W, H, xi, yi, ch, bank[2 * o], bank[2 * o + 1], bicubic)), (o, ch)


@pytest.mark.parametrize("bicubic", [False, True])
def test_server_resize_stream_equals_the_reference_server_byte_for_byte(fhe, oracle_mod, tmp_path, bicubic):
    """the same ciphertext stream through homo/server_resize.cpp (unchanged, facade, op at a time) and through
    server.server_resize (batched circuits): identical output files"""
    from refrun import ref_bin, run_server_resize
    if not ref_bin("ref_server_resize", True):
        pytest.skip("oracle/_ref/ref_server_resize not built (needs /root/reference at build time)")
    orc = oracle_mod.Oracle.preset("P4096")
    ctx = fhe.SEALContext.preset("P4096")
    W, H, w, h = 7, 9, 5, 6
    pix = orc.random_ct(W * H * 3, seed=31).reshape(W * H, 3, 2, orc.k, orc.n)
    bank = orc.random_ct(w * h * 2, seed=32)
    ref = run_server_resize(str(tmp_path), orc, pix, W, H, w, h, bicubic, bank, gpu=True, n_arg=4096)
    ref_bytes = open(tmp_path / "image" / "out.ct", "rb").read()
    mine = tmp_path / "mine.ct"
    fhe.server.server_resize(ctx, str(tmp_path / "image" / "in.ct"), str(mine), W, H, w, h, bicubic, _fraction_bank(fhe, bank), rows_per_step=2)
    assert open(mine, "rb").read() == ref_bytes
    assert ref.shape[0] == w * h * 3


def test_server_resize_with_real_encryptions_decrypts_to_the_plain_sampler(fhe, oracle_mod, tmp_path):
    """keys + fresh server-side encryptions of the fractions (make_fraction_encryptor, OS CSPRNG); decrypt and
    compare with the closed form of Linear (homo/fhe_resize.h:191-204)"""
    from refrun import read_records, sample_origins, write_record
    p = oracle_mod.PRESETS["P4096"]
    ctx, orc = fhe.SEALContext(p["n"], p["q"], p["t"]), oracle_mod.Oracle(p["n"], p["q"], p["t"])
    sk, pk = orc.keygen(5)
    W, H, w, h = 5, 4, 3, 3
    vals = np.array([[(37 * x + 11 * y + 50 * c) % 256 for c in range(3)] for y in range(H) for x in range(W)], dtype=np.float64)
    fin, fout = tmp_path / "in.ct", tmp_path / "out.ct"
    with open(fin, "wb") as f:
        for i in range(W * H):
            for c in range(3):
                write_record(f, orc.encrypt(pk, orc.encode(vals[i, c]), seed=100 + 3 * i + c))
    enc = fhe.server.make_fraction_encryptor(ctx, fhe.to_device(pk))
    fhe.server.server_resize(ctx, str(fin), str(fout), W, H, w, h, False, enc, rows_per_step=2)
    out = read_records(str(fout), 4, orc.k, orc.n, w * h * 3)
    f32 = np.float32
    for o, (xi, yi) in enumerate(sample_origins(W, H, w, h)):
        y, x = divmod(o, w)
        u = f32(f32(x) / f32(w - 1) * f32(W)) - f32(0.5)
        v = f32(f32(y) / f32(h - 1) * f32(H)) - f32(0.5)
        fx, fy = float(u - f32(np.floor(u))), float(v - f32(np.floor(v)))
        P = lambda dx, dy, c: vals[min(max(yi + dy, 0), H - 1) * W + min(max(xi + dx, 0), W - 1), c]
        for c in range(3):
            expect = (1 - fy) * ((1 - fx) * P(0, 0, c) + fx * P(1, 0, c)) + fy * ((1 - fx) * P(0, 1, c) + fx * P(1, 1, c))
            plain, budget = orc.decrypt(sk, out[o * 3 + c])
            assert budget > 0 and abs(orc.decode(plain) - expect) < 1e-6, (o, c)


# ---------------------------------------------------------------------------------------------
# the same streaming loop from a C++ host (seal/server_jpeg_hip.cpp): no Python, no torch
# ---------------------------------------------------------------------------------------------
def _server_jpeg_hip(*argv, env=None):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fully-homomorphic-image-processing_amd", "seal", "server_jpeg_hip")
    if not os.path.exists(exe):
        pytest.skip("seal/server_jpeg_hip not built")
    r = subprocess.run([exe] + [str(a) for a in argv], capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    return r.returncode, (json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]) if r.returncode == 0 else r.stderr)


@pytest.mark.parametrize("quant", [0, 1])
def test_cpp_streaming_server_writes_the_python_servers_bytes(fhe, oracle_mod, tmp_path, quant):
    """five colour blocks at the BASELINE parameter set in waves of two (ragged last wave), two passes over mapped files:
    the C++ host's output stream == server.server_jpeg's, byte for byte, and its first block equals the oracle's op-by-op result"""
    ctx = fhe.SEALContext.preset("P4096")
    orc = oracle_mod.Oracle.preset("P4096")
    n_blocks = 5
    fin, mine, cpp = tmp_path / "in.ct", tmp_path / "py.ct", tmp_path / "cpp.ct"
    cts = orc.random_ct(n_blocks * 192, seed=77)
    with open(fin, "wb") as f:
        for c in cts:
            fhe.server.write_ciphertext(f, c)
    assert fhe.server.server_jpeg(ctx, str(fin), str(mine), n_blocks, wave_blocks=2, quant=list(fhe.YQT) if quant else None) == n_blocks
    rc, res = _server_jpeg_hip(fin, cpp, n_blocks, 2, 4, 2, quant, 1 << 14, 4096)
    assert rc == 0, res
    assert res["blocks"] == n_blocks and len(res["seconds_per_pass"]) == 2
    got = open(cpp, "rb").read()
    assert got == open(mine, "rb").read()
    blk = cts[:192].reshape(3, 64, 2, orc.k, orc.n)
    ycc = np.zeros_like(blk)
    for i in range(64):
        ycc[0, i], ycc[1, i], ycc[2, i] = orc.rgb_to_ycc(blk[0, i], blk[1, i], blk[2, i])
    rec = 24 + 2 * orc.k * orc.n * 8
    first = np.frombuffer(got[24:rec], dtype=np.uint64).reshape(2, orc.k, orc.n)
    want = orc.dct_quant(ycc[0], oracle_mod.YQT) if quant else orc.encrypted_dct(ycc[0])
    assert np.array_equal(first, want[0])


def test_cpp_streaming_server_rejects_a_short_or_foreign_stream(fhe, tmp_path):
    ctx = fhe.SEALContext.preset("P4096")
    fin = tmp_path / "in.ct"
    with open(fin, "wb") as f:
        for c in fhe.to_host(ctx.random_ct(192, seed=3)):
            fhe.server.write_ciphertext(f, c)
    rc, err = _server_jpeg_hip(fin, tmp_path / "o.ct", 2)                    # two blocks asked for, one present
    assert rc == 1 and "stream ended" in err
    raw = bytearray(open(fin, "rb").read())
    raw[5 * (24 + 2 * ctx.k * ctx.n * 8)] ^= 0xFF                            # break the magic of the sixth record
    open(fin, "wb").write(bytes(raw))
    rc, err = _server_jpeg_hip(fin, tmp_path / "o.ct", 1)
    assert rc == 1 and "not a ciphertext record" in err


def test_cpp_streaming_server_feeds_the_reference_client(fhe, tmp_path):
    """the reference's client --send, the C++ streaming server, the reference's client --recieve: the published RMSError"""
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
    rc, res = _server_jpeg_hip(tmp_path / "image" / "ct_in.txt", tmp_path / "image" / "ct_out.txt", 36, 12, 8, 1, 0, 3001, 4096)
    assert rc == 0, res
    out = run([client, "--recieve", "-f", "image/in.jpg", "-i", "image/ct_out.txt", "-o", "image/out.jpg"] + par)
    line = [ln for ln in out.splitlines() if ln.startswith("RMSError,")]
    assert line and line[0].split(",")[1] == "1.71767", out[-500:]


def test_streaming_servers_reject_unreduced_residues(fhe, tmp_path):
    """record headers are checked by the I/O layer, the payload by fhe_count_unreduced on every uploaded wave: one residue equal to
    its modulus (what seal::Ciphertext::load would reject, homo/server_jpeg.cpp:117-123) fails the job in the Python server and in
    the C++ one; the same stream with the word repaired passes"""
    ctx = fhe.SEALContext.preset("P4096")
    cts = fhe.to_host(ctx.random_ct(192, seed=3)).copy()
    fin = tmp_path / "in.ct"

    def write():
        with open(fin, "wb") as f:
            for c in cts:
                fhe.server.write_ciphertext(f, c)
    good = cts[100, 1, 2, 77]
    cts[100, 1, 2, 77] = ctx.q[2]                                             # q_2 itself: not below the modulus
    write()
    with pytest.raises(ValueError, match="not reduced"):
        fhe.server.server_jpeg(ctx, str(fin), str(tmp_path / "o.ct"), 1)
    assert os.path.getsize(tmp_path / "o.ct") == 0                            # no complete-looking output stream is left behind
    rc, err = _server_jpeg_hip(fin, tmp_path / "o2.ct", 1)
    assert rc == 1 and "not reduced" in err and os.path.getsize(tmp_path / "o2.ct") == 0
    assert fhe.server.server_jpeg(ctx, str(fin), str(tmp_path / "o3.ct"), 1, validate=False) == 1      # the check can be waived for streams the server wrote itself
    cts[100, 1, 2, 77] = good
    write()
    assert fhe.server.server_jpeg(ctx, str(fin), str(tmp_path / "o.ct"), 1) == 1
    rc, res = _server_jpeg_hip(fin, tmp_path / "o2.ct", 1)
    assert rc == 0 and open(tmp_path / "o.ct", "rb").read() == open(tmp_path / "o2.ct", "rb").read()


def test_cpp_server_decode_writes_the_python_servers_bytes(fhe, tmp_path):
    """seal/server_decode_hip.cpp (C++ host over seal/hip_circuits.h: per channel one fhe_encrypt_batch, one load, one fhe_decode_channel)
    against server.server_decode with the same sampler key: the same output stream byte for byte (a channel without runs included), and
    the image it decrypts to; a stream with a residue that is not reduced is refused and the output left empty"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fully-homomorphic-image-processing_amd", "seal", "server_decode_hip")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    ctx = fhe.SEALContext.preset("P4096")
    kg = fhe.KeyGenerator(ctx, seed=8)
    enc = fhe.FractionalEncoder(ctx)
    rgb = np.zeros((2, 2, 3), dtype=np.uint8)
    rgb[:, :, 0] = 40
    rgb[1, :, 0] = 200                       # two runs
    rgb[:, :, 1] = 0                         # one run of zeros
    rgb[0, 0, 2], rgb[0, 1, 2], rgb[1, :, 2] = 7, 99, 180     # three runs
    fin, f_py, f_cpp, f_pk = (str(tmp_path / x) for x in ("runs.ct", "py.ct", "cpp.ct", "pubkey.txt"))
    w, h, pairs = fhe.client.send_decode(ctx, fhe.DeviceEncryptor(ctx, kg.public_key()), enc, rgb, fin)
    with open(f_pk, "wb") as f:
        fhe.server.write_ciphertext(f, fhe.to_host(kg.public_key()))
    degree = 2
    zeros = fhe.server.make_zero_encryptor(ctx, kg.public_key(), seed=77, device=True)
    fhe.server.server_decode(ctx, fin, f_py, w, h, pairs, zeros, order=64, degree=degree)
    key = fhe.server._sampler_key(77).hex()
    argv = [exe, fin, f_cpp, f_pk, str(w), str(h)] + [str(p) for p in pairs] + ["64", str(degree), "0.5", str(ctx.n), str(ctx.t), key]
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert open(f_cpp, "rb").read() == open(f_py, "rb").read()
    # the relinearised mode from an evaluation-key file: records of two polynomials, the Python server's bytes again
    f_evk, f_py2, f_cpp2 = (str(tmp_path / x) for x in ("evk.txt", "py_relin.ct", "cpp_relin.ct"))
    keys = kg.generate_evaluation_keys(16).contiguous()
    with open(f_evk, "wb") as f:
        fhe.server.write_evaluation_keys(f, keys, 16)
    fhe.server.server_decode(ctx, fin, f_py2, w, h, pairs, fhe.server.make_zero_encryptor(ctx, kg.public_key(), seed=77, device=True), order=64, degree=degree, relin=(keys, 16))
    r = subprocess.run([exe, fin, f_cpp2] + argv[3:] + [f_evk], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert open(f_cpp2, "rb").read() == open(f_py2, "rb").read() and os.path.getsize(f_cpp2) < os.path.getsize(f_cpp)
    # without a key argument: fresh randomness (at this small n the circuit's noise budget is gone, so the decrypted values are not compared)
    r = subprocess.run(argv[:-1], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and open(f_cpp, "rb").read() != open(f_py, "rb").read() and os.path.getsize(f_cpp) == os.path.getsize(f_py)
    # a residue that is not reduced: refused, nothing complete-looking left behind
    raw = bytearray(open(fin, "rb").read())
    raw[fhe.server.RECORD_HEADER:fhe.server.RECORD_HEADER + 8] = b"\xff" * 8
    bad = str(tmp_path / "bad.ct")
    open(bad, "wb").write(bytes(raw))
    f_bad_out = str(tmp_path / "bad_out.ct")
    r = subprocess.run([exe, bad, f_bad_out] + argv[3:], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "not reduced" in r.stderr and (not os.path.exists(f_bad_out) or os.path.getsize(f_bad_out) == 0)


@pytest.mark.parametrize("bicubic,W,H,w,h,rows,shared", [(True, 7, 9, 5, 6, 2, False), (False, 6, 5, 4, 3, 4, False), (True, 12, 16, 3, 3, 1, False), (True, 7, 9, 5, 6, 2, True),
                                                         (True, 12, 16, 3, 3, 4, True)])
def test_cpp_server_resize_writes_the_python_servers_bytes(fhe, tmp_path, bicubic, W, H, w, h, rows, shared):
    """seal/server_resize_hip.cpp (C++ host: reader / upload / circuits / download / writer pipeline over seal/hip_circuits.h + fhe_stream.h,
    one fhe_encrypt_batch per step) against server.server_resize with the same sampler key: the same output stream byte for byte -- the
    reference's sliding row window, float index arithmetic, tap order and encryption order restated twice and compared; incl. strong
    down-scaling (the window jumps) and one-row steps"""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fully-homomorphic-image-processing_amd", "seal", "server_resize_hip")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    ctx = fhe.SEALContext.preset("P4096")
    kg = fhe.KeyGenerator(ctx, seed=12)
    enc = fhe.FractionalEncoder(ctx)
    rgb = np.random.default_rng(W * H).integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    fin, f_py, f_cpp, f_pk = (str(tmp_path / x) for x in ("in.ct", "py.ct", "cpp.ct", "pubkey.txt"))
    assert fhe.client.send_resize(ctx, fhe.DeviceEncryptor(ctx, kg.public_key()), enc, rgb, fin) == (W, H)
    with open(f_pk, "wb") as f:
        fhe.server.write_ciphertext(f, fhe.to_host(kg.public_key()))
    fractions = fhe.server.make_fraction_encryptor(ctx, kg.public_key(), enc, seed=5, device=True)
    assert fhe.server.server_resize(ctx, fin, f_py, W, H, w, h, bicubic, fractions, rows_per_step=rows, shared_offsets=shared) == w * h
    argv = [exe, fin, f_cpp, f_pk, str(W), str(H), str(w), str(h), "1" if bicubic else "0", str(rows), "4", str(ctx.n), str(ctx.t), fhe.server._sampler_key(5).hex()]
    tail = ["1", "1"] if shared else []                            # passes, shared offsets (one ciphertext per output column / row)
    r = subprocess.run(argv + tail, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.getsize(f_cpp) == os.path.getsize(f_py)
    assert open(f_cpp, "rb").read() == open(f_py, "rb").read()
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert '"server_side_encryptions": %d' % ((w + h) if shared else 2 * w * h) in line
    # and the image it decrypts to equals the Python server's (fresh randomness without the key argument)
    r = subprocess.run(argv[:-1] + (["-"] + tail if shared else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and open(f_cpp, "rb").read() != open(f_py, "rb").read()
    if not bicubic:                              # the bicubic circuit leaves no noise budget to speak of at n = 4096: fresh randomness may flip a high coefficient
        dec = fhe.Decryptor(ctx, kg.secret_key())
        a, b = [], []
        fhe.client.receive_resize(ctx, dec, enc, f_py, w, h, decoded=a)
        fhe.client.receive_resize(ctx, dec, enc, f_cpp, w, h, decoded=b)
        assert a == b


@pytest.mark.parametrize("placement,shared,bicubic", [("product", False, True), ("cubic", False, True), ("sample", False, True), ("cubic", True, True), ("sample", False, False)])
def test_cpp_server_resize_in_the_relinearised_modes_writes_the_python_servers_bytes(fhe, tmp_path, placement, shared, bicubic):
    """The C++ host with an evaluation-key FILE (seal::EvaluationKeys::load; written here by server.write_evaluation_keys -- what a client that
    honoured the reference's parsed-and-unused --dbc would send, homo/client_resize.cpp:26,47,72) and a relinearisation placement (after every
    product / once per Cubic / once per output pixel) against server.server_resize(relin=(keys, dbc[, placement])) with the same sampler key:
    records of two polynomials, the same stream byte for byte."""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fully-homomorphic-image-processing_amd", "seal", "server_resize_hip")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    ctx = fhe.SEALContext.preset("SEAL23_4096")
    kg = fhe.KeyGenerator(ctx, seed=12)
    enc = fhe.FractionalEncoder(ctx)
    W, H, w, h, dbc = 7, 9, 5, 6, 30
    rgb = np.random.default_rng(7).integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    fin, f_py, f_cpp, f_pk, f_evk = (str(tmp_path / x) for x in ("in.ct", "py.ct", "cpp.ct", "pubkey.txt", "evk.txt"))
    assert fhe.client.send_resize(ctx, fhe.DeviceEncryptor(ctx, kg.public_key()), enc, rgb, fin) == (W, H)
    with open(f_pk, "wb") as f:
        fhe.server.write_ciphertext(f, fhe.to_host(kg.public_key()))
    count = {"product": 1, "cubic": 2, "sample": 4}[placement]
    keys = kg.generate_evaluation_keys(dbc, count).contiguous()
    with open(f_evk, "wb") as f:
        fhe.server.write_evaluation_keys(f, keys, dbc)
    relin = (keys, dbc) if placement == "product" else (keys, dbc, placement)
    fractions = fhe.server.make_fraction_encryptor(ctx, kg.public_key(), enc, seed=5, device=True)
    assert fhe.server.server_resize(ctx, fin, f_py, W, H, w, h, bicubic, fractions, rows_per_step=2, shared_offsets=shared, relin=relin) == w * h
    rec = fhe.server.RECORD_HEADER + 2 * ctx.k * ctx.n * 8
    assert os.path.getsize(f_py) == w * h * 3 * rec                    # two polynomials per record
    argv = [exe, fin, f_cpp, f_pk, str(W), str(H), str(w), str(h), "1" if bicubic else "0", "2", "4", str(ctx.n), str(ctx.t), fhe.server._sampler_key(5).hex(), "1",
            "1" if shared else "0", f_evk, str({"product": 0, "cubic": 1, "sample": 2}[placement])]
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=dict(os.environ, FHE_SEAL23_MODULI="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert open(f_cpp, "rb").read() == open(f_py, "rb").read()
    # a truncated key file is refused before anything is computed
    open(f_evk, "r+b").truncate(os.path.getsize(f_evk) - 8)
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=dict(os.environ, FHE_SEAL23_MODULI="1"))
    assert r.returncode != 0 and "truncated evaluation key stream" in r.stderr
    # a self-consistent key file whose digit count cannot belong to its decomposition bit count on this context (ONE digit of 30 bits for primes
    # of 54 / 55 bits; the library would read two digits' worth behind the pointer): refused by the consumer, EvaluationKeys::require_for
    import struct
    one_digit = count * ctx.k * 1 * 2 * ctx.k * ctx.n
    with open(f_evk, "wb") as f:
        f.write(struct.pack("<8sIIIIII", b"FHEHIPK\0", dbc, 1, count, ctx.k, ctx.n, 0) + bytes(one_digit * 8))
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=dict(os.environ, FHE_SEAL23_MODULI="1"))
    assert r.returncode != 0 and "digit count does not fit" in r.stderr, r.stderr[-1000:]


def test_server_resize_with_shared_offsets(fhe, tmp_path):
    """server_resize(shared_offsets=True): one offset ciphertext per output column and row instead of two per output pixel.  Not the
    reference's ciphertexts -- but the bytes equal the shared-offset circuit called on the whole image with the same encryptions, two row
    shards write the whole run's file, and the decrypted image equals the per-pixel mode's, value for value"""
    ctx = fhe.SEALContext.preset("P8192")
    kg = fhe.KeyGenerator(ctx, seed=14)
    enc, ev = fhe.FractionalEncoder(ctx), fhe.Evaluator(ctx)
    W, H, w, h = 6, 7, 4, 4
    rgb = np.random.default_rng(2).integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    fin, f_sh, f_parts, f_px = (str(tmp_path / x) for x in ("in.ct", "shared.ct", "parts.ct", "pixel.ct"))
    client = fhe.DeviceEncryptor(ctx, kg.public_key(), key=bytes(32))
    pix = client.encrypt_values(rgb.reshape(-1).astype(np.float64))                 # [H * W * 3, 2, k, n], the stream's order
    with open(fin, "wb") as f:
        for c in fhe.to_host(pix):
            fhe.server.write_ciphertext(f, c)
    mk = lambda: fhe.server.make_fraction_encryptor(ctx, kg.public_key(), enc, seed=3, device=True)
    assert fhe.server.server_resize(ctx, fin, f_sh, W, H, w, h, True, mk(), rows_per_step=2, shared_offsets=True) == w * h
    whole = open(f_sh, "rb").read()
    for rows in ((0, 1), (1, 4)):
        fhe.server.server_resize(ctx, fin, f_parts, W, H, w, h, True, mk(), rows_per_step=2, rows=rows, shared_offsets=True)
    assert open(f_parts, "rb").read() == whole
    # the circuit called directly on the whole image with the same encryptions (columns first, then rows)
    _, xs, ys = fhe.circuits.resize_sample_plan(W, H, w, h, bicubic=True)
    direct = mk()
    fr = direct([xs[x] for x in range(w)] + [ys[y * w] for y in range(h)])
    pc = fhe.circuits.PlainCache(ctx)
    rec = fhe.server.RECORD_HEADER + 6 * ctx.k * ctx.n * 8
    pix3 = pix.view(H * W, 3, 2, ctx.k, ctx.n)
    for ch in range(3):
        out = fhe.to_host(fhe.circuits.resize_bicubic_shared(ev, pc, pix3[:, ch].contiguous(), W, H, w, h, fr[:w].contiguous(), fr[w:].contiguous()))
        for p in range(w * h):
            off = (p * 3 + ch) * rec + fhe.server.RECORD_HEADER
            assert np.array_equal(np.frombuffer(whole[off:off + rec - fhe.server.RECORD_HEADER], dtype=np.uint64).reshape(6, ctx.k, ctx.n), out[p]), (ch, p)
    # the per-pixel mode (the reference's) decrypts to the same values
    assert fhe.server.server_resize(ctx, fin, f_px, W, H, w, h, True, mk(), rows_per_step=2) == w * h
    dec = fhe.Decryptor(ctx, kg.secret_key())
    a, b = [], []
    fhe.client.receive_resize(ctx, dec, enc, f_sh, w, h, decoded=a)
    fhe.client.receive_resize(ctx, dec, enc, f_px, w, h, decoded=b)
    assert a == b and len(a) == w * h * 3
    with pytest.raises(ValueError):
        fhe.server.server_resize(ctx, fin, f_px, W, H, w, h, False, mk(), shared_offsets=True)
