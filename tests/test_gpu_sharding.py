"""GPU: the multi-GPU partitions of the three circuits, with the HIP library as the per-shard compute.

One device is enough for everything except the RCCL runs: a shard is a function of (rank, world) only, so the shards of a
world of R are evaluated one after the other on the same GPU and must reproduce the single-process result bit for bit.

configs[4]  BASELINE.json "Batch 64 images x 1024 blocks DCT+quant, n=4096, block-sharded across 8x MI355X": one GPU's
            share -- 8,192 blocks -- through fhe_dct8x8_quant at its stated size; its position-dependent digest equals
            the sum of the eight 1,024-block shard digests whose inputs are generated with parallel.shard_first_index
            (any GPU count produces the same bytes), and sampled blocks equal the oracle (homo/server_jpeg.cpp:113-138).
resize      destination rows [y0, y1) + source rows +- halo (homo/fhe_resize.h:350-388): library entry point and the
            streaming server, bands of worlds 2 and 3 == the whole image, byte for byte in one output file.
decode      (channel, position) units (homo/server_decode.cpp:120-137): library entry point and the streaming server.
RCCL        two-device variants of the resize / decode benches (skipped on a one-GPU box)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1


# ------------------------------------------------------------------------------------------------
# configs[4]: one GPU's share at its stated size
# ------------------------------------------------------------------------------------------------
def test_config4_one_gpu_share_8192_blocks(fhe, oracle_mod):
    import torch
    ctx = fhe.SEALContext.preset("P4096")
    orc = oracle_mod.Oracle.preset("P4096")
    ev, plan = fhe.Evaluator(ctx), fhe.DctPlan(ctx, fhe.YQT)
    share, ranks = 8192, 8                                  # 64 images x 1024 blocks over 8 GPUs; this GPU plays global rank 3
    n_global, rank = share * ranks, 3
    wpb = fhe.parallel.words_per_block(ctx.k, ctx.n)
    start, end = fhe.parallel.block_range(rank, ranks, n_global)
    assert (start, end) == (rank * share, (rank + 1) * share)
    first = fhe.parallel.shard_first_index(rank, ranks, n_global, ctx.k, ctx.n)
    assert first == start * wpb
    # (a) eight 1,024-block pieces of the share, each generated and digested on its own with global indices
    pieces = 0
    for r in range(8):
        f = fhe.parallel.shard_first_index(rank * 8 + r, ranks * 8, n_global, ctx.k, ctx.n)
        assert f == first + r * 1024 * wpb
        blocks = ctx.random_ct(1024, 64, seed=fhe.SEED, first_index=f)
        out = ev.dct8x8_quant(plan, blocks)
        pieces = (pieces + ctx.digest(out.view(-1), index0=f)) & M64
        if r == 5:
            keep_in, keep_out = fhe.to_host(blocks[17]), fhe.to_host(out[17])
        del blocks, out
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert np.array_equal(keep_out, orc.dct_quant(keep_in, oracle_mod.YQT))                  # block 5 * 1024 + 17 of the share
    # (b) the share as ONE call: 8,192 blocks resident (96 GiB in + 96 GiB out), or two halves when the device is short of memory
    free, _ = torch.cuda.mem_get_info()
    halves = 1 if free > 205 * (1 << 30) else 2
    whole, per = 0, share // halves
    for hh in range(halves):
        f = first + hh * per * wpb
        blocks = ctx.random_ct(per, 64, seed=fhe.SEED, first_index=f)
        out = ev.dct8x8_quant(plan, blocks)
        whole = (whole + ctx.digest(out.view(-1), index0=f)) & M64
        if hh == 0:
            assert np.array_equal(fhe.to_host(out[5 * 1024 + 17]), keep_out)
            b0 = orc.dct_quant(fhe.to_host(blocks[0]), oracle_mod.YQT)
            assert np.array_equal(fhe.to_host(out[0]), b0)
        if hh == halves - 1:
            bl = orc.dct_quant(fhe.to_host(blocks[per - 1]), oracle_mod.YQT)
            assert np.array_equal(fhe.to_host(out[per - 1]), bl)                              # the last block of the share
        del blocks, out
        torch.cuda.empty_cache()
    assert whole == pieces


# ------------------------------------------------------------------------------------------------
# resize: destination rows
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("preset,W,H,w,h,band,batch", [("SMALL", 16, 12, 8, 7, 3, 16), ("SMALL", 9, 9, 17, 17, 4, 64), ("P8192", 24, 24, 12, 12, 4, 48)])
def test_resize_row_shards_equal_the_whole_image(fhe, preset, W, H, w, h, band, batch):
    import torch
    ctx = fhe.SEALContext(1024, [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], 1 << 14, 0) if preset == "SMALL" else fhe.SEALContext.preset(preset)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    pixels = ctx.random_ct(W * H, size=2, seed=5)
    xf, yf = ctx.random_ct(w, size=2, seed=11), ctx.random_ct(h, size=2, seed=12)
    whole = fhe.circuits.resize_bicubic_shared(ev, pc, pixels, W, H, w, h, xf, yf, batch=batch, band_rows=band)
    words = 6 * ctx.k * ctx.n
    d_whole = ctx.digest(whole, index0=0)
    for world in (2, 3, 8):
        total, covered = 0, 0
        for rank in range(world):
            y0, y1 = fhe.parallel.row_range(rank, world, h)
            if y0 == y1:
                continue
            first, count = fhe.circuits.resize_source_rows(H, h, y0, y1)
            assert (first, count) == fhe.parallel.source_rows(H, h, y0, y1)
            mine = pixels[first * W:(first + count) * W].clone()                             # this rank's rows +- halo and nothing else
            got = fhe.circuits.resize_bicubic_shared(ev, pc, mine, W, H, w, h, xf, yf[y0:y1].contiguous(), batch=batch, band_rows=band,
                                                     rows=(y0, y1), src_rows=(first, count))
            assert got.shape == ((y1 - y0) * w, 6, ctx.k, ctx.n)
            assert torch.equal(got, whole[y0 * w:y1 * w]), (world, rank)
            total = (total + ctx.digest(got, index0=y0 * w * words)) & M64
            covered += y1 - y0
            # streamed form of a shard: global first_pixel indices
            seen = []
            assert fhe.circuits.resize_bicubic_shared(ev, pc, mine, W, H, w, h, xf, yf[y0:y1].contiguous(), batch=batch, band_rows=band, rows=(y0, y1),
                                                      src_rows=(first, count), consume=lambda f, t: seen.append((f, int(t.shape[0])))) is None
            assert seen[0][0] == y0 * w and sum(c for _, c in seen) == (y1 - y0) * w
        assert covered == h and total == d_whole
    # a shard whose resident rows miss the halo is refused before anything is enqueued
    y0, y1 = fhe.parallel.row_range(1, 2, h)
    first, count = fhe.circuits.resize_source_rows(H, h, y0, y1)
    if count > 1:
        with pytest.raises(fhe._lib.FheError, match="outside the resident rows"):
            fhe.circuits.resize_bicubic_shared(ev, pc, pixels[(first + 1) * W:(first + count) * W].clone(), W, H, w, h, xf, yf[y0:y1].contiguous(),
                                               rows=(y0, y1), src_rows=(first + 1, count - 1))


@pytest.mark.parametrize("bicubic", [True, False])
def test_server_resize_row_shards_write_the_single_process_file(fhe, oracle_mod, tmp_path, bicubic):
    """three processes' worth of server_resize(rows=...) into ONE output file == one whole-image run, byte for byte, with
    the server-side encryptions indexed by their position in the reference's sequence (make_fraction_encryptor(indexed))"""
    from refrun import write_record
    p = oracle_mod.PRESETS["P4096"]
    ctx, orc = fhe.SEALContext(p["n"], p["q"], p["t"]), oracle_mod.Oracle(p["n"], p["q"], p["t"])
    _, pk = orc.keygen(5)
    W, H, w, h = 7, 10, 5, 8
    pix = orc.random_ct(W * H * 3, seed=31)
    fin = tmp_path / "in.ct"
    with open(fin, "wb") as f:
        for i in range(W * H * 3):
            write_record(f, pix[i])
    one = tmp_path / "one.ct"
    enc = fhe.server.make_fraction_encryptor(ctx, fhe.to_device(pk), seed=9, indexed=True)
    assert fhe.server.server_resize(ctx, str(fin), str(one), W, H, w, h, bicubic, enc, rows_per_step=2) == w * h
    for world in (2, 3):
        out = tmp_path / ("sharded%d.ct" % world)
        done = 0
        for rank in reversed(range(world)):                                                   # any order: bands land at their own offsets
            enc_r = fhe.server.make_fraction_encryptor(ctx, fhe.to_device(pk), seed=9, indexed=True)
            stats = {}
            done += fhe.server.server_resize(ctx, str(fin), str(out), W, H, w, h, bicubic, enc_r, rows_per_step=2,
                                             rows=fhe.parallel.row_range(rank, world, h), stats=stats)
            rec_in = 24 + 2 * ctx.k * ctx.n * 8
            assert stats["bytes_in"] < W * H * 3 * rec_in                                     # a shard reads its rows +- halo, not the image
        assert done == w * h
        assert open(out, "rb").read() == open(one, "rb").read()


# ------------------------------------------------------------------------------------------------
# decode: (channel, position) units
# ------------------------------------------------------------------------------------------------
def test_decode_position_shards_equal_the_whole_run(fhe):
    import torch
    ctx = fhe.SEALContext(1024, [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], 1 << 14, 0)
    ev, pc = fhe.Evaluator(ctx), fhe.circuits.PlainCache(ctx)
    npos, degree, pairs = 5, 2, 2
    amp, idx, cnt = (ctx.random_ct(1, size=2, seed=900 + i) for i in range(3))
    zeros = ctx.random_ct(npos * degree * 2, size=2, seed=77).reshape(npos, degree, 2, 2, ctx.k, ctx.n)
    whole = torch.cat(fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, 64, degree, 0.5, npos, 1, zeros))
    for (p0, p1) in ((0, 2), (2, 5), (4, 5), (0, 5)):
        got = torch.cat(fhe.circuits.approximated_step(ev, pc, amp, idx, cnt, 64, degree, 0.5, npos, 1, zeros[p0:p1].contiguous(), positions=(p0, p1)))
        assert torch.equal(got, whole[p0:p1]), (p0, p1)
    # the channel loop: every shard advances its own copy of `index` to the same end value
    runs = ctx.random_ct(pairs * 2, size=2, seed=41).reshape(pairs, 2, 2, ctx.k, ctx.n)
    acc0 = ctx.random_ct(npos, size=2, seed=42)
    zr = ctx.random_ct(pairs * npos * degree * 2, size=2, seed=43).reshape(pairs, npos, degree, 2, 2, ctx.k, ctx.n)
    index = idx.clone()
    whole = fhe.circuits.decode_channel(ev, pc, runs, index, acc0, zr, 64, degree, 0.5, npos, 1)
    for (p0, p1) in ((0, 3), (3, 5)):
        mine = idx.clone()
        got = fhe.circuits.decode_channel(ev, pc, runs, mine, acc0[p0:p1].contiguous(), zr[:, p0:p1].contiguous(), 64, degree, 0.5, npos, 1, positions=(p0, p1))
        assert torch.equal(got, whole[p0:p1]) and torch.equal(mine, index)
    with pytest.raises(fhe._lib.FheError, match="not a range"):
        fhe.circuits.decode_channel(ev, pc, runs, idx.clone(), acc0[:1].contiguous(), zr[:, :1].contiguous(), 64, degree, 0.5, npos, 1, positions=(5, 6))


def test_server_decode_unit_shards_write_the_single_process_file(fhe, oracle_mod, tmp_path):
    p = oracle_mod.PRESETS["P4096"]
    ctx, orc = fhe.SEALContext(p["n"], p["q"], p["t"]), oracle_mod.Oracle(p["n"], p["q"], p["t"])
    _, pk = orc.keygen(5)
    pairs, w, h, degree = (2, 0, 1), 2, 2, 2
    runs = orc.random_ct(2 * sum(pairs), seed=77)
    fin = tmp_path / "in.ct"
    with open(fin, "wb") as f:
        for r in range(runs.shape[0]):
            fhe.server.write_ciphertext(f, runs[r])
    one = tmp_path / "one.ct"
    enc = fhe.server.make_zero_encryptor(ctx, fhe.to_device(pk), seed=3, indexed=True)
    assert fhe.server.server_decode(ctx, str(fin), str(one), w, h, pairs, enc, degree=degree) == w * h
    for world in (2, 5):
        out = tmp_path / ("sharded%d.ct" % world)
        done = 0
        for rank in reversed(range(world)):
            enc_r = fhe.server.make_zero_encryptor(ctx, fhe.to_device(pk), seed=3, indexed=True)
            done += fhe.server.server_decode(ctx, str(fin), str(out), w, h, pairs, enc_r, degree=degree, shard=(rank, world))
        assert done == 3 * w * h
        assert open(out, "rb").read() == open(one, "rb").read()


# ------------------------------------------------------------------------------------------------
# RCCL: two devices
# ------------------------------------------------------------------------------------------------
def _bench_circuits(args, nproc, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    script = os.path.join(ROOT, "bench_circuits.py")
    if nproc == 1:
        cmd = [sys.executable, script] + args
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), script] + args + ["--gpus", str(nproc)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_circuits_sharded_modes_on_one_gpu():
    """bench_circuits.py --shared resize and decode carry the BENCH-schema objects and a digest of everything produced"""
    a = _bench_circuits(["resize", "--preset", "P4096", "--src", "24", "--dst", "12", "--pixels", "48", "--shared", "--cpu-pixels", "1"], 1)
    assert a["n_gpus"] == 1 and a["roofline"]["bound"] == "hbm" and a["cpu_baseline"]["kind"] == "port" and a["output_digest"]
    # both relinearised placements carry the CPU baseline too (the oracle's op-by-op composition of the mode), and say which mode they are
    r1 = _bench_circuits(["resize", "--preset", "SEAL23_4096", "--src", "24", "--dst", "12", "--pixels", "48", "--relin", "30", "--cpu-pixels", "1"], 1)
    r2 = _bench_circuits(["resize", "--preset", "SEAL23_4096", "--src", "24", "--dst", "12", "--pixels", "48", "--relin", "30", "--relin-placement", "cubic", "--cpu-pixels", "1"], 1)
    for r in (r1, r2):
        assert r["out_size"] == 2 and r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["value"] > 0
    assert "every multiply" in r1["config"]["mode"] and "ONE evaluator.relinearize" in r2["config"]["mode"] and r1["output_digest"] != r2["output_digest"]
    b = _bench_circuits(["decode", "--preset", "P4096", "--positions", "4", "--degree", "2"], 1)
    assert b["n_gpus"] == 1 and b["output_digest"] and b["roofline"]["frac"] > 0


def test_world2_paths_of_bench_circuits_on_one_gpu_over_gloo():
    """the row-sharded resize (per-pixel and shared offsets) and the position-sharded decode as TWO processes on device 0, gloo for
    the collectives (FHE_BENCH_BACKEND=gloo): the launch the driver would make on two GPUs, minus RCCL.  Same digests as one rank."""
    for base in (["resize", "--preset", "P4096", "--src", "24", "--dst", "12", "--pixels", "48", "--shared"],
                 ["resize", "--preset", "P4096", "--src", "24", "--dst", "12", "--pixels", "48"],
                 ["decode", "--preset", "P4096", "--positions", "4", "--degree", "2"]):
        one, two = _bench_circuits(base, 1), _bench_circuits(base, 2, FHE_BENCH_BACKEND="gloo")
        assert two["n_gpus"] == 2 and one["output_digest"] == two["output_digest"], base
        # the fixed-size job's in-run N = 1 leg (rank 0 alone over the WHOLE job): speed-up and efficiency in the same line
        ss = two["strong_scaling"]
        assert ss["n1_seconds"] > 0 and abs(ss["speedup"] - ss["n1_seconds"] / two["seconds"]) < 1e-9 and abs(ss["efficiency"] - ss["speedup"] / 2) < 1e-12
        assert "strong_scaling" not in one


def test_two_gpus_resize_rows_and_decode_units_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (RCCL refuses two ranks on one device)")
    base = ["resize", "--preset", "P4096", "--src", "24", "--dst", "12", "--pixels", "48", "--shared"]
    assert _bench_circuits(base, 1)["output_digest"] == _bench_circuits(base, 2)["output_digest"]
    base = ["decode", "--preset", "P4096", "--positions", "4", "--degree", "2"]
    one, two = _bench_circuits(base, 1), _bench_circuits(base, 2)
    assert one["output_digest"] == two["output_digest"] and two["n_gpus"] == 2


# ------------------------------------------------------------------------------------------------
# the N > 1 path from a C++ host (seal/multi_gpu_dct.cpp): one fhe_ctx + one thread per rank, RCCL gather on >= 2 devices
# ------------------------------------------------------------------------------------------------
def _multi_gpu_dct(*argv):
    exe = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "seal", "multi_gpu_dct")
    if not os.path.exists(exe):
        pytest.skip("seal/multi_gpu_dct not built")
    r = subprocess.run([exe] + [str(a) for a in argv], capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_cpp_host_block_shards_equal_one_rank(fhe):
    """three ranks (ragged split 43 + 43 + 42, waves of 16) in one C++ process: rank digests add up to the single-rank
    digest, which is also what the Python path computes for the same 128 synthetic blocks"""
    res = _multi_gpu_dct(128, 3, 16, 1)
    assert res["ranks"] == 3 and res["digests_equal"] is True
    ctx = fhe.SEALContext.preset("P4096")
    ev, plan = fhe.Evaluator(ctx), fhe.DctPlan(ctx, fhe.YQT)
    out = ev.dct8x8_quant(plan, ctx.random_ct(128, 64, seed=fhe.SEED))
    assert "%016x" % ctx.digest(out.view(-1), index0=0) == res["output_digest"]


def test_cpp_host_resident_mode_times_what_bench_py_times(fhe):
    """`multi_gpu_dct ... resident`: inputs generated before the clock, outputs digested after it, 256-block waves, several passes --
    the timed region of bench.py from a C++ host.  One rank over 1024 blocks: the digest is the verifying loop's, and the rate is the
    kernels' (bench.py: ~80 k blocks/s; the verifying loop: ~39 k), asserted here at >= 70 k so that box-to-box noise does not flake.
    Ragged resident shards (3 ranks over 200 blocks in waves of 32) add up to the same digest as one rank."""
    res = _multi_gpu_dct(1024, 1, 256, 0, "resident", 10)
    assert res["mode"].startswith("resident") and res["reps"] == 10 and res["digests_equal"] is True
    assert res["blocks_per_s"] >= 70000, res
    assert abs(res["efficiency_vs_single_rank"] - 1.0) < 0.1
    ragged = _multi_gpu_dct(200, 3, 32, 1, "resident", 2)
    assert ragged["ranks"] == 3 and ragged["digests_equal"] is True


def test_cpp_host_two_devices_rccl_gather():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (RCCL refuses two ranks on one device)")
    res = _multi_gpu_dct(256, 2, 32, 1)
    assert res["digests_equal"] is True and res["gathered_digest_equals_senders"] is True
    res = _multi_gpu_dct(512, 2, 64, 1, "resident", 3)
    assert res["digests_equal"] is True and res["gathered_digest_equals_senders"] is True
