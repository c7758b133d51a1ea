"""GPU: the N > 1 path with the HIP evaluator as the per-shard compute.  Needs two devices for the RCCL run
(skipped on a one-GPU box -- no box this repository has run on had two, so NO RCCL transfer has executed yet); the
single-device checks of the gather mode and the world-2 code paths over gloo run everywhere."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, nproc=1, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_gather_wave_mode_on_one_gpu_gives_the_same_digest():
    """--gather wave with one rank: the same kernels wave by wave, drained by digest; the digest of the whole output
    must equal the ungathered run's (order-independent digest, additive over waves)."""
    base = ["--blocks", "128", "--steps", "2", "--warmup", "1", "--cpu-blocks", "0"]
    a = _bench(base)
    b = _bench(base + ["--gather", "wave", "--gather-wave-blocks", "32"])
    assert a["verified_bit_exact_vs_oracle"] and b["verified_bit_exact_vs_oracle"]
    assert a["output_digest"] == b["output_digest"]
    assert b["config"]["gather"] == "wave"


def test_two_gpus_sharded_run_and_rccl_wave_gather():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (RCCL refuses two ranks on one device)")
    base = ["--blocks", "128", "--steps", "2", "--warmup", "1", "--cpu-blocks", "0"]
    one = _bench(["--blocks", "256", "--steps", "1", "--warmup", "0", "--cpu-blocks", "0"])
    two = _bench(base, nproc=2)
    gathered = _bench(base + ["--gather", "wave", "--gather-wave-blocks", "32"], nproc=2)
    # 2 x 128 blocks generate the same bytes as 1 x 256 (global block index seeds the inputs)
    assert one["output_digest"] == two["output_digest"] == gathered["output_digest"]
    assert two["n_gpus"] == 2 and gathered["config"]["gather"] == "wave"


def test_world2_path_of_bench_on_one_gpu_over_gloo():
    """bench.py's world > 1 code path (global-index seeding per rank, barrier-bracketed timing, MAX / SUM all-reduces, all-gather of
    the per-rank times, digest all-reduce) with two ranks that share device 0 and gloo for the collectives (FHE_BENCH_BACKEND=gloo):
    what a one-GPU box can run of the launch the driver makes on 2/4/8 GPUs.  2 x 128 blocks == 1 x 256 blocks, by digest."""
    one = _bench(["--blocks", "256", "--steps", "1", "--warmup", "0", "--cpu-blocks", "0"])
    two = _bench(["--blocks", "128", "--steps", "2", "--warmup", "1", "--cpu-blocks", "0"], nproc=2, FHE_BENCH_BACKEND="gloo")
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and len(two["ms_per_step_per_rank"]) == 2
    assert two["collective_backend"].startswith("gloo") and two["verified_bit_exact_vs_oracle"]
    assert one["output_digest"] == two["output_digest"]
    assert abs(two["value"] - 2 * 128 / (two["ms_per_step"] * 1e-3)) < 1e-6 * two["value"]


def _direct(script, args, **extra_env):
    """the launch the driver makes: `python <script> --gpus N ...`, no launcher around it"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)


def test_bench_gpus_2_is_the_whole_launch():
    """`python bench.py --gpus 2` by itself starts two ranks (bench.ensure_world re-executes under torch.distributed.run); on a
    one-GPU box the ranks share the device over gloo.  n_gpus == rccl_ranks == 2 and 2 x 128 blocks == 1 x 256 blocks by digest."""
    one = _bench(["--blocks", "256", "--steps", "1", "--warmup", "0", "--cpu-blocks", "0"])
    r = _direct("bench.py", ["--gpus", "2", "--blocks", "128", "--steps", "2", "--warmup", "1", "--cpu-blocks", "0"], FHE_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    two = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and len(two["ms_per_step_per_rank"]) == 2
    assert two["output_digest"] == one["output_digest"] and two["verified_bit_exact_vs_oracle"]


def test_the_one_driver_command_prints_the_whole_multi_gpu_record():
    """`python bench.py --gpus 2 --steps 3` -- the command the driver runs at N > 1, nothing else on the line -- prints ONE JSON line that
    carries the compute-only `value`, the CPU baseline of the same run, an in-run N = 1 leg with the weak-scaling efficiency, BOTH
    gather legs (wave-to-root with its xGMI link ceiling, local PCIe drain) and the aggregate roofline.  On a one-GPU box the two ranks
    share the device over gloo (FHE_BENCH_BACKEND=gloo): the schema and every code path of the record, not the physics."""
    import torch
    shared = torch.cuda.device_count() < 2
    r = _direct("bench.py", ["--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "256", "--cpu-blocks", "2"], **({"FHE_BENCH_BACKEND": "gloo"} if shared else {}))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["config"]["gather"] == "none"
    assert d["verified_bit_exact_vs_oracle"] and len(d["output_digest"]) == 16
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "blocks/s"
    ws = d["weak_scaling"]
    assert ws["n1_blocks_per_s"] > 0 and abs(ws["per_gpu_blocks_per_s"] - d["value"] / 2) < 1e-6 * d["value"]
    assert abs(ws["efficiency"] - d["value"] / (2 * ws["n1_blocks_per_s"])) < 1e-9
    if shared:
        assert 0.3 < ws["efficiency"] < 0.75 and "TEST MODE" in ws["how"]       # two ranks on one device: about one half by construction
    g = d["gather"]
    wave, local = g["wave"], g["local"]
    assert "error" not in wave and "error" not in local, g
    assert wave["digest_matches_compute_only"] is True and wave["peers"] == 1
    assert abs(wave["link_ceiling_blocks_per_s_per_peer"] - 153e9 / g["output_bytes_per_block"]) < 1e-6 * wave["link_ceiling_blocks_per_s_per_peer"]
    assert wave["xgmi_GB_per_s_into_root"] > 0 and 0 < wave["blocks_per_s"] <= 1.05 * d["value"]
    assert local["drained_bytes_equal_in_hbm_result_on_every_rank"] is True
    assert local["pcie_GB_per_s_per_gpu"] > 0 and abs(local["pcie_GB_per_s_aggregate"] - 2 * local["pcie_GB_per_s_per_gpu"]) < 1e-9 * local["pcie_GB_per_s_aggregate"]
    agg = d["roofline"]["aggregate"]
    assert agg["peak"] == 2 * d["roofline"]["peak"] and len(agg["achieved_per_rank"]) == 2 and abs(agg["frac"] - agg["achieved"] / agg["peak"]) < 1e-12


def test_a_gather_leg_that_hangs_is_abandoned_and_the_line_still_appears():
    """the RCCL wave gather has never run between two devices: a leg that does not complete must not take the line with it.  With
    FHE_BENCH_GATHER_TIMEOUT=0.001 the watchdog of the first leg fires at once on every rank: rank 0 prints the line measured so far
    (value, cpu_baseline, weak_scaling) with the leg marked, and every rank leaves with exit code 0."""
    r = _direct("bench.py", ["--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "128", "--cpu-blocks", "0"], FHE_BENCH_BACKEND="gloo", FHE_BENCH_GATHER_TIMEOUT="0.001")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["weak_scaling"]["efficiency"] > 0
    assert "watchdog" in d["gather"]["local"]["error"]


def test_bench_gpus_more_than_devices_fails_loudly():
    """--gpus 8 on a box with fewer devices must not print an N = 1 line labelled as 8: non-zero exit with the device count in the
    message; and a launcher whose WORLD_SIZE disagrees with --gpus is refused the same way"""
    import torch
    have = torch.cuda.device_count()
    for script, extra in (("bench.py", []), ("bench_circuits.py", ["decode"])):
        r = _direct(script, extra + ["--gpus", str(have + 7)])
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert "exposes %d HIP device" % have in r.stderr, r.stderr[-500:]
        r = _direct(script, extra + ["--gpus", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
        assert r.returncode != 0 and "WORLD_SIZE is 1" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_one_rank_process_group_runs_the_collectives_on_rccl():
    """no box this repository has seen exposes two devices, so the multi-rank RCCL transfers have never run; what CAN run is the
    collective code of bench.py on a ONE-rank RCCL communicator (FHE_BENCH_FORCE_DIST=1 under the launcher with one process):
    communicator initialisation on the device, barrier, MAX / SUM all-reduces and the all-gather on device tensors, the digest
    all-reduce (parallel.combine_digests) -- RCCL's kernels execute, and the line equals the plain one-process run"""
    plain = _bench(["--blocks", "128", "--steps", "2", "--warmup", "1", "--cpu-blocks", "0"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FHE_BENCH_FORCE_DIST="1")
    env.pop("FHE_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "128", "--steps", "2", "--warmup", "1", "--cpu-blocks", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    one = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert one["collective_backend"] == "rccl" and one["rccl_ranks"] == 1 and one["n_gpus"] == 1 and len(one["ms_per_step_per_rank"]) == 1
    assert one["output_digest"] == plain["output_digest"] and one["verified_bit_exact_vs_oracle"]


@pytest.mark.parametrize("fail", [False, True])
def test_rccl_probe_and_its_gloo_fall_back_under_the_launcher(fail):
    """bench.open_process_group: before the job's own RCCL initialisation every rank runs RCCL's rendezvous, an all-reduce, the send/recv ring and
    a closing verdict all-reduce in a CHILD process on its own store (bench.rccl_probe) -- the first RCCL contact between two devices will happen
    on the driver's node, and a fabric problem there must not hang the measurement.  On a one-GPU box the probe is forced (FHE_BENCH_PROBE_FORCE=1)
    on a one-rank job under the real launcher (agent store in the environment): it passes and the job runs on RCCL; with the child made to fail
    (FHE_BENCH_PROBE_FAIL=1) the SAME job continues on gloo, says so, and computes the same digest."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FHE_BENCH_FORCE_DIST="1", FHE_BENCH_PROBE_FORCE="1", FHE_BENCH_RCCL_PROBE_TIMEOUT="240")
    env.pop("FHE_BENCH_BACKEND", None)
    if fail:
        env["FHE_BENCH_PROBE_FAIL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "128", "--steps", "2", "--warmup", "1", "--cpu-blocks", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    one = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    probe = one["rccl_probe"]
    assert probe["ok"] == (not fail) and probe["seconds"] < 240
    if fail:
        assert one["collective_backend"].startswith("gloo (FALL-BACK") and "exit code 3" in probe["detail"]
    else:
        assert one["collective_backend"] == "rccl" and probe["detail"] is None
    assert one["rccl_ranks"] == 1 and one["verified_bit_exact_vs_oracle"] and one["value"] > 0
