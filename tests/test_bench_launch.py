"""CPU: `python bench.py --gpus N` is the whole multi-GPU launch (bench.ensure_world) -- the refusals that need no device:
more ranks than HIP devices, and a launcher whose WORLD_SIZE disagrees with --gpus, exit non-zero WITHOUT printing a bench line
(round 4's bench.py parsed --gpus and never read it: `--gpus 8` printed an N = 1 line)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, args, **extra_env):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FHE_BENCH_BACKEND"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)


@pytest.mark.parametrize("script,extra", [("bench.py", []), ("bench_circuits.py", ["decode"])])
def test_gpus_flag_is_checked_against_launcher_and_devices(script, extra):
    import torch
    have = torch.cuda.device_count()
    r = _run(script, extra + ["--gpus", str(have + 7)])
    assert r.returncode != 0 and "exposes %d HIP device" % have in r.stderr, r.stderr[-800:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = _run(script, extra + ["--gpus", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE is 1" in r.stderr, r.stderr[-800:]
    r = _run(script, extra + ["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE is 2" in r.stderr, r.stderr[-800:]
    r = _run(script, extra + ["--gpus", "0"])
    assert r.returncode != 0
    if have == 0:                           # without a device the one-rank run refuses too: there is no CPU path
        r = _run(script, extra + ["--gpus", "1"])
        assert r.returncode != 0 and "needs a HIP device" in r.stderr


def test_rccl_probe_host_logic_without_a_device(monkeypatch):
    """bench.rccl_probe / backend_label: the child program parses; a child that exits before touching a device is reported as a failed probe with
    its exit code (the fall-back branch of open_process_group); a child that does not finish is killed by its own handle at the limit; the
    children get their own store port and never the launcher's agent store"""
    sys.path.insert(0, ROOT)
    import bench
    compile(bench._RCCL_PROBE_CHILD, "rccl_probe_child", "exec")
    monkeypatch.setenv("FHE_BENCH_PROBE_FAIL", "1")
    monkeypatch.setenv("MASTER_PORT", "65530")
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    rec = bench.rccl_probe(5.0)
    assert rec["ok"] is False and "exit code 3" in rec["detail"] and rec["seconds"] < 30
    seen = {}
    real = subprocess.Popen

    class Hung(real):                                       # a child that never finishes: sleeps instead of probing
        def __init__(self, cmd, **kw):
            seen["env"] = kw["env"]
            super().__init__([sys.executable, "-c", "import time; time.sleep(600)"], **kw)
    monkeypatch.setattr(subprocess, "Popen", Hung)
    rec = bench.rccl_probe(-29.0)                           # limit + 30 = 1 s
    assert rec["ok"] is False and "no completion" in rec["detail"] and rec["seconds"] < 20
    assert seen["env"]["MASTER_PORT"] == "65501" and "TORCHELASTIC_USE_AGENT_STORE" not in seen["env"]      # 65530 + 29 would leave the port range
    assert bench.backend_label("nccl", "nccl", False) == "rccl"
    assert bench.backend_label("gloo", "nccl", False).startswith("gloo (FALL-BACK")
    assert "test mode" in bench.backend_label("gloo", "gloo", True)
