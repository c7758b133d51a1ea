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
