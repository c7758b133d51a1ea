#!/usr/bin/env python3
"""Measure HBM-side traffic of the fused DCT kernels with rocprofv3 PMC counters (run on the GPU box).

Separate --pmc passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2);
gfx950 correction: FETCH_SIZE tallies 128-byte read requests at 64 bytes, so reads are doubled;
WRITE_SIZE matched the known output byte count of k_dct_cols within 2% and is taken as is.
Both counters are in KiB.  Writes profiles/r01_pmc_traffic.json (read by bench.py).
"""
import csv
import json
import os
import subprocess
import sys
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = 128
WAVE = 128       # blocks per dispatch here: min(default FHE_DCT_WAVE_BLOCKS = 256, BLOCKS)
CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--blocks", str(BLOCKS), "--cpu-blocks", "0", "--no-verify"]


def run_pass(counters, tag):
    out = os.path.join(ROOT, "gpurun_out", "traffic_" + tag)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--pmc", *counters, "--"] + CMD,
                   check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(os.path.join(out, "p_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            name = "k_dct_rows" if "k_dct_rows" in k else "k_dct_cols" if "k_dct_cols" in k else None
            if name:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    rd = run_pass(["FETCH_SIZE"], "fetch")
    wr = run_pass(["WRITE_SIZE"], "write")
    per_kernel, total = {}, 0.0
    for k in ("k_dct_rows", "k_dct_cols"):
        fetch = rd[k]["FETCH_SIZE"] * 1024 * 2 / WAVE      # bytes per block, gfx950 x2 read correction
        write = wr[k]["WRITE_SIZE"] * 1024 / WAVE
        per_kernel[k] = {"read_bytes_per_block": fetch, "write_bytes_per_block": write,
                         "FETCH_SIZE_KiB_per_dispatch": rd[k]["FETCH_SIZE"], "WRITE_SIZE_KiB_per_dispatch": wr[k]["WRITE_SIZE"]}
        total += fetch + write
    res = {"hbm_bytes_per_block": total, "algorithmic_bytes_per_block": 25165824, "ratio_to_algorithmic": total / 25165824,
           "per_kernel": per_kernel, "blocks_per_dispatch": WAVE,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --blocks 128 --steps 2`, "
                     "FETCH_SIZE x2 (gfx950), per block; tools/collect_traffic.py"}
    with open(os.path.join(ROOT, "gpurun_out", "r01_pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
