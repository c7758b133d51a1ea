#!/bin/bash
# round 6, GPU call: the whole -m gpu suite (timed), the RESULTS.md measurements, the encryption traffic counters
set -x
cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r06_pytest_gpu.txt 2>&1; tail -8 gpurun_out/r06_pytest_gpu.txt
python tools/results_table.py > gpurun_out/r06_results_table.json 2> gpurun_out/r06_results_table.err; echo rc=$?; tail -3 gpurun_out/r06_results_table.err
cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/collect_traffic.py encrypt > $GRAFT_REPO_ROOT/gpurun_out/r06_collect_traffic_encrypt.log 2>&1; echo rc=$?; tail -c 600 $GRAFT_REPO_ROOT/gpurun_out/r06_collect_traffic_encrypt.log
