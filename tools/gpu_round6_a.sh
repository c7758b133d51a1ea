#!/bin/bash
# round 6, first GPU call: the new bench.py record (N = 1 line; world-2 over gloo on one device with both gather legs) and the new tests
set -x
cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
python -c "import torch; print('devices', torch.cuda.device_count())"
python bench.py --steps 5 --warmup 2 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo rc=$?; tail -c 600 gpurun_out/r06_bench_n1.err
FHE_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r06_bench_gloo2.json 2> gpurun_out/r06_bench_gloo2.err; echo rc=$?; tail -c 1500 gpurun_out/r06_bench_gloo2.err
python -m pytest tests/test_gpu_parity.py -x -q -k "rgb or error_codes" 2>&1 | tail -5
python -m pytest tests/test_gpu_encrypt.py tests/test_gpu_multi.py -x -q 2>&1 | tail -8
