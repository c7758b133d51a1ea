#!/bin/bash
# round 6, GPU call: the fused encryption kernel (parity + rate: default 4 waves/SIMD build, FHE_ENC_OCC=2, the five launches) and a probe
# of the reference's benchmark grid at n = 8192 / 16384 (ONE set each: does it pass, how long) before it goes into the suite
set -x
cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encrypt.py -x -q 2>&1 | tail -4
rm -f gpurun_out/r06_bench_encrypt.txt
for p in P8192 P4096 SEAL23_4096; do
  python tools/bench_encrypt.py $p 512 2>/dev/null | tail -1 | tee -a gpurun_out/r06_bench_encrypt.txt | cut -c1-330
  FHE_ENC_OCC=2 python tools/bench_encrypt.py $p 512 2>/dev/null | tail -1 | sed 's/^/OCC2 /' | tee -a gpurun_out/r06_bench_encrypt.txt | cut -c1-330
  FHE_ENC_UNFUSED=1 python tools/bench_encrypt.py $p 512 2>/dev/null | tail -1 | sed 's/^/UNFUSED /' | tee -a gpurun_out/r06_bench_encrypt.txt | cut -c1-330
done
python tools/bench_encrypt.py P8192 8192 2>/dev/null | tail -1 | tee -a gpurun_out/r06_bench_encrypt.txt | cut -c1-330
for n in 8192 16384; do
  ( time python oracle/pin_against_reference.py --gpu --n $n --pmod 3001 --jobs 1 ) > gpurun_out/r06_grid_jpeg_$n.txt 2>&1; tail -6 gpurun_out/r06_grid_jpeg_$n.txt | cut -c1-300
done
( time python oracle/pin_against_reference.py --gpu --n 16384 --resize bicubic --pmod 31 --jobs 1 ) > gpurun_out/r06_grid_bicubic_16384.txt 2>&1; tail -6 gpurun_out/r06_grid_bicubic_16384.txt | cut -c1-300
( time python oracle/pin_against_reference.py --gpu --n 16384 --resize bilinear --pmod 100003 --jobs 1 ) > gpurun_out/r06_grid_bilinear_16384.txt 2>&1; tail -6 gpurun_out/r06_grid_bilinear_16384.txt | cut -c1-300
df -h /tmp | tail -1
