#!/bin/bash
# round 6, third GPU call: the fused encryption kernel (parity + rate, both paths), the one-pass key switch, and a probe of the
# reference's benchmark grid at n = 8192 / 16384 (time, disk, memory of ONE set each) before it goes into the suite
set -x
cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
df -h /tmp | tail -1; free -g | head -2; nproc
python -m pytest tests/test_gpu_encrypt.py tests/test_gpu_relin.py -x -q 2>&1 | tail -6
for p in P8192 P4096 SEAL23_4096; do
  python tools/bench_encrypt.py $p 512 2>/dev/null | tail -1 | tee -a gpurun_out/r06_bench_encrypt.txt
  FHE_ENC_UNFUSED=1 python tools/bench_encrypt.py $p 512 2>/dev/null | tail -1 | sed 's/^/UNFUSED /' | tee -a gpurun_out/r06_bench_encrypt.txt
done
python bench_circuits.py resize --relin 30 --relin-placement cubic > gpurun_out/r06_bc2_resize_relin30_cubic.json 2>/dev/null; echo rc=$?
FHE_RELIN_STEPS=1 python bench_circuits.py resize --relin 30 --relin-placement cubic > gpurun_out/r06_bc2_resize_relin30_cubic_steps.json 2>/dev/null; echo rc=$?
python bench_circuits.py resize --relin 60 --relin-placement cubic > gpurun_out/r06_bc2_resize_relin60_cubic.json 2>/dev/null; echo rc=$?
for n in 8192 16384; do
  /usr/bin/time -v python oracle/pin_against_reference.py --gpu --n $n --pmod 3001 --jobs 1 2> gpurun_out/r06_grid_jpeg_$n.time | tee gpurun_out/r06_grid_jpeg_$n.txt
  grep -E "Elapsed|Maximum resident" gpurun_out/r06_grid_jpeg_$n.time
done
/usr/bin/time -v python oracle/pin_against_reference.py --gpu --n 16384 --resize bicubic --pmod 31 --jobs 1 2> gpurun_out/r06_grid_bicubic_16384.time | tee gpurun_out/r06_grid_bicubic_16384.txt
grep -E "Elapsed|Maximum resident" gpurun_out/r06_grid_bicubic_16384.time
/usr/bin/time -v python oracle/pin_against_reference.py --gpu --n 16384 --resize bilinear --pmod 100003 --jobs 1 2> gpurun_out/r06_grid_bilinear_16384.time | tee gpurun_out/r06_grid_bilinear_16384.txt
grep -E "Elapsed|Maximum resident" gpurun_out/r06_grid_bilinear_16384.time
