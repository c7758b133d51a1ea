#!/bin/bash
# round 6, second GPU call: per-Cubic relinearisation (parity + configs[2] timings in all three modes), the SEAL cross-check on
# libfhe_hip.so, the C++ host's resident mode, bench_circuits' world-2 record over gloo
set -x
cd "$GRAFT_REPO_ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_relin.py tests/test_seal_crosscheck.py tests/test_gpu_sharding.py tests/test_gpu_multi.py -x -q -m gpu -s 2>&1 | tail -25
python -m pytest tests/test_gpu_parity.py -x -q -k "relinearize or rgb or error_codes" 2>&1 | tail -4
for mode in "" "--relin 30" "--relin 30 --relin-placement cubic" "--relin 60 --relin-placement cubic"; do
  tag=$(echo "resize$mode" | tr -d ' -' )
  python bench_circuits.py resize $mode --cpu-pixels 2 > gpurun_out/r06_bc_$tag.json 2> gpurun_out/r06_bc_$tag.err; echo rc=$?; tail -c 300 gpurun_out/r06_bc_$tag.err
  python bench_circuits.py resize --shared $mode > gpurun_out/r06_bc_shared_$tag.json 2> gpurun_out/r06_bc_shared_$tag.err; echo rc=$?
done
FHE_BENCH_BACKEND=gloo python bench_circuits.py resize --shared --gpus 2 > gpurun_out/r06_bc_shared_gloo2.json 2> gpurun_out/r06_bc_shared_gloo2.err; echo rc=$?; tail -c 400 gpurun_out/r06_bc_shared_gloo2.err
FHE_BENCH_BACKEND=gloo python bench_circuits.py decode --gpus 2 > gpurun_out/r06_bc_decode_gloo2.json 2> gpurun_out/r06_bc_decode_gloo2.err; echo rc=$?
fully-homomorphic-image-processing_amd/seal/multi_gpu_dct 1024 1 256 0 resident 20 > gpurun_out/r06_cpp_multi_gpu_dct_resident.json; cat gpurun_out/r06_cpp_multi_gpu_dct_resident.json
fully-homomorphic-image-processing_amd/seal/multi_gpu_dct 1024 1 64 0 verify > gpurun_out/r06_cpp_multi_gpu_dct_verify.json; cat gpurun_out/r06_cpp_multi_gpu_dct_verify.json
