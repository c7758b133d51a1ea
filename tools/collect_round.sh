#!/bin/bash
# Collects the round's measurements on the GPU box into gpurun_out/$1/ (copy what is quoted into profiles/).
# usage (via gpurun): tools/collect_round.sh r04        -- tools/isa_counts.py $1 must have been run before (no GPU needed)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
# counters first: bench.py's issue_roofline and traffic fields read profiles/ (hash-matched to the running sources)
python tools/collect_counters.py $1 > $O/collect_counters.log 2>&1
cp gpurun_out/$1/counters.json profiles/${1}_counters.json 2>/dev/null; cp gpurun_out/$1/counters.txt $O/counters_summary.txt 2>/dev/null
python tools/issue_roofline.py $1 > $O/issue_roofline.txt 2>&1
python tools/collect_traffic.py > $O/collect_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json
python tools/collect_traffic.py ctct > $O/collect_traffic_ctct.log 2>&1; cp gpurun_out/pmc_traffic_ctct.json $O/pmc_traffic_ctct.json
FHE_BEHZ_FUSED_PREPARE=1 python tools/collect_traffic.py ctct > $O/collect_traffic_ctct_fused.log 2>&1; cp gpurun_out/pmc_traffic_ctct.json $O/pmc_traffic_ctct_fused_prepare.json
cp $O/pmc_traffic_ctct.json profiles/pmc_traffic_ctct.json
python tools/collect_traffic.py circuits > $O/collect_traffic_circuits.log 2>&1; cp gpurun_out/pmc_traffic_circuits.json $O/pmc_traffic_circuits.json
python tools/collect_traffic.py encrypt > $O/collect_traffic_encrypt.log 2>&1; cp gpurun_out/pmc_traffic_encrypt.json $O/pmc_traffic_encrypt.json
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --cpu-blocks 0 > $O/bench_default_with_traffic.json 2>/dev/null
python bench.py --cpu-blocks 0 --blocks 8192 --steps 3 --warmup 1 > $O/bench_config4_share_8192.json 2>/dev/null
python bench.py --cpu-blocks 0 --gather local --steps 3 --warmup 1 > $O/bench_gather_local.json 2>/dev/null
for p in SEAL23_4096 SEAL3_8192 P8192; do python bench.py --preset $p --cpu-blocks 0 --steps 5 --blocks 512 > $O/bench_$p.json 2>/dev/null; done
python tools/bench_ops.py P4096 4096 > $O/bench_ops_P4096.txt 2>&1
python tools/bench_ops.py P8192 2048 > $O/bench_ops_P8192.txt 2>&1
FHE_BEHZ_FUSED_PREPARE=1 python tools/bench_ops.py P8192 2048 > $O/bench_ops_P8192_fused_prepare.txt 2>&1
python tools/sweep_ctct_chunk.py P8192 1024 > $O/bench_ctct_chunks.txt 2>&1
python bench_circuits.py resize --cpu-pixels 4 > $O/bench_circuits_resize.json 2> /dev/null
python bench_circuits.py resize --shared --cpu-pixels 4 > $O/bench_circuits_resize_shared.json 2> /dev/null
python bench_circuits.py decode --cpu-terms 2 > $O/bench_circuits_decode.json 2> /dev/null
# the relinearised mode: the same workloads with evaluator.relinearize after every product, dbc 30 (the reference's unused DBC) and 60
for dbc in 30 60; do
  python bench_circuits.py resize --relin $dbc --relin-placement cubic --cpu-pixels 2 > $O/bench_circuits_resize_relin${dbc}_cubic.json 2> /dev/null
  python bench_circuits.py resize --shared --relin $dbc --relin-placement cubic > $O/bench_circuits_resize_shared_relin${dbc}_cubic.json 2> /dev/null
  python bench_circuits.py resize --relin $dbc --relin-placement sample --cpu-pixels 2 > $O/bench_circuits_resize_relin${dbc}_sample.json 2> /dev/null
  python bench_circuits.py resize --shared --relin $dbc --relin-placement sample > $O/bench_circuits_resize_shared_relin${dbc}_sample.json 2> /dev/null
  python bench_circuits.py resize --relin $dbc > $O/bench_circuits_resize_relin$dbc.json 2> /dev/null
  python bench_circuits.py resize --shared --relin $dbc > $O/bench_circuits_resize_shared_relin$dbc.json 2> /dev/null
  python bench_circuits.py decode --relin $dbc > $O/bench_circuits_decode_relin$dbc.json 2> /dev/null
done
python tools/bench_rgb.py > $O/bench_rgb.txt 2>&1
python tools/bench_server.py > $O/bench_server.txt 2>&1
python tools/bench_server_resize.py > $O/bench_server_resize.txt 2>&1
python tools/bench_server_resize.py --bilinear >> $O/bench_server_resize.txt 2>&1
python tools/bench_server_resize.py --encrypt device --shared >> $O/bench_server_resize.txt 2>&1
# the servers' own encryptions (csrc/encrypt.hip): per-ciphertext rates, and the two streaming servers with their encryptions made for real
python tools/bench_encrypt.py P8192 512 2>/dev/null | tail -1 > $O/bench_encrypt.txt
python tools/bench_encrypt.py P4096 512 2>/dev/null | tail -1 >> $O/bench_encrypt.txt
python tools/bench_encrypt.py P8192 8192 2>/dev/null | tail -1 >> $O/bench_encrypt.txt
for m in bank device host; do python tools/bench_server_resize.py --encrypt $m 2>/dev/null | tail -1 >> $O/bench_server_resize_encryptions.txt; done
# the streaming server end to end in every mode (records of 6 polynomials against 2)
for sh in "" "--shared"; do for mode in "" "--relin 30" "--relin 30 --relin-placement cubic" "--relin 60 --relin-placement cubic" "--relin 30 --relin-placement sample" "--relin 60 --relin-placement sample"; do
  python tools/bench_server_resize.py --encrypt device $sh $mode 2>/dev/null | tail -1 >> $O/bench_server_resize_modes.txt; done; done
for m in device host; do python tools/bench_server_decode.py --encrypt $m 2>/dev/null | tail -1 >> $O/bench_server_decode_encryptions.txt; done
fully-homomorphic-image-processing_amd/seal/multi_gpu_dct 1024 4 64 1 > $O/cpp_multi_gpu_dct.json 2>&1
fully-homomorphic-image-processing_amd/seal/multi_gpu_dct 1024 1 256 0 resident 20 > $O/cpp_multi_gpu_dct_resident.json 2>&1
# the one command the driver runs at N > 1, on this one device over gloo (schema + every leg; the physics needs N devices)
FHE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_world2_gloo_one_device.json 2>/dev/null
FHE_BENCH_BACKEND=gloo python bench_circuits.py resize --shared --gpus 2 > $O/bench_circuits_resize_shared_world2_gloo.json 2>/dev/null
python tools/results_table.py > $O/results_table.json 2>/dev/null
FHE_ENC_UNFUSED=1 python tools/bench_encrypt.py P8192 512 2>/dev/null | tail -1 | sed 's/^/UNFUSED /' >> $O/bench_encrypt.txt
FHE_ENC_OCC=4 python tools/bench_encrypt.py P8192 512 2>/dev/null | tail -1 | sed 's/^/OCC4 /' >> $O/bench_encrypt.txt
fully-homomorphic-image-processing_amd/seal/bench_resize > $O/bench_resize_cpp_host.txt 2>&1
python tools/run_ref_cli.py 48 48 --golden --pmod 3001 > $O/ref_cli_config0_lazy.txt 2>&1
FHE_FACADE_EAGER=1 python tools/run_ref_cli.py 48 48 --golden --pmod 3001 > $O/ref_cli_config0_eager.txt 2>&1
python tools/run_ref_resize.py bicubic 4096 101 > $O/ref_cli_resize.txt 2>&1
python tools/run_ref_resize.py bilinear 4096 101 >> $O/ref_cli_resize.txt 2>&1
FHE_FACADE_RELIN=30 python tools/run_ref_resize.py bicubic 4096 101 > $O/ref_cli_resize_relin30.txt 2>&1
# the last two columns of the reference's grid end to end (benchmark/benchmark.py:6)
for n in 8192 16384; do python oracle/pin_against_reference.py --gpu --n $n --pmod 101 1009 3001 --jobs 2 > $O/ref_grid_jpeg_$n.txt 2>&1; done
python oracle/pin_against_reference.py --gpu --n 16384 --resize bicubic --pmod 31 100003 --jobs 2 > $O/ref_grid_bicubic_16384.txt 2>&1
python oracle/pin_against_reference.py --gpu --n 16384 --resize bilinear --pmod 31 100003 --jobs 2 > $O/ref_grid_bilinear_16384.txt 2>&1
tools/prof.sh ${1}_bench python $R/bench.py --cpu-blocks 0 --no-verify > $O/kernel_stats_bench_default.txt 2>&1
tools/prof.sh ${1}_resize python $R/bench_circuits.py resize > $O/kernel_stats_resize.txt 2>&1
tools/prof.sh ${1}_resize_shared python $R/bench_circuits.py resize --shared > $O/kernel_stats_resize_shared.txt 2>&1
tools/prof.sh ${1}_decode python $R/bench_circuits.py decode > $O/kernel_stats_decode.txt 2>&1
tools/prof.sh ${1}_decode_relin30 python $R/bench_circuits.py decode --relin 30 > $O/kernel_stats_decode_relin30.txt 2>&1
tools/prof.sh ${1}_resize_shared_relin30 python $R/bench_circuits.py resize --shared --relin 30 > $O/kernel_stats_resize_shared_relin30.txt 2>&1
tools/prof.sh ${1}_ops8192 python $R/tools/bench_ops.py P8192 2048 > $O/kernel_stats_ops_P8192.txt 2>&1
tools/prof.sh ${1}_encrypt python $R/tools/bench_encrypt.py P8192 512 > $O/kernel_stats_encrypt.txt 2>&1
tools/prof.sh ${1}_seal23 python $R/bench.py --preset SEAL23_4096 --cpu-blocks 0 --no-verify --blocks 512 > $O/kernel_stats_bench_SEAL23_4096.txt 2>&1
for d in bench resize resize_shared decode decode_relin30 resize_shared_relin30 ops8192 seal23 encrypt; do cp $R/gpurun_out/prof_${1}_$d/p_kernel_stats.csv $O/kernel_stats_$d.csv 2>/dev/null; done
python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
python tools/soak.py > $O/soak.txt 2>&1
python tools/soak.py 320 P8192 >> $O/soak.txt 2>&1
python tools/soak.py 320 SEAL23_4096 >> $O/soak.txt 2>&1
ls -la $O
