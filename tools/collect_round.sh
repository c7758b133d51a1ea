#!/bin/bash
# Collects the round's measurements on the GPU box into gpurun_out/$1/ (copy what is quoted into profiles/).
# usage (via gpurun): tools/collect_round.sh r02
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/collect_traffic.py > $O/collect_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json
python bench.py --cpu-blocks 0 > $O/bench_default_with_traffic.json 2>/dev/null
for p in SEAL23_4096 SEAL3_8192 P8192; do python bench.py --preset $p --cpu-blocks 0 --steps 5 --blocks 512 > $O/bench_$p.json 2>/dev/null; done
python tools/bench_ops.py P4096 4096 > $O/bench_ops_P4096.txt 2>&1
python tools/bench_ops.py P8192 2048 > $O/bench_ops_P8192.txt 2>&1
python bench_circuits.py resize --shared --cpu-pixels 2 > $O/bench_circuits_resize.txt 2>&1
python bench_circuits.py decode > $O/bench_circuits_decode.txt 2>&1
python tools/bench_rgb.py > $O/bench_rgb.txt 2>&1
python tools/bench_server.py > $O/bench_server.txt 2>&1
python tools/bench_server_resize.py > $O/bench_server_resize.txt 2>&1
python tools/bench_server_resize.py >> $O/bench_server_resize.txt 2>&1
python tools/bench_server_resize.py --bilinear >> $O/bench_server_resize.txt 2>&1
tools/prof.sh ${1}_bench python $R/bench.py --cpu-blocks 0 --no-verify > $O/kernel_stats_bench_default.txt 2>&1
tools/prof.sh ${1}_resize python $R/bench_circuits.py resize > $O/kernel_stats_resize.txt 2>&1
tools/prof.sh ${1}_resize_shared python $R/bench_circuits.py resize --shared --max-pixels 256 > $O/kernel_stats_resize_shared.txt 2>&1
tools/prof.sh ${1}_decode python $R/bench_circuits.py decode > $O/kernel_stats_decode.txt 2>&1
tools/prof.sh ${1}_ops8192 python $R/tools/bench_ops.py P8192 2048 > $O/kernel_stats_ops_P8192.txt 2>&1
tools/prof.sh ${1}_seal23 python $R/bench.py --preset SEAL23_4096 --cpu-blocks 0 --no-verify --blocks 512 > $O/kernel_stats_bench_SEAL23_4096.txt 2>&1
for d in bench resize resize_shared decode ops8192 seal23; do cp $R/gpurun_out/prof_${1}_$d/p_kernel_stats.csv $O/kernel_stats_$d.csv 2>/dev/null; done
ls -la $O
