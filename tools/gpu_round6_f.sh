#!/bin/bash
# round 6: the per-sample relinearisation placement -- parity, configs[2] lines, the streaming server
cd "$GRAFT_REPO_ROOT"; export HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p gpurun_out
python -m pytest tests/test_gpu_relin.py tests/test_reference_vs_batched_cpp.py -x -q -s 2>&1 | grep -v "homomorphic_sin(" | tail -14
for dbc in 60 30; do
  python bench_circuits.py resize --relin $dbc --relin-placement sample --cpu-pixels 2 > gpurun_out/r06_bc3_resize_relin${dbc}_sample.json 2>/dev/null; echo rc=$?
  python bench_circuits.py resize --shared --relin $dbc --relin-placement sample > gpurun_out/r06_bc3_resize_shared_relin${dbc}_sample.json 2>/dev/null; echo rc=$?
done
python bench_circuits.py resize > gpurun_out/r06_bc3_resize.json 2>/dev/null
python bench_circuits.py resize --shared > gpurun_out/r06_bc3_resize_shared.json 2>/dev/null
O=gpurun_out/r06_bench_server_resize_modes_sample.txt; rm -f $O
for sh in "" "--shared"; do
  for mode in "" "--relin 60 --relin-placement sample" "--relin 30 --relin-placement sample" "--relin 60 --relin-placement cubic"; do
    python tools/bench_server_resize.py --encrypt device $sh $mode 2>/dev/null | tail -1 >> $O
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_bc3_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f.split("bc3_")[1], "%.1f ms"%d["ms_per_step"], d["out_size"])
for l in open("gpurun_out/r06_bench_server_resize_modes_sample.txt"):
    d=json.loads(l); print(d["offsets"][:10], "|", d["mode"][:48], "| %.3f s  dev %.3f s  out %.1f GB" % (d["seconds"], d["device_compute_seconds"], d["stream_GB_out"]))
PY
