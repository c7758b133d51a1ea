#!/bin/bash
# round 6: the streaming server_resize end to end in the three modes (what the per-Cubic placement is for: a third of the output bytes)
cd "$GRAFT_REPO_ROOT"; export HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p gpurun_out
O=gpurun_out/r06_bench_server_resize_modes.txt; rm -f $O
for sh in "" "--shared"; do
  for mode in "" "--relin 30" "--relin 30 --relin-placement cubic" "--relin 60 --relin-placement cubic"; do
    python tools/bench_server_resize.py --encrypt device $sh $mode 2>/dev/null | tail -1 >> $O
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_bench_server_resize_modes.txt"):
    d=json.loads(l); print(d["offsets"][:10], "|", d["mode"][:45], "| %.3f s  dev %.3f s  out %.1f GB  write %.3f s" % (d["seconds"], d["device_compute_seconds"], d["stream_GB_out"], d["file_write_seconds"]))
PY
