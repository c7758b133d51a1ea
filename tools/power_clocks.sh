#!/bin/bash
# rocm-smi power and clock samples (once per second) while a command runs on the GPU box.
# usage (via gpurun): tools/power_clocks.sh <command ...>      default: the headline pair for ~30 s
R=$GRAFT_REPO_ROOT
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/.*sclk clock level: //; s/.*Power (W): /W=/' | tr '\n' ' '; echo; }
if [ $# -eq 0 ]; then set -- python $R/bench.py --cpu-blocks 0 --no-verify --steps 2500 --warmup 2; fi
echo "# $*"
echo "idle: $(sample)"
"$@" > /tmp/power_clocks_cmd.out 2>/dev/null &
BP=$!
for i in $(seq 1 120); do echo "t=$i $(sample)"; sleep 1; kill -0 $BP 2>/dev/null || break; done
wait $BP
tail -c 600 /tmp/power_clocks_cmd.out | cut -c1-400
