#!/bin/bash
R=$GRAFT_REPO_ROOT
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/.*sclk clock level: //; s/.*Power (W): /W=/' | tr '\n' ' '; echo; }
echo "idle: $(sample)"
python $R/bench.py --cpu-blocks 0 --no-verify --steps 2500 --warmup 2 > /tmp/b.json 2>/dev/null &
BP=$!
for i in $(seq 1 60); do echo "t=$i $(sample)"; sleep 1; kill -0 $BP 2>/dev/null || break; done
wait $BP
python -c "import json; d=json.load(open('/tmp/b.json')); print('bench', round(d['value']), 'blocks/s over', d['steps'], 'steps')"
