#!/bin/bash
# usage (on the GPU box): tools/variant_ops.sh <tag> [lib.so]  -- per-op rates + fused u64 DCT rates with an alternative build of the library
R=$GRAFT_REPO_ROOT; tag=$1; lib=$2
[ -n "$lib" ] && cp $R/$lib $R/fully-homomorphic-image-processing_amd/libfhe_hip.so
cd $R
python tools/bench_ops.py P8192 2048 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('$tag', d.get('op'), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if k != 'op'})
"
for p in SEAL23_4096 P8192; do python bench.py --preset $p --cpu-blocks 0 --steps 5 --blocks 512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$p', round(d['value']), d['verified_bit_exact_vs_oracle'])"; done
