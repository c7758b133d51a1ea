#!/bin/bash
# usage: tools/prof.sh <name> <command...>   (on the GPU box)  -> gpurun_out/prof_<name>/ kernel stats as CSV
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- "$@" > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, sys, os
p = os.path.join(sys.argv[1], "p_kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
print("%-90s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:25]:
    print("%-90s %8s %12.1f %10.2f %6.2f" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
