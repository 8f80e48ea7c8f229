#!/bin/bash
# Per-kernel times of the stage operators (scripts/bench_ops.py) under rocprofv3.  GPU box:
#   bash scripts/prof_ops.sh <tag> [W H D]
set -u
tag=${1:-ops}; shift
out=gpurun_out/prof_ops_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/bench_ops.py "$@" > $out/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $out/kt.log; exit 1; }
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
cp "$f" $out/kernel_stats.csv
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls %4s avg %10.3f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
