#!/bin/bash
# End-of-round evidence: the full GPU suite and the driver's bench command (scripts/gpu_full.sh), a kernel trace of
# that same command, then scripts/profile.sh (kernel trace + three --pmc passes of the reduced command + summary).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_full.sh r03_end || exit 1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_end_kt -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_end_kt_bench.json 2> gpurun_out/r03_end_kt.err || { echo "kernel trace of the driver's command failed"; tail -5 gpurun_out/r03_end_kt.err; exit 1; }
find gpurun_out/r03_end_kt -name "*kernel_stats.csv" | head -3
bash scripts/profile.sh r03_end
