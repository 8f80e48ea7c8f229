#!/bin/bash
# PMC passes over the stage operators (scripts/bench_ops.py): HBM bytes and VALU activity per kernel.
#   bash scripts/prof_ops_pmc.sh <tag> [W H D]
set -u
tag=${1:-ops}; shift
out=gpurun_out/prof_ops_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/bench_ops.py "$@" > $out/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $out/kt.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -- python3 scripts/bench_ops.py "$@" > $out/pmc1.log 2>&1 || { echo "pmc1 failed"; tail -5 $out/pmc1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc2 -- python3 scripts/bench_ops.py "$@" > $out/pmc2.log 2>&1 || { echo "pmc2 failed"; tail -5 $out/pmc2.log; exit 1; }
python3 scripts/prof_summary.py $out > $out/summary.txt
cat $out/kt.log | tail -30
