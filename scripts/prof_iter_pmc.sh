#!/bin/bash
# PMC passes over smhip_fused_iter_wta_keys (scripts/bench_iter.py): VALU / LDS activity of the fused kernels.
#   bash scripts/prof_iter_pmc.sh <tag> [bench_iter args]
set -u
tag=${1:-iter}; shift
out=gpurun_out/prof_iter_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/bench_iter.py "$@" > $out/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $out/kt.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -- python3 scripts/bench_iter.py "$@" > $out/pmc1.log 2>&1 || { echo "pmc1 failed"; tail -5 $out/pmc1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc2 -- python3 scripts/bench_iter.py "$@" > $out/pmc2.log 2>&1 || { echo "pmc2 failed"; tail -5 $out/pmc2.log; exit 1; }
python3 scripts/prof_summary.py $out > $out/summary.txt
grep -E "iter_px|calls" $out/summary.txt
