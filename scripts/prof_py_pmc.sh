#!/bin/bash
# Kernel trace + three PMC passes over any GPU script of this repo:
#   bash scripts/prof_py_pmc.sh <tag> scripts/bench_exp2.py [args]
# Results: gpurun_out/prof_<tag>/{kt,pmc1,pmc2,pmc3}, summary.txt, summary.json, condensed.json
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
run() { # name, counters...
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d $out/$name -- python3 "${PY[@]}" > $out/$name.log 2>&1 || { echo "$name failed"; tail -5 $out/$name.log; exit 1; }
}
PY=("$@")
run kt --kernel-trace --stats
run pmc1 --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run pmc2 --pmc WRITE_SIZE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY
run pmc3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_ATOMIC_sum TCC_REQ_sum TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
python3 scripts/prof_summary.py $out > $out/summary.txt
python3 scripts/prof_ops_summary.py $out $out/condensed.json
