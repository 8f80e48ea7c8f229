#!/bin/bash
# Kernel trace + PMC passes of the default bench workload.  Run on the GPU box:
#   bash scripts/profile.sh <tag>
# Outputs under gpurun_out/prof_<tag>/; copy the summaries you keep into profiles/.
set -u
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $BENCH > $out/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $out/kt.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc1 -- $BENCH > $out/pmc1.log 2>&1 || { echo "pmc1 failed"; tail -5 $out/pmc1.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc2 -- $BENCH > $out/pmc2.log 2>&1 || { echo "pmc2 failed"; tail -5 $out/pmc2.log; exit 1; }
# pass 3: where pass A's stores wait (L2 write path / texture addresser); counter names as rocprofv3 -L lists them
timeout -s KILL 180 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_64B_sum TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc3 -- $BENCH > $out/pmc3.log 2>&1 || { echo "pmc3 failed (counter names?)"; tail -5 $out/pmc3.log; }
python3 scripts/prof_summary.py $out
