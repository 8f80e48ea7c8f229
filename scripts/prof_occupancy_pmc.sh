#!/bin/bash
# Counters of pass B at 3840x2160x256 without a bound on the resident workgroups and with two per CU
# (SMHIP_OPT_B_WAVES 0 / 8): one rocprofv3 --pmc pass per setting.  GPU box: bash scripts/prof_occupancy_pmc.sh
set -u
export TMPDIR=/tmp
out=gpurun_out/prof_occ
mkdir -p $out
for s in 0 8; do
  timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $out/w$s -- python3 scripts/ab_fused.py --width 3840 --height 2160 --levels 256 --steps 6 --opts "{\"b_waves\":$s}" > $out/w$s.log 2>&1 || { echo "pass $s failed"; tail -5 $out/w$s.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for s in (0, 8):
    f = glob.glob('gpurun_out/prof_occ/w%d/*/*counter_collection.csv' % s)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'fused_boxv' in k or 'fused_rawbox' in k:
            acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print('b_waves', s, k, {c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())})
PY
