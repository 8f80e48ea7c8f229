#!/bin/bash
# pass A with the staged matching row split by position parity (conflict-free ds_read_b64): parity, times, conflicts
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_split_tests.log 2>&1 || { tail -30 gpurun_out/r3_split_tests.log; exit 1; }
tail -2 gpurun_out/r3_split_tests.log
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:16], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
run && run && run --window 5 && run --window 11 &&
run --width 3840 --height 2160 --disp-total 256 &&
run --width 7680 --height 4320 --disp-total 128 &&
run --width 1920 --height 1080 --disp-total 128 &&
run --width 3840 --height 2160 --disp-total 512 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r3_split_pmc -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --sustained-seconds 0 > gpurun_out/r3_split_pmc.log 2>&1 || { echo pmc failed; tail -5 gpurun_out/r3_split_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3_split_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "fused" in k:
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
