#!/bin/bash
# pass B with the wave minimum per group of rows through LDS (smhip_fused_b.h, LRED): parity, then narrow shapes both ways
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_fuzz.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_lred_tests.log 2>&1 || { tail -30 gpurun_out/r3_lred_tests.log; exit 1; }
tail -3 gpurun_out/r3_lred_tests.log
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:16], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
for lred in 0 1; do
run --width 1920 --height 1080 --disp-total 128 --option b_lred=$lred &&
run --width 1920 --height 1080 --disp-total 64 --option b_lred=$lred &&
run --width 1280 --height 720 --disp-total 128 --option b_lred=$lred &&
run --width 2560 --height 1440 --disp-total 128 --option b_lred=$lred &&
run --width 3072 --height 1728 --disp-total 128 --option b_lred=$lred &&
run --width 1920 --height 1080 --disp-total 128 --window 5 --option b_lred=$lred &&
run --width 1920 --height 1080 --disp-total 128 --window 11 --option b_lred=$lred || exit 1
done
