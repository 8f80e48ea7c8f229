#!/bin/bash
# LRED pass B on frames wider than 2048 columns, with the round-filling key buffer
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:16], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
for lred in 0 1; do
run --width 2560 --height 1440 --disp-total 128 --option b_lred=$lred &&
run --width 3072 --height 1728 --disp-total 128 --option b_lred=$lred &&
run --width 3520 --height 1980 --disp-total 128 --option b_lred=$lred &&
run --width 2240 --height 1260 --disp-total 128 --option b_lred=$lred &&
run --width 2560 --height 1440 --disp-total 64 --option b_lred=$lred &&
run --width 3840 --height 2160 --disp-total 64 --option b_lred=$lred &&
run --width 3840 --height 2160 --disp-total 128 --option paired=0 --option b_lred=$lred || exit 1
done
