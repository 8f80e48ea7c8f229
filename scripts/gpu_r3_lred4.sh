#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:16], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
run --width 3328 --height 1872 --disp-total 64 --option b_lred=1 &&
run --width 3840 --height 2160 --disp-total 64 --option b_lred=1 &&
run --width 4096 --height 2304 --disp-total 64 --option b_lred=1 &&
run --width 4096 --height 2304 --disp-total 64 --option b_lred=0 &&
run --width 3072 --height 1728 --disp-total 64 &&
run --width 1920 --height 1080 --disp-total 64 &&
run --width 3072 --height 1728 --disp-total 128 --option b_lred=1
