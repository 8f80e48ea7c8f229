#!/bin/bash
# one-wave pass A workgroups on the large frames (the LDS floor of one-wave workgroups: five / six per CU)
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:16], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
for nw in 0 1; do
run --width 3840 --height 2160 --disp-total 256 --option a_nw=$nw &&
run --width 7680 --height 4320 --disp-total 256 --option a_nw=$nw &&
run --width 3840 --height 2160 --disp-total 512 --option a_nw=$nw &&
run --width 2560 --height 1440 --disp-total 256 --option a_nw=$nw --option a_lds_kb=26 &&
run --width 3840 --height 2160 --disp-total 256 --option a_nw=$nw --option a_lds_kb=31 || exit 1
done
