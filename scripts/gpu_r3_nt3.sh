#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_zx_gpu_bench_sharded.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_nt3_tests.log 2>&1 || { tail -30 gpurun_out/r3_nt3_tests.log; exit 1; }
tail -2 gpurun_out/r3_nt3_tests.log
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:16], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
run --width 7680 --height 4320 --disp-total 256 &&
run --width 7680 --height 4320 --disp-total 128 &&
run --width 7680 --height 4320 --disp-total 512 &&
MASTER_ADDR=127.0.0.1 timeout -k 10 300 python bench.py --gpus 1 --one-rank-rehearsal --steps 10 --warmup 3 > gpurun_out/r3_nt3_rehearsal.json 2> gpurun_out/r3_nt3_rehearsal.err || { tail -5 gpurun_out/r3_nt3_rehearsal.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3_nt3_rehearsal.json')); print('one-rank rehearsal', d['value'], d['ms_per_step'], d['per_rank_compute_ms'])"
