# 128-level slabs: row-pair pass B against the general pass B (pair_b option) -- A/B via scripts/ab_fused.py style
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:22], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
run --width 7680 --height 4320 --disp-total 128 &&
run --width 3840 --height 2160 --disp-total 128 &&
run --width 1920 --height 1080 --disp-total 128
