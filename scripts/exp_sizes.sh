run() { timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4
run
run --window 7
run --window 11
run --window 17
run --bands 4
run --width 7680 --height 4320 --disp-per-gpu 128
