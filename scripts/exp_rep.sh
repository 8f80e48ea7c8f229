run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
mkdir -p gpurun_out
run; run; run --steps 50; run
