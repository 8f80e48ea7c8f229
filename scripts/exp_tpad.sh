mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1; echo tests=$?; tail -3 gpurun_out/gputests.log
for pad in 0 64 4160; do
SMHIP_TPAD=$pad timeout -k 10 120 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$pad.json 2> gpurun_out/bench.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/bench_$pad.json')); print('pad $pad', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"
done
