mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1; rc=$?; echo tests=$rc; tail -3 gpurun_out/gputests.log
[ $rc -ne 0 ] && { grep -B5 -A25 "^____\|Error" gpurun_out/gputests.log | head -80; exit 1; }
run() { env "$@" timeout -k 10 120 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
run SMHIP_TYSHIFT=0
run SMHIP_TYSHIFT=2
run SMHIP_TYSHIFT=3
run SMHIP_TYSHIFT=4
run SMHIP_TYSHIFT=5
run SMHIP_TYSHIFT=3
