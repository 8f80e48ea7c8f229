#!/bin/bash
# GPU box: parity tests, then the default bench (prints a short summary).
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1
rc=$?; echo tests=$rc; tail -4 gpurun_out/gputests.log
[ $rc -ne 0 ] && { grep -B30 "Error\|error" gpurun_out/gputests.log | head -60; exit 1; }
timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'])
for k in d['kernels']:
    print(' ', k['kernel'], k['avg_ms'], 'ms', k['achieved'], 'GB/s frac', k['frac'])
print('fused frac', d['roofline_fused']['frac'], d.get('cpu_baseline', {}).get('value'))
PY
