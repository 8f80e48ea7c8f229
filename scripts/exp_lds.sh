mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1; echo tests=$?; tail -3 gpurun_out/gputests.log
run() { env "$@" timeout -k 10 120 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
run X=0
run SMHIP_LDS_A=32768
run SMHIP_LDS_A=40000
run SMHIP_LDS_B=65536
run SMHIP_LDS_B=81920
run SMHIP_LDS_A=32768 SMHIP_LDS_B=65536
