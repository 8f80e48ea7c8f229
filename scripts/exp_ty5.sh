# tyshift (rows per band of the slab T) at the config-5 per-GPU shape: 7680x4320, 128 levels
run() { env "$@" timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --width 7680 --height 4320 --disp-per-gpu 128 > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
mkdir -p gpurun_out
run SMHIP_TYSHIFT=3
run SMHIP_TYSHIFT=4
run SMHIP_TYSHIFT=5
run SMHIP_TYSHIFT=6
run SMHIP_TYSHIFT=7
run SMHIP_TYSHIFT=8
