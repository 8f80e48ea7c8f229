# 4-band frames: pass A stages 16 bytes per pixel of the row in LDS
run() { timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
mkdir -p gpurun_out
run --bands 4
run --bands 4 --width 3840 --height 2160 --disp-per-gpu 256
run --bands 4 --width 7680 --height 4320 --disp-per-gpu 128
run --bands 4 --width 7680 --height 4320 --disp-per-gpu 256
