# which of (frame size, levels per slab) costs the config-5 slab its efficiency?
run() { timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
mkdir -p gpurun_out
run --width 1920 --height 1080 --disp-per-gpu 256
run --width 1920 --height 1080 --disp-per-gpu 128
run --width 3840 --height 2160 --disp-per-gpu 128
run --width 3840 --height 2160 --disp-per-gpu 256
run --width 7680 --height 4320 --disp-per-gpu 128
run --width 7680 --height 4320 --disp-per-gpu 256
run --width 7680 --height 1080 --disp-per-gpu 128
run --width 1920 --height 4320 --disp-per-gpu 128
