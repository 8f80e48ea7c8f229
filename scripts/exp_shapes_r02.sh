# fused path on shapes other than config 3 (one line each: args, G px*disp/s, ms/step, per-kernel (ms, frac), pair frac)
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 "$@" > gpurun_out/b.json 2> gpurun_out/bench.err || { tail -3 gpurun_out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$*', d['value'], d['ms_per_step'], [(k['kernel'][:16], k['avg_ms'], k['frac']) for k in d['kernels']], d['roofline_fused']['frac'])"; }
mkdir -p gpurun_out
run &&
run --window 3 &&
run --window 5 &&
run --window 7 &&
run --window 11 &&
run --window 13 &&
run --window 15 &&
run --window 17 &&
run --bands 4 &&
run --width 3840 --height 2160 --disp-total 512 &&
run --width 7680 --height 4320 --disp-total 128 &&
run --width 7680 --height 4320 --disp-total 256 &&
# (added at the end of round 2, with the occupancy policy)
run --width 3840 --height 2160 --disp-total 256 &&
run --width 3840 --height 2160 --disp-total 128 &&
run --width 1920 --height 1080 --disp-total 512 &&
run --width 1920 --height 1080 --disp-total 128 &&
run --width 2560 --height 1440 --disp-total 256
