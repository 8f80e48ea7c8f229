#!/bin/bash
# smhip_fused_iter_wta_keys with one level tile per workgroup and long segments (iter_direct): parity, then times
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_iter_fused.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_direct_tests.log 2>&1 || { tail -30 gpurun_out/r3_direct_tests.log; exit 1; }
tail -2 gpurun_out/r3_direct_tests.log
for o in '{}' '{"iter_direct":1}' '{"iter_direct":1,"tseg":512}' '{"iter_direct":1,"tseg":1088}'; do
echo "== all 512 levels $o"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 0 512 "$o" 2>&1 | grep -v amdgpu.ids || exit 1
done
for o in '{}' '{"iter_direct":1}' '{"iter_direct":1,"tseg":1088}'; do
echo "== levels 256..319 $o"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 256 64 "$o" 2>&1 | grep -v amdgpu.ids || exit 1
done
