#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
SMHIP_FUZZ_ITER_N=800 timeout -k 10 900 python -u -m pytest tests/test_gpu_iter_fused.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_host_driver.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_direct3_tests.log 2>&1 || { tail -30 gpurun_out/r3_direct3_tests.log; exit 1; }
tail -2 gpurun_out/r3_direct3_tests.log
echo "== all 512 levels, default"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 0 512 '{}' 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
echo "== levels 256..319, default"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 256 64 '{}' 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
echo "== 1920x1080x256, default"
timeout -k 10 300 python -u scripts/bench_iter.py 1920 1080 256 0 256 '{}' 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
echo "== 7680x4320x1024 levels 0..127, default / not direct"
timeout -k 10 300 python -u scripts/bench_iter.py 7680 4320 1024 0 128 '{}' 2>&1 | grep "binomial x4\|diffusion x6" | cut -c1-100 || exit 1
timeout -k 10 300 python -u scripts/bench_iter.py 7680 4320 1024 0 128 '{"iter_direct":0}' 2>&1 | grep "binomial x4\|diffusion x6" | cut -c1-100 || exit 1
