#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
SMHIP_FUZZ_ITER_N=800 timeout -k 10 900 python -u -m pytest tests/test_gpu_iter_fused.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_perm4_tests.log 2>&1 || { tail -30 gpurun_out/r3_perm4_tests.log; exit 1; }
tail -2 gpurun_out/r3_perm4_tests.log
echo "== all 512 levels"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 0 512 '{}' 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
echo "== levels 256..319"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 256 64 '{}' 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
