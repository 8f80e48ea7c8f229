#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
SMHIP_FUZZ_OPS_N=600 timeout -k 10 900 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_iter_fused.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_perm2_tests.log 2>&1 || { tail -30 gpurun_out/r3_perm2_tests.log; exit 1; }
tail -2 gpurun_out/r3_perm2_tests.log
timeout -k 10 300 python -u scripts/bench_binom.py 2>&1 | grep -v amdgpu.ids | tail -12
