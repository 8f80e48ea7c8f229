#!/bin/bash
# iter_px_kernel with the pixel order (0, 2, 1, 3), DPP moves without fallback registers, folded WTA selects
set -o pipefail
mkdir -p gpurun_out
SMHIP_FUZZ_ITER_N=600 SMHIP_FUZZ_OPS_N=400 timeout -k 10 900 python -u -m pytest tests/test_gpu_iter_fused.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_host_driver.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_perm_tests.log 2>&1 || { tail -30 gpurun_out/r3_perm_tests.log; exit 1; }
tail -2 gpurun_out/r3_perm_tests.log
echo "== all 512 levels"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 0 512 '{}' 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
echo "== levels 256..319"
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 256 64 '{}' 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
echo "== stage operators (diffusion_run px form, binomial forms)"
timeout -k 10 300 python -u scripts/bench_binom.py 2>&1 | grep -v amdgpu.ids | tail -12
