set -e
cd /root/repo
timeout -k 10 900 python -u -m pytest tests/test_gpu_iter_fused.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_iter_t.log 2>&1 || { tail -40 gpurun_out/r3_iter_t.log; exit 1; }
tail -3 gpurun_out/r3_iter_t.log
timeout -k 10 300 python -u scripts/bench_iter.py > gpurun_out/r3_iter_b.log 2>&1 || { tail -30 gpurun_out/r3_iter_b.log; exit 1; }
timeout -k 10 300 python -u scripts/bench_iter.py 3840 2160 512 0 512 >> gpurun_out/r3_iter_b.log 2>&1 || { tail -30 gpurun_out/r3_iter_b.log; exit 1; }
cat gpurun_out/r3_iter_b.log
