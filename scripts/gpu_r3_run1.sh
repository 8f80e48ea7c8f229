set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 120 --timeout-method thread -k "binomial or multi_iteration or tile_kernels" > gpurun_out/r3_t1.log 2>&1 || { tail -30 gpurun_out/r3_t1.log; exit 1; }
tail -3 gpurun_out/r3_t1.log
timeout -k 10 300 python -u scripts/bench_binom.py > gpurun_out/r3_binom1.log 2>&1 || { tail -30 gpurun_out/r3_binom1.log; exit 1; }
cat gpurun_out/r3_binom1.log
