set -e
cd /root/repo
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3e_tests.log 2>&1 || { tail -30 gpurun_out/r3e_tests.log; exit 1; }
tail -2 gpurun_out/r3e_tests.log
timeout -k 10 300 python -u scripts/bench_exp2.py > gpurun_out/r3_exp2.log 2>&1 || { tail -20 gpurun_out/r3_exp2.log; exit 1; }
cat gpurun_out/r3_exp2.log
