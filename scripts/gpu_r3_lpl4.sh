set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_store_hazard.py -x -q --timeout 300 --timeout-method thread -k "config3_full_size or store_hazard" > gpurun_out/r3_lpl4_t.log 2>&1 || { tail -30 gpurun_out/r3_lpl4_t.log; exit 1; }
tail -3 gpurun_out/r3_lpl4_t.log
timeout -k 10 200 python scripts/ab_fused.py --steps 60 --opts '{}' '{"a_nw": 4}' '{"a_nw": 1}' '{"a_nw": 4, "store_policy": 0}' '{"a_nw": 4, "store_policy": 18}' '{}' > gpurun_out/r3_lpl4.log 2>&1 || { tail -20 gpurun_out/r3_lpl4.log; exit 1; }
cat gpurun_out/r3_lpl4.log
timeout -k 10 200 python scripts/ab_fused.py --width 3840 --height 2160 --steps 20 --opts '{}' '{"a_nw": 4}' '{}' '{"a_nw": 4}' >> gpurun_out/r3_lpl4.log 2>&1
tail -8 gpurun_out/r3_lpl4.log
