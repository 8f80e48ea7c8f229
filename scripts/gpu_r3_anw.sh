set -e
cd /root/repo
bash scripts/gpu_full.sh r3b
timeout -k 10 200 python scripts/ab_fused.py --steps 60 --opts '{}' '{"a_nw": 1}' '{"a_nw": 1, "a_lds_kb": 26}' '{"a_nw": 1, "a_lds_kb": 31}' '{"a_nw": 1, "a_lds_kb": 20}' '{}' > gpurun_out/r3_anw.log 2>&1 || { tail -20 gpurun_out/r3_anw.log; exit 1; }
cat gpurun_out/r3_anw.log
