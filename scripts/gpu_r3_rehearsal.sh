cd /root/repo
for c in allreduce reduce_scatter; do
  MASTER_ADDR=127.0.0.1 timeout -k 10 300 python bench.py --gpus 1 --one-rank-rehearsal --steps 10 --warmup 3 --collective $c > gpurun_out/r3_rehearsal_$c.json 2> gpurun_out/r3_rehearsal_$c.err || { tail -5 gpurun_out/r3_rehearsal_$c.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_rehearsal_$c.json')); print('$c', d['value'], d['ms_per_step'], d['per_rank_compute_ms'], d['exposed_collective_ms'], d['collective_bytes_sent_per_rank'])"
done
