#!/bin/bash
# Full GPU suite + the driver's bench command, as the round end runs them.  bash scripts/gpu_full.sh [tag]
tag=${1:-full}
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"], d["roofline_kernel"], d.get("verified_vs_oracle"))
for k, v in d.get("extra", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("ms", "value", "ms_per_step", "pass_a_ms", "pass_b_ms", "G_pixel_disparity_per_s")} if isinstance(v, dict) else v)
PY
