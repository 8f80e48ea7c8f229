#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r3_b4pmc -- python3 bench.py --bands 4 --steps 10 --warmup 2 --no-cpu-baseline --no-extra --sustained-seconds 0 > gpurun_out/r3_b4pmc.log 2>&1 || { echo pmc failed; tail -5 gpurun_out/r3_b4pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3_b4pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "fused" in k:
        c = {n: sum(x) / len(x) for n, x in v.items()}
        cyc = c["GRBM_GUI_ACTIVE"] / 8 * 1024
        print(k, {n: round(x) for n, x in c.items()}, "VALU %.1f%% LDS %.1f%%" % (c["SQ_ACTIVE_INST_VALU"] * 4 / cyc * 100, c["SQ_ACTIVE_INST_LDS"] * 4 / cyc * 100))
PY
