"""Summarises rocprofv3 csv output of scripts/profile.sh: per-kernel average duration and counters."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
kt = glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True)
if kt:
    dur = defaultdict(list)
    with open(kt[0]) as f:
        for r in csv.DictReader(f):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in dur.values())
    res["kernel_trace"] = {k: {"calls": len(v), "avg_us": sum(v) / len(v) / 1e3,
                               "min_us": min(v) / 1e3, "pct": 100.0 * sum(v) / tot}
                           for k, v in dur.items()}
for p in sorted(d for d in os.listdir(out) if d.startswith("pmc") and os.path.isdir(os.path.join(out, d))):
    cc = glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    with open(cc[0]) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        res.setdefault("pmc", {}).setdefault(k, {}).update(
            {c: sum(v) / len(v) for c, v in d.items()})
with open(os.path.join(out, "summary.json"), "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
for k, v in sorted(res.get("kernel_trace", {}).items(), key=lambda kv: -kv[1]["pct"]):
    print("%-70s calls %3d avg %9.1f us min %9.1f us  %5.1f%%" % (k[:70], v["calls"], v["avg_us"], v["min_us"], v["pct"]))
for k, d in res.get("pmc", {}).items():
    if "fused" in k or len(res["pmc"]) <= 12:
        print(k[:60], {c: round(x, 1) for c, x in sorted(d.items())})
