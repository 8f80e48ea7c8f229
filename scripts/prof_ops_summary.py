"""Condenses scripts/prof_ops_pmc.sh output: per kernel the traced average time, the HBM bytes the PMC
counters saw (FETCH_SIZE * 2 + WRITE_SIZE, KiB: profiles/README.md), the rate they give and the share of
SIMD cycles with a vector instruction in flight (SQ_ACTIVE_INST_VALU counts quad-cycles; kernel cycles =
GRBM_GUI_ACTIVE / 8 XCDs; 1024 SIMDs).  usage: prof_ops_summary.py <prof dir> <out.json>"""
import json
import sys

d = json.load(open(sys.argv[1] + "/summary.json"))
out = {}
for k, t in d["kernel_trace"].items():
    c = d.get("pmc", {}).get(k)
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    nbytes = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out[k.split("(")[0].replace("void ", "")] = {
        "calls": t["calls"], "avg_us": round(t["avg_us"], 1),
        "measured_bytes": int(nbytes), "achieved_GBs": round(nbytes / t["avg_us"] / 1e3, 1),
        "frac_of_8TBs_on_measured_bytes": round(nbytes / t["avg_us"] / 1e3 / 8000, 3),
        "valu_active_pct": round(100.0 * c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), 1),
        "valu_insts": int(c["SQ_INSTS_VALU"]),
        "lds_active_pct": round(100.0 * c.get("SQ_ACTIVE_INST_LDS", 0) * 4 / (256 * cyc), 1)}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"]):
    print("%-48s %9.1f us %7.2f GB %7.1f GB/s  VALU %5.1f%%  LDS %5.1f%%" % (
        k[:48], v["avg_us"], v["measured_bytes"] / 1e9, v["achieved_GBs"], v["valu_active_pct"], v["lds_active_pct"]))
