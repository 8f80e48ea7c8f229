#!/usr/bin/env python3
"""BASELINE configs 1-2 at their own size (tsukuba 384x288, b = 4, D = 16): device time per frame of
the exp1 (SSD, 9x9 box, WTA) and exp2 (+ 9x9 min filter) configurations, images resident.
GPU box only:  python scripts/bench_tsukuba.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
from stereomatch_amd import capi, pnm  # noqa: E402


def main():
    g = os.path.join(ROOT, "tests", "golden", "tsukuba")
    ref = pnm.read_image(os.path.join(g, "scene1.row3.col3.ppm"))
    mtc = pnm.read_image(os.path.join(g, "scene1.row3.col4.ppm"))
    H, W, b = ref.shape
    D, reps = 16, 200
    ctx = capi.Context(0)
    p = capi.match_params(W, H, b, D, match_fn=2, match_max=1000)
    d_ref, d_mtc = ctx.upload(ref), ctx.upload(mtc)
    N, re = W * H * D, W * D
    cost, cost2 = ctx.alloc(N * 4), ctx.alloc(N * 4)
    disp, best = ctx.alloc(W * H * 4), ctx.alloc(W * H * 4)

    def exp1():
        ctx.fused_box_wta(d_ref, d_mtc, W * b, p, 9, disp, best)

    def exp2():
        ctx.raw_costs(d_ref, d_mtc, W * b, p, cost, re)
        ctx.box_filter(cost, W, H, D, re, 9)
        ctx.min_filter_to(cost, cost2, W, H, D, re, 9)
        ctx.pad_costs(cost2, re, p)
        ctx.wta(cost2, W, H, D, re, disp, best)

    out = {}
    for name, fn in (("exp1 SSD09 (fused)", exp1), ("exp2 SSDmf09 (staged)", exp2)):
        for _ in range(10):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[name] = {"ms": round(ms, 4), "G px*disp/s": round(N / ms / 1e6, 2)}
        print("%-26s %8.4f ms/frame  %7.2f G px*disp/s" % (name, ms, N / ms / 1e6), flush=True)
    print(json.dumps({"shape": [W, H, D], "bands": b, "configs": out}))


if __name__ == "__main__":
    main()
