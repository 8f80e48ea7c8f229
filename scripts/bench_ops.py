#!/usr/bin/env python3
"""Times the stage operators of the C-ABI on a resident DSI (BASELINE configs 2 and 4 shapes) and
prints achieved GB/s against the algorithmic bytes of SURVEY 8d (8 bytes per voxel and filter
application; WTA 4 bytes per voxel).  The iterated "(run)" entries hold up to 4 iterations in LDS per pass
over the DSI, so their bytes are 8 (membrane: 12) per voxel and PASS, ceil(n / 4) passes; their iteration
rate is the "Gvox/s" column x n.  The HBM bytes the counters see for each kernel are in
profiles/r02_stage_ops_summary.json (scripts/prof_ops_pmc.sh).  GPU box only:
python scripts/bench_ops.py [W H D]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from stereomatch_amd import capi, synth  # noqa: E402


def main():
    W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (3840, 2160, 512)
    reps = 5
    ctx = capi.Context(0)
    quick = False
    if len(sys.argv) >= 5:               # a JSON object of context options; "quick": skip the slow optimisers
        opts = json.loads(sys.argv[4])
        quick = bool(opts.pop("quick", 0))
        for k, v in opts.items():
            ctx.set_option(k, v)
    left, right, _ = synth.scene_pair(W, H, min(D, 256))
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
    N = W * H * D
    re = W * D
    cost, cost2 = ctx.alloc(N * 4), ctx.alloc(N * 4)
    disp, best = ctx.alloc(W * H * 4), ctx.alloc(W * H * 4)
    disp_src = ctx.alloc(N * 4)          # a third volume: cost0 of the membrane model
    out = {}

    def timed(name, nbytes, fn):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[name] = {"ms": round(ms, 3), "GB/s": round(nbytes / ms / 1e6, 1),
                     "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3),
                     "Gvox/s": round(N / ms / 1e6, 1)}
        print("%-22s %9.3f ms  %8.1f GB/s  frac %.3f" % (name, ms, nbytes / ms / 1e6,
                                                          nbytes / ms / 1e6 / 8000), flush=True)

    timed("raw_costs (BT-SAD)", 4.0 * N, lambda: ctx.raw_costs(d_ref, d_mtc, W, p, cost, re))
    timed("box_filter 9x9", 8.0 * N, lambda: ctx.box_filter(cost, W, H, D, re, 9))
    timed("min_filter 9x9", 8.0 * N, lambda: ctx.min_filter(cost, W, H, D, re, 9))
    timed("min_filter_to 9x9", 8.0 * N, lambda: ctx.min_filter_to(cost, cost2, W, H, D, re, 9))
    timed("convolve_14641", 8.0 * N, lambda: ctx.convolve_14641(cost, W, H, D, re))
    timed("convolve_14641_to", 8.0 * N, lambda: ctx.convolve_14641_to(cost, cost2, W, H, D, re))
    timed("diffusion_iter", 8.0 * N, lambda: ctx.diffusion_iter(cost, cost2, None, W, H, D, re, 0.15, 0.0))
    timed("membrane_iter", 12.0 * N, lambda: ctx.diffusion_iter(cost, cost2, cost, W, H, D, re, 0.15, 0.5))
    for n in (2, 4, 8):
        timed("binomial x%d (run)" % n, 8.0 * N * ((n + 3) // 4),
              lambda: ctx.convolve_14641_run(cost, cost2, W, H, D, re, n))
    for n in (4, 10):
        timed("diffusion x%d (run)" % n, 8.0 * N * ((n + 3) // 4),
              lambda: ctx.diffusion_run(cost, cost2, None, W, H, D, re, 0.15, 0.0, n))
    timed("membrane x4 (run)", 12.0 * N,
          lambda: ctx.diffusion_run(cost, cost2, disp_src, W, H, D, re, 0.15, 0.5, 4))
    keys = ctx.alloc(W * H * 8)
    half = D // 2 if D >= 64 else D
    for fn, name, its in ((2, "binomial", (4, 8)), (3, "diffusion", (6, 10))):
        for it in its:                   # raw costs -> iterations -> PadCosts -> WTA keys, fused (no DSI at <= 4 / 6)
            timed("fused %s x%d -> keys" % (name, it), 2.0 * W * H + 8.0 * W * H,
                  lambda: [ctx.fused_iter_wta_keys(d_ref, d_mtc, W, p, fn, it, s0, half, keys) for s0 in range(0, D, half)])
    timed("pad_costs", 4.0 * D * D / 2 * H, lambda: ctx.pad_costs(cost, re, p))
    timed("wta", 4.0 * N + 8.0 * W * H, lambda: ctx.wta(cost, W, H, D, re, disp, best))
    if quick:
        print(json.dumps({"shape": [W, H, D], "ops": out}))
        return
    timed("opt_so (scan-line opt.)", 4.0 * N + 2.0 * N,
          lambda: ctx.opt_so(cost, W, H, D, re, d_ref, W, 1, 20.0, 8.0, 2.0, disp, best))
    timed("opt_dp (dynamic programming)", 4.0 * N + 2.0 * N,
          lambda: ctx.opt_dp(cost, W, H, D, re, d_ref, W, 1, 20.0, 8.0, 2.0, 20, disp, best))
    print(json.dumps({"shape": [W, H, D], "ops": out}))


if __name__ == "__main__":
    main()
