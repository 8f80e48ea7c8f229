#!/usr/bin/env python3
"""exp2 pipeline (box 9x9 + 9x9 min filter + PadCosts + WTA, Scripts/exp2_costSSDMF9.txt) at
1920x1080xD: staged route against the fused route (smhip_fused_box_dsi + smhip_min_pad_wta_keys).
GPU box only:  python scripts/bench_exp2.py [W H D]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from stereomatch_amd import capi, synth  # noqa: E402


def main():
    W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 256)
    reps = 20
    ctx = capi.Context(0)
    if len(sys.argv) >= 5:               # a JSON object of context options
        for k, v in json.loads(sys.argv[4]).items():
            ctx.set_option(k, v)
    left, right, _ = synth.scene_pair(W, H, min(D, 256))
    p = capi.match_params(W, H, 1, D, match_fn=2, match_max=20)
    d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
    N, re = W * H * D, W * D
    cost, cost2 = ctx.alloc(N * 4), ctx.alloc(N * 4)
    disp, best, keys = ctx.alloc(W * H * 4), ctx.alloc(W * H * 4), ctx.alloc(W * H * 8)
    disp2, best2 = ctx.alloc(W * H * 4), ctx.alloc(W * H * 4)

    def staged():
        ctx.raw_costs(d_ref, d_mtc, W, p, cost, re)
        ctx.box_filter(cost, W, H, D, re, 9)
        ctx.min_filter_to(cost, cost2, W, H, D, re, 9)
        ctx.pad_costs(cost2, re, p)
        ctx.wta(cost2, W, H, D, re, disp, best)

    def fused():
        ctx.fused_box_dsi(d_ref, d_mtc, W, p, 9, 0, D, cost, re)
        ctx.min_pad_wta_keys(cost, re, p, 9, 0, D, keys)
        ctx.keys_decode(keys, W, H, disp2, best2)

    def box_dsi_only():
        ctx.fused_box_dsi(d_ref, d_mtc, W, p, 9, 0, D, cost, re)

    def min_keys_only():
        ctx.min_pad_wta_keys(cost, re, p, 9, 0, D, keys)

    def min_keys_seg64():
        with ctx.options(tseg=64):
            ctx.min_pad_wta_keys(cost, re, p, 9, 0, D, keys)

    def min_keys_seg16():
        with ctx.options(tseg=16):
            ctx.min_pad_wta_keys(cost, re, p, 9, 0, D, keys)

    out = {}
    for name, fn, nbytes in (("staged (raw, box, min, pad, wta)", staged, 16.0 * N),
                             ("fused (box->DSI, min+pad+WTA keys)", fused, 16.0 * N),
                             ("  smhip_fused_box_dsi", box_dsi_only, 12.0 * N),
                             ("  smhip_min_pad_wta_keys", min_keys_only, 4.0 * N),
                             ("  (same, 64-row segments)", min_keys_seg64, 4.0 * N),
                             ("  (same, 16-row segments)", min_keys_seg16, 4.0 * N)):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[name.strip()] = {"ms": round(ms, 3), "G px*disp/s": round(N / ms / 1e6, 1),
                             "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3)}
        print("%-38s %8.3f ms  %8.1f G px*disp/s  frac %.3f of %2.0f N" % (
            name, ms, N / ms / 1e6, nbytes / ms / 1e6 / 8000, nbytes / N), flush=True)
    a = ctx.download(disp, (H, W), np.int32)
    b = ctx.download(disp2, (H, W), np.int32)
    print("identical disparity maps:", bool(np.array_equal(a, b)))
    print(json.dumps({"shape": [W, H, D], "routes": out}))


if __name__ == "__main__":
    main()
