#!/usr/bin/env python3
"""Times smhip_fused_iter_wta_keys (raw costs -> n binomial iterations -> PadCosts -> WTA keys) against the stage
calls on a private DSI (option iter_fused = 0).  GPU box only:  python scripts/bench_iter.py [W H D d_first d_count]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from stereomatch_amd import capi, synth  # noqa: E402


def main():
    W, H, D, d0, n = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (3840, 2160, 512, 256, 64)
    opts = json.loads(sys.argv[6]) if len(sys.argv) >= 7 else {}
    ctx = capi.Context(0)
    left, right, _ = synth.scene_pair(W, H, min(D, 256))
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
    keys = ctx.alloc(W * H * 8)

    def timed(fn, reps=5):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    for k, v in opts.items():
        ctx.set_option(k, v)
    for fn, name, its in ((2, "binomial", (2, 4, 8)), (3, "diffusion", (4, 6, 10, 50))):
        for it in its:
            row = []
            for fused in (1, 0):
                ctx.set_option("iter_fused", fused)
                ms = timed(lambda: ctx.fused_iter_wta_keys(d_ref, d_mtc, W, p, fn, it, d0, n, keys), 5 if it < 50 else 2)
                row.append(ms)
            ctx.set_option("iter_fused", -1)
            print("%dx%d levels [%d,%d) %s x%d: fused %8.3f ms (%6.1f G px*disp/s)   stage calls %8.3f ms" % (
                W, H, d0, d0 + n, name, it, row[0], W * H * n / row[0] / 1e6, row[1]), flush=True)


if __name__ == "__main__":
    main()
