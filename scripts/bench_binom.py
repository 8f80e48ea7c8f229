#!/usr/bin/env python3
"""Times smhip_convolve_14641_run (multi-iteration binomial passes) on a resident DSI under context options.
GPU box only:  python scripts/bench_binom.py [W H D]   (default 3840 2160 512)"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from stereomatch_amd import capi, synth  # noqa: E402


def main():
    W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (3840, 2160, 512)
    ctx = capi.Context(0)
    left, right, _ = synth.scene_pair(W, H, min(D, 256))
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
    N, re = W * H * D, W * D
    a, b = ctx.alloc(N * 4), ctx.alloc(N * 4)

    def timed(fn, reps=4):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    variants = [dict(maxk=4), dict(maxk=4, tap_slow=1), dict(maxk=3), dict(maxk=2), dict(maxk=4, multi_px=32),
                dict(maxk=4, tseg=64), dict(maxk=4, tseg=256), dict(maxk=4, tseg=512), dict(maxk=4, stage_lds_kb=41),
                dict(maxk=4, multi_px=0), dict(maxk=4, multi_px=0, multi_txt=128, tseg=256)]
    if len(sys.argv) >= 5:               # a comma-separated list of variant indices
        variants = [variants[int(i)] for i in sys.argv[4].split(",")]
    for opts in variants:
        for k, v in opts.items():
            ctx.set_option(k, v)
        line = []
        for n in ((4,) if len(sys.argv) >= 5 else (4, 8)):
            ctx.raw_costs(d_ref, d_mtc, W, p, a, re)          # fresh integer costs: the guard stays on the fast path
            ms = timed(lambda: ctx.convolve_14641_run(a, b, W, H, D, re, n))
            line.append("x%d %7.3f ms" % (n, ms))
        print("%-40s %s" % (opts, "   ".join(line)), flush=True)
        for k in opts:
            ctx.set_option(k, -1)


if __name__ == "__main__":
    main()
