#!/usr/bin/env python3
"""One GPU's share of a slab-sharded staged pipeline, timed stage by stage (HIP-synchronised wall
clock): raw costs -> n binomial iterations -> PadCosts + WTA -> keys.
    python scripts/slab_pipeline.py [W H levels d_first D iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereomatch_amd import capi, synth  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:]]
    W, H, n, d0, D, it = (a + [3840, 2160, 64, 256, 512, 4][len(a):])[:6]
    ctx = capi.Context(0)
    left, right, _ = synth.scene_pair(W, H, 256)
    d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
    whole = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    ps = capi.match_params(W, H, 1, n, disp_min=d0, match_fn=1, match_interval=1)
    A, B, keys = ctx.alloc(W * H * n * 4), ctx.alloc(W * H * n * 4), ctx.alloc(W * H * 8)
    vol = W * H * n * 4 / 1e9

    def timed(name, fn, nbytes, reps=5):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print("%-28s %8.3f ms  %7.1f GB/s (algorithmic %.2f GB)" % (name, dt * 1e3, nbytes / dt, nbytes))

    timed("raw costs", lambda: ctx.raw_costs(d_ref, d_mtc, W, ps, A, W * n), vol)
    timed("binomial x%d" % it, lambda: ctx.convolve_14641_run(A, B, W, H, n, W * n, it), 2 * vol)
    timed("pad + wta -> keys", lambda: ctx.pad_wta_keys(A, W * n, whole, d0, n, keys), vol)
    timed("pad only", lambda: ctx.pad_costs(A, W * n, ps), 0.0)
    disp, best = ctx.alloc(W * H * 4), ctx.alloc(W * H * 4)
    timed("wta only", lambda: ctx.wta(A, W, H, n, W * n, disp, best), vol)


if __name__ == "__main__":
    main()
