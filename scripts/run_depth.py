#!/usr/bin/env python3
"""Iterated aggregation on small / narrow volumes: time of smhip_*_run for the SMHIP_MAXK in the
environment.   python scripts/run_depth.py W H D"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereomatch_amd import capi, synth  # noqa: E402

W, H, D = [int(x) for x in sys.argv[1:4]]
ctx = capi.Context(0)
left, right, _ = synth.scene_pair(W, H, min(D, 64))
p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
a, b = ctx.alloc(W * H * D * 4), ctx.alloc(W * H * D * 4)
ctx.raw_costs(d_ref, d_mtc, W, p, a, W * D)
for name, fn in (("binomial x4", lambda: ctx.convolve_14641_run(a, b, W, H, D, W * D, 4)),
                 ("binomial x6", lambda: ctx.convolve_14641_run(a, b, W, H, D, W * D, 6)),
                 ("diffusion x10", lambda: ctx.diffusion_run(a, b, None, W, H, D, W * D, 0.15, 0.0, 10)),
                 ("membrane x10", lambda: ctx.diffusion_run(a, b, a, W, H, D, W * D, 0.15, 0.5, 10))):
    for _ in range(3):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    ctx.sync()
    print("%dx%dx%d maxk=%s %-14s %.3f ms" % (W, H, D, os.environ.get("SMHIP_MAXK", "policy"), name,
                                             (time.perf_counter() - t0) / 20 * 1e3))
