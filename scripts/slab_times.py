#!/usr/bin/env python3
"""Fused path per rank of `bench.py --gpus N`: time of the 256-level slab at every rank's d_first
(weak scaling is bounded by the slowest rank)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereomatch_amd import capi, synth  # noqa: E402

W, H, n, world = 1920, 1080, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = capi.Context(0)
left, right, _ = synth.scene_pair(W, H, 256)
d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
keys = ctx.alloc(W * H * 8)
p = capi.match_params(W, H, 1, n * world, match_fn=1, match_interval=1, match_max=1000)
for rank in range(world):
    fn = lambda: ctx.fused_box_wta_keys(d_ref, d_mtc, W, p, 9, rank * n, n, keys)
    for _ in range(3):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    ctx.sync()
    print("rank %d  d_first %4d  %.3f ms" % (rank, rank * n, (time.perf_counter() - t0) / 20 * 1e3))
