"""Runs the two scan-line optimisers a few times on a synthetic DSI (for rocprofv3):
python scripts/run_opt.py W H D"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stereomatch_amd import capi, synth

W, H, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 256)
ctx = capi.Context(0)
left, right, _ = synth.scene_pair(W, H, min(D, 256))
p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
cost = ctx.alloc(W * H * D * 4)
disp, best = ctx.alloc(W * H * 4), ctx.alloc(W * H * 4)
ctx.raw_costs(d_ref, d_mtc, W, p, cost, W * D)
for _ in range(3):
    ctx.opt_so(cost, W, H, D, W * D, d_ref, W, 1, 20.0, 8.0, 2.0, disp, best)
    ctx.opt_dp(cost, W, H, D, W * D, d_ref, W, 1, 20.0, 8.0, 2.0, 20, disp, best)
ctx.sync()
print("done")
