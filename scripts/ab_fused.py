#!/usr/bin/env python3
"""A/B timing of the fused pair at config 3 (or --width/--height/--levels/--first) under context options:
pass A / pass B event times as bench.py records them.  GPU box only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stereomatch_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--levels", type=int, default=256)
ap.add_argument("--total", type=int, default=0)
ap.add_argument("--first", type=int, default=0)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--bands", type=int, default=1)
ap.add_argument("--window", type=int, default=9)
ap.add_argument("--interval", type=int, default=1)
ap.add_argument("--opts", nargs="+", default=["{}", '{"float_a": 0}'], help="option sets as JSON objects")
args = ap.parse_args()
W, H, n = args.width, args.height, args.levels
D = args.total or n
dev = torch.device("cuda", 0)
nb = args.bands
left, right, _ = synth.scene_pair(W, H, 256, n_bands=nb)
ctx = capi.Context(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
t_ref, t_mtc = torch.from_numpy(left.reshape(H, W * nb)).to(dev), torch.from_numpy(right.reshape(H, W * nb)).to(dev)
keys = torch.empty((W, H), dtype=torch.int64, device=dev)
disp = torch.empty((H, W), dtype=torch.int32, device=dev)
best = torch.empty((H, W), dtype=torch.float32, device=dev)
p = capi.match_params(W, H, nb, D, match_fn=1, match_interval=args.interval, match_max=1000)


def step():
    ctx.fused_box_wta_keys(t_ref.data_ptr(), t_mtc.data_ptr(), W * nb, p, args.window, args.first, n, keys.data_ptr())
    ctx.keys_decode(keys.data_ptr(), W, H, disp.data_ptr(), best.data_ptr())


ref = None
for rep in range(2):
    for opts in [json.loads(o) for o in args.opts]:
        with ctx.options(**opts):
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            ctx.profile_begin(args.steps)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            k, a, b = ctx.profile_end()
            out = (disp.cpu(), best.cpu())
            if ref is None:
                ref = out
            same = bool((out[0] == ref[0]).all() and (out[1].view(torch.int32) == ref[1].view(torch.int32)).all())
            print("%-32s A %.4f ms  B %.4f ms  step %.4f ms  %.1f G px*disp/s  same=%s" % (
                opts, a / k, b / k, e0.elapsed_time(e1) / args.steps, W * H * n / (e0.elapsed_time(e1) / args.steps) / 1e6, same),
                flush=True)
