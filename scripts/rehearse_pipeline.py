#!/usr/bin/env python3
"""Rehearsal of the N > 1 frame pipeline (sharding.KeyReducePipeline) on ONE GPU with the gloo
backend: both ranks use cuda:0.  argv: profile(0/1) frames [W H levels_per_rank]
  torchrun --nproc-per-node 2 scripts/rehearse_pipeline.py 1 8"""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from stereomatch_amd import capi, sharding, synth  # noqa: E402

faulthandler.dump_traceback_later(45, exit=True)
profile, frames = int(sys.argv[1]), int(sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
W, H, Dg = (int(a) for a in sys.argv[3:6]) if len(sys.argv) >= 6 else (640, 360, 64)
D = Dg * world
left, right, _ = synth.scene_pair(W, H, D)
ctx = capi.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
t_ref, t_mtc = torch.from_numpy(left.reshape(H, W)).to(dev), torch.from_numpy(right.reshape(H, W)).to(dev)
bufs = [torch.empty((W, H), dtype=torch.int64, device=dev) for _ in range(2)]
disp = torch.empty((H, W), dtype=torch.int32, device=dev)
best = torch.empty((H, W), dtype=torch.float32, device=dev)
p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
out = []


def produce(buf, f):
    ctx.fused_box_wta_keys(t_ref.data_ptr(), t_mtc.data_ptr(), W, p, 9, rank * Dg, Dg, buf.data_ptr())


def consume(buf, f):
    ctx.keys_decode(buf.data_ptr(), W, H, disp.data_ptr(), best.data_ptr())
    out.append(disp.clone())


pipe = sharding.KeyReducePipeline(bufs, produce, consume)
for rnd in range(2):
    if profile and rnd == 1:
        ctx.profile_begin(frames)
    for f in range(frames):
        pipe.submit(f)
        print(rank, "round", rnd, "frame", f, "submitted", flush=True)
    pipe.flush()
    dist.barrier()
    torch.cuda.synchronize()
    if profile and rnd == 1:
        print(rank, "profile", ctx.profile_end(), flush=True)
# single-process result over all levels for comparison
ctx.fused_box_wta(t_ref.data_ptr(), t_mtc.data_ptr(), W, p, 9, disp.data_ptr(), best.data_ptr())
torch.cuda.synchronize()
ok = all(bool(torch.equal(o, disp)) for o in out)
print(rank, "frames", len(out), "all equal to the single-device result:", ok, flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
