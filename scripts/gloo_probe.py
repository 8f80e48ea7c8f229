#!/usr/bin/env python3
"""Does a two-rank gloo int64 MIN all-reduce of a 1920x1080 key image stall on this box without any of
this repo's code?  argv: cpu|cuda iterations.   torchrun --nproc-per-node 2 scripts/gloo_probe.py cuda 40"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(50, exit=True)
mode, iters = sys.argv[1], int(sys.argv[2])
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
dev = torch.device("cuda", 0) if mode == "cuda" else torch.device("cpu")
bufs = [torch.full((1920, 1080), rank + 1, dtype=torch.int64, device=dev) for _ in range(2)]
scratch = torch.empty((64, 1024, 1024), dtype=torch.float32, device=dev) if mode == "cuda" else None
pending = None
t0 = time.time()
for i in range(iters):
    b = bufs[i & 1]
    if scratch is not None:
        scratch.fill_(float(i))          # some GPU work between the collectives
    b.fill_(rank + 1 + i)
    if pending is not None:
        pending.wait()
    pending = dist.all_reduce(b, op=dist.ReduceOp.MIN, async_op=True)
    if rank == 0 and i % 10 == 0:
        print("iteration", i, "%.1f s" % (time.time() - t0), flush=True)
pending.wait()
if mode == "cuda":
    torch.cuda.synchronize()
assert int(bufs[(iters - 1) & 1][0, 0]) == iters
dist.barrier()
if rank == 0:
    print("gloo probe %s x%d ok in %.1f s" % (mode, iters, time.time() - t0), flush=True)
dist.destroy_process_group()
