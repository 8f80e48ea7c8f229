#!/usr/bin/env python3
"""Minimal check of what the gloo backend does with CUDA int64 tensors (rehearsal aid for the
N > 1 path of bench.py on a one-GPU box).  torchrun --nproc-per-node 2 scripts/rehearse_gloo.py"""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(60, exit=True)
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
t = torch.full((1000,), 10 + rank, dtype=torch.int64, device="cuda")
print(rank, "sync all_reduce MIN on a CUDA int64 tensor ...", flush=True)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
print(rank, "ok", int(t[0]), flush=True)
t = torch.full((1000,), 20 + rank, dtype=torch.int64, device="cuda")
print(rank, "async ...", flush=True)
w = dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=True)
u = torch.ones(10, device="cuda") * 2          # more work on the current stream meanwhile
w.wait()
print(rank, "ok", int(t[0]), float(u[0]), flush=True)
dist.barrier()
dist.destroy_process_group()
