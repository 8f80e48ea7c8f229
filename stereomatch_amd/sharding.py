"""Disparity-slab sharding of the fused path across GPUs.

Every stage of the hot path is independent per disparity level (StcRawCosts.cpp:405-416,
BoxFilter.cpp:119-125), so rank r runs `smhip_fused_box_wta_keys` on levels
[d_first, d_first + d_count) and the only exchange is one int64 MIN all-reduce of the
per-pixel keys (RCCL on GPUs, gloo in the CPU tests).  key = (ordered_bits(cost) << 32) | d:
the minimum key is the lowest cost and, among equal costs, the lowest level -- OptWTA's
strict '<' scan (StcOptimize.cpp:43-52).

The numpy encode / decode below mirror `ordered_bits` / `make_key` of csrc/smhip_kernels.h;
they exist for host-side checks and for callers that merge keys without a device.
"""
import numpy as np


def slab_bounds(disp_n, world_size, rank):
    """Levels [d_first, d_first + d_count) of `rank`: contiguous, sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside [0, %d)" % (rank, world_size))
    if world_size > disp_n:
        raise ValueError("more ranks (%d) than disparity levels (%d)" % (world_size, disp_n))
    d_first = disp_n * rank // world_size
    return d_first, disp_n * (rank + 1) // world_size - d_first


def ordered_bits(cost):
    """float32 order -> int32 order (-0 folded into +0), like the device function."""
    b = (np.asarray(cost, np.float32) + np.float32(0.0)).view(np.int32)
    return b ^ ((b >> 31) & np.int32(0x7FFFFFFF))


def encode_keys(cost, disp):
    s = ordered_bits(cost).astype(np.int64)
    return (s << 32) | np.asarray(disp, np.int64)


def decode_keys(keys):
    """Returns (disparity int32, best cost float32)."""
    keys = np.asarray(keys, np.int64)
    d = (keys & 0xFFFFFFFF).astype(np.int32)
    s = (keys >> 32).astype(np.int32)
    return d, (s ^ ((s >> 31) & np.int32(0x7FFFFFFF))).view(np.float32)


def all_reduce_min_keys(keys_tensor, group=None):
    """In-place int64 MIN all-reduce of a torch key tensor (any layout, element-wise)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(keys_tensor, op=dist.ReduceOp.MIN, group=group)
    return keys_tensor
