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


def padded_key_count(n_keys, world_size):
    """Elements a key buffer needs for the reduce-scatter mode: n_keys rounded up to a multiple of the ranks."""
    return (n_keys + world_size - 1) // world_size * world_size


class KeyReducePipeline:
    """Overlaps the key reduction of frame i with the slab computation of frame i + 1.

    collective = "allreduce": one int64 MIN all-reduce of the key image; every rank ends up with the merged
    keys and `consume(buf, frame)` runs on every rank.
    collective = "reduce_scatter": int64 MIN reduce-scatter (rank r receives the minimum of chunk r, 1 / N of
    the keys, from every peer), then the reduced chunks are gathered on rank 0 only.  Per frame a rank sends
    (N-1)/N of the key image for the reduction plus its 1/N chunk to the root, and only the root receives a
    whole merged image: half the xGMI traffic of the all-reduce (which is this reduce-scatter followed by an
    all-gather to ranks that have no use for the keys).  `consume(buf, frame)` gets the merged key image on
    rank 0 and None on the other ranks.  Key buffers are flat int64 tensors of padded_key_count() elements.

    `produce(buf, frame)` enqueues the slab computation that fills key buffer `buf`;
    `consume(buf, frame)` enqueues whatever reads the merged keys (e.g. smhip_keys_decode).
    Two key buffers alternate.  The all-reduce is issued asynchronously right after
    `produce` and after the previous frame's reduce has been waited for (a stream wait under
    RCCL, no host block) and consumed, so xGMI traffic runs under the next frame's kernels and
    one collective is in flight at a time.  Stream order on the compute stream -- produce(i),
    wait(i-1), consume(i-1), [reduce(i) issued], produce(i+1), ... -- also guarantees a buffer
    is not rewritten before its consumer ran.  Call flush() after the last frame.
    """

    def __init__(self, key_buffers, produce, consume, group=None, force_collective=False, host_staging=False,
                 collective="allreduce"):
        if len(key_buffers) != 2:
            raise ValueError("KeyReducePipeline needs exactly two key buffers")
        if collective not in ("allreduce", "reduce_scatter"):
            raise ValueError("collective must be 'allreduce' or 'reduce_scatter'")
        self.bufs, self.produce, self.consume, self.group = key_buffers, produce, consume, group
        self.collective = collective
        self._chunk = None            # reduce_scatter: this rank's reduced chunk (two, alternating with the buffers)
        self._full = None             # reduce_scatter, rank 0: the gathered key image
        self.force_collective = force_collective    # issue the all-reduce even in a group of one (tests)
        # host_staging: the collective runs on a pinned CPU copy of the keys (copy out, all-reduce, copy
        # back).  For the gloo backend with device key buffers: gloo's own device-tensor path stalls when
        # two ranks share one GPU (scripts/gloo_probe.py), its CPU path does not.  No overlap in this mode.
        self.host_staging = host_staging
        self._host = None
        self.n = 0
        self.pending = None

    class _StagedReduce:
        """wait() = copy the keys to the host, all-reduce them there, copy them back (stream-ordered)."""

        def __init__(self, pipe, buf):
            self.pipe, self.buf = pipe, buf

        def wait(self):
            import torch
            import torch.distributed as dist
            p = self.pipe
            if p._host is None or p._host.shape != self.buf.shape:
                p._host = torch.empty(self.buf.shape, dtype=self.buf.dtype, device="cpu").pin_memory()
            p._host.copy_(self.buf, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            dist.all_reduce(p._host, op=dist.ReduceOp.MIN, group=p.group)
            self.buf.copy_(p._host, non_blocking=True)

    class _ScatterReduce:
        """reduce-scatter of `buf` into this rank's chunk, then the chunks gathered on rank 0.  RCCL: the
        reduce-scatter is issued asynchronously at construction (it runs under the next frame's kernels) and
        wait() enqueues the gather behind it.  Other backends (gloo has no reduce-scatter): one MIN reduce per
        chunk to the rank that owns it, on a pinned host copy when the keys live on a device
        (`host_staging`), then the gather."""

        def __init__(self, pipe, buf, slot):
            import torch
            import torch.distributed as dist
            self.pipe, self.buf = pipe, buf
            self.world, self.rank = dist.get_world_size(pipe.group), dist.get_rank(pipe.group)
            if buf.numel() % self.world:
                raise ValueError("key buffer of %d elements is not a multiple of %d ranks (padded_key_count)" % (
                    buf.numel(), self.world))
            n = buf.numel() // self.world
            if pipe._chunk is None or pipe._chunk[0].numel() != n or pipe._chunk[0].device != buf.device:
                pipe._chunk = [torch.empty(n, dtype=buf.dtype, device=buf.device) for _ in range(2)]
            if self.rank == 0 and (pipe._full is None or pipe._full.numel() != buf.numel()
                                   or pipe._full.device != buf.device):
                pipe._full = torch.empty(buf.numel(), dtype=buf.dtype, device=buf.device)
            self.chunk = pipe._chunk[slot]
            self.native = dist.get_backend(pipe.group) == "nccl" and buf.is_cuda
            self.work = (dist.reduce_scatter_tensor(self.chunk, buf.view(-1), op=dist.ReduceOp.MIN, group=pipe.group,
                                                    async_op=True) if self.native else None)

        def wait(self):
            import torch
            import torch.distributed as dist
            p = self.pipe
            flat = self.buf.view(-1)
            n = self.chunk.numel()
            if self.native:
                self.work.wait()
                dist.gather(self.chunk, list(p._full.view(self.world, n)) if self.rank == 0 else None, dst=0,
                            group=p.group)
                return p._full if self.rank == 0 else None
            src = flat
            if flat.is_cuda:                      # host staging (two ranks on one device in the rehearsals)
                if p._host is None or p._host.shape != flat.shape:
                    p._host = torch.empty(flat.shape, dtype=flat.dtype, device="cpu").pin_memory()
                p._host.copy_(flat, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                src = p._host
            for r in range(self.world):           # chunk r reduced onto rank r
                dist.reduce(src[r * n:(r + 1) * n], dst=r, op=dist.ReduceOp.MIN, group=p.group)
            mine = src[self.rank * n:(self.rank + 1) * n]
            full_host = torch.empty_like(src) if self.rank == 0 else None
            dist.gather(mine.contiguous(), list(full_host.view(self.world, n)) if self.rank == 0 else None, dst=0,
                        group=p.group)
            if self.rank != 0:
                return None
            if flat.is_cuda:
                p._full.copy_(full_host, non_blocking=True)
                return p._full
            return full_host

    def _reduce_async(self, buf):
        import torch.distributed as dist
        if dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force_collective):
            if self.collective == "reduce_scatter":
                return KeyReducePipeline._ScatterReduce(self, buf, (self.n - 1) & 1)
            if self.host_staging and buf.is_cuda:
                return KeyReducePipeline._StagedReduce(self, buf)
            return dist.all_reduce(buf, op=dist.ReduceOp.MIN, group=self.group, async_op=True)
        return None

    def _drain(self):
        if self.pending is not None:
            work, buf, frame = self.pending
            self.pending = None
            if work is not None:
                res = work.wait()
                if isinstance(work, KeyReducePipeline._ScatterReduce):
                    buf = res if res is None else res.view(buf.shape)     # merged keys on rank 0, None elsewhere
            self.consume(buf, frame)

    def submit(self, frame):
        buf = self.bufs[self.n & 1]
        self.n += 1
        self.produce(buf, frame)
        # the previous frame's reduce ran under the kernels just enqueued; it is waited for and
        # consumed before this frame's reduce is issued, so one collective is in flight at a time
        # (two overlapping 16 MB gloo all-reduces were seen to stall each other in rehearsals; RCCL
        # serialises collectives of one communicator on its stream anyway)
        self._drain()
        self.pending = (self._reduce_async(buf), buf, frame)

    def flush(self):
        self._drain()
