"""The collective leg on real RCCL, as far as one GPU allows: a process group of one rank with the
nccl backend.  It proves what the CPU (gloo) tests cannot: that RCCL takes an int64 MIN all-reduce
of the key image on the library's stream, and that KeyReducePipeline's asynchronous issue / stream
wait order works with RCCL's work handles.  (torch and libsmhip.so share one HIP runtime whatever the import order:
capi._share_torch_hip_runtime.)"""
import os

import numpy as np
import pytest

import util
from stereomatch_amd import capi, sharding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_group():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    store = dist.TCPStore("127.0.0.1", util.free_port(), 1, True)
    dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_pipeline_over_rccl_single_rank(gpu_ctx, oracle, rccl_group):
    import torch
    dist = rccl_group
    W, H, D, win = 200, 64, 64, 9
    left, right, _ = synth.scene_pair(W, H, D)
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    gpu_ctx.set_stream(stream.cuda_stream)
    try:
        t_ref = torch.from_numpy(left.reshape(H, W)).to(dev)
        t_mtc = torch.from_numpy(right.reshape(H, W)).to(dev)
        bufs = [torch.empty((W, H), dtype=torch.int64, device=dev) for _ in range(2)]
        disp = torch.empty((H, W), dtype=torch.int32, device=dev)
        best = torch.empty((H, W), dtype=torch.float32, device=dev)
        got = []

        def produce(buf, frame):
            gpu_ctx.fused_box_wta_keys(t_ref.data_ptr(), t_mtc.data_ptr(), W, p, win + 2 * (frame % 2), 0, D,
                                       buf.data_ptr())

        def consume(buf, frame):
            gpu_ctx.keys_decode(buf.data_ptr(), W, H, disp.data_ptr(), best.data_ptr())
            got.append((frame, disp.cpu().numpy().copy(), best.cpu().numpy().copy()))

        pipe = sharding.KeyReducePipeline(bufs, produce, consume, force_collective=True)
        for frame in range(5):
            pipe.submit(frame)
        pipe.flush()
        torch.cuda.synchronize()
        # a MIN against a second key image through RCCL itself (one rank: identity), values kept
        k = bufs[0].clone()
        dist.all_reduce(k, op=dist.ReduceOp.MIN)
        assert torch.equal(k, bufs[0])
    finally:
        gpu_ctx.set_stream(None)
    assert [g[0] for g in got] == list(range(5))
    for frame, d, b in got:
        q = oracle.make_params(W, H, 1, disp_max=D - 1, match_fn=1, match_interval=1,
                               aggr_window_size=win + 2 * (frame % 2))
        od, ob, _, _ = oracle.compute_correspondence(left, right, q)
        assert np.array_equal(d, od), frame
        util.assert_bit_equal(b, ob, "best cost, frame %d" % frame)
