"""smhip_fused_box_wta_noalloc -- the entry point that allocates nothing (its slab, interval images and keys
live in a static device array) -- against smhip_fused_box_wta AND against the CPU oracle, through the C-ABI, bit
for bit: two calls in a row, the second with a smaller input (the array keeps the first call's contents), then
problems that do not fit (refused before anything is queued, also on the shapes that run a Birchfield-Tomasi
pre-pass) and a call after them; the call under HIP graph capture."""
import numpy as np
import pytest

import util
from oracle import oracle as O
from stereomatch_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_both(ctx, w, h, b, D, win, **kw):
    left, right, _ = synth.scene_pair(w, h, D, n_bands=b)
    p = capi.match_params(w, h, b, D, **kw)
    d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
    outs = []
    for fn in (ctx.fused_box_wta, ctx.fused_box_wta_noalloc):
        d_disp, d_best = ctx.alloc(w * h * 4), ctx.alloc(w * h * 4)
        fn(d_ref, d_mtc, w * b, p, win, d_disp, d_best)
        outs.append((ctx.download(d_disp, (h, w), np.int32), ctx.download(d_best, (h, w), np.float32)))
    util.assert_bit_equal(outs[1][0], outs[0][0], "disparity %dx%dx%d" % (w, h, D))
    util.assert_bit_equal(outs[1][1], outs[0][1], "best cost %dx%dx%d" % (w, h, D))
    # and directly against the oracle on the same inputs
    op = O.make_params(w, h, b, disp_min=0, disp_max=D - 1, aggr_window_size=win, **kw)
    odisp, obest = O.compute_correspondence(left, right, op)[:2]
    util.assert_bit_equal(outs[1][0], odisp, "no-alloc disparity vs oracle %dx%dx%d" % (w, h, D))
    util.assert_bit_equal(outs[1][1], obest, "no-alloc best cost vs oracle %dx%dx%d" % (w, h, D))
    return outs[0]


def test_noalloc_equals_allocating_entry_point():
    ctx = capi.Context(0)          # a context of its own: its workspace must stay empty
    try:
        # tsukuba's shape: 4 bands, SSD, 16 levels; then a smaller one-band BT-SAD problem of 40 levels
        a = run_both(ctx, 384, 288, 4, 16, 9, match_fn=2, match_max=1000)
        b = run_both(ctx, 200, 100, 1, 40, 5, match_fn=1, match_interval=1, match_max=20)
        assert a[0].max() > 0 and b[0].max() > 0
        # too large for the 64 MiB array: refused before anything is launched
        w, h, D = 1920, 1080, 256
        left, right, _ = synth.scene_pair(w, h, D)
        p = capi.match_params(w, h, 1, D, match_fn=1, match_interval=1)
        d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
        d_disp, d_best = ctx.alloc(w * h * 4), ctx.alloc(w * h * 4)
        with pytest.raises(capi.SmhipError, match="scratch"):
            ctx.fused_box_wta_noalloc(d_ref, d_mtc, w, p, 9, d_disp, d_best)
        # shapes that run a Birchfield-Tomasi pre-pass kernel (four bands with intervals; the forced pre-pass):
        # the refusal must come before that launch too -- the scratch is asked for in full first.  A sentinel
        # written into the output buffers must survive, and so must the static array's use by the next call.
        for bands, opts in ((4, {}), (1, {"bt_prepass": 1})):
            w, h, D = 1600, 900, 128
            left, right, _ = synth.scene_pair(w, h, D, n_bands=bands)
            p = capi.match_params(w, h, bands, D, match_fn=1, match_interval=1)
            d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
            sentinel = np.full((h, w), -7, np.int32)
            d_disp, d_best = ctx.upload(sentinel), ctx.alloc(w * h * 4)
            with ctx.options(**opts):
                with pytest.raises(capi.SmhipError, match="scratch"):
                    ctx.fused_box_wta_noalloc(d_ref, d_mtc, w * bands, p, 9, d_disp, d_best)
            ctx.sync()
            assert np.array_equal(ctx.download(d_disp, (h, w), np.int32), sentinel)
        # and the entry point still works afterwards
        run_both(ctx, 200, 100, 1, 40, 5, match_fn=1, match_interval=1, match_max=20)
    finally:
        ctx.close()


def test_noalloc_with_paired_columns_and_bounded_occupancy():
    """The launch shapes added late in round 2 inside the no-allocation call: a 128-level slab with its columns in
    pairs (an odd width: the slab is one column wider than the unpaired one) and the LDS floors of both passes."""
    ctx = capi.Context(0)
    try:
        with ctx.options(paired=1, a_lds_kb=26, b_waves=8):
            run_both(ctx, 151, 60, 1, 128, 9, match_fn=1, match_interval=1, match_max=30)
            run_both(ctx, 90, 40, 1, 100, 5, match_fn=2, match_max=1000)
        with ctx.options(a_lds_kb=54, b_waves=8):
            run_both(ctx, 120, 50, 1, 256, 9, match_fn=1, match_interval=1)
        ctx.sync()
        assert ctx.workspace_bytes() > 0       # (run_both's allocating call grew it; the next test checks the other side)
    finally:
        ctx.close()


def test_noalloc_grows_no_workspace():
    ctx = capi.Context(0)
    try:
        w, h, b, D = 160, 90, 1, 64
        left, right, _ = synth.scene_pair(w, h, D)
        p = capi.match_params(w, h, b, D, match_fn=1, match_interval=1)
        d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
        d_disp, d_best = ctx.alloc(w * h * 4), ctx.alloc(w * h * 4)
        ctx.fused_box_wta_noalloc(d_ref, d_mtc, w, p, 9, d_disp, d_best)
        ctx.sync()
        assert ctx.workspace_bytes() == 0
    finally:
        ctx.close()


def test_noalloc_under_graph_capture():
    """The callers the entry point is for include code under HIP graph capture: the call is captured on a
    stream (nothing may allocate, synchronise or free during capture), the graph is replayed twice with new
    images in the same device buffers, and every replay equals the oracle."""
    import torch
    w, h, b, D, win = 192, 96, 1, 48, 7
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    stream = torch.cuda.Stream(device=dev)
    try:
        p = capi.match_params(w, h, b, D, match_fn=1, match_interval=1, match_max=50)
        t_ref = torch.zeros((h, w), dtype=torch.uint8, device=dev)
        t_mtc = torch.zeros((h, w), dtype=torch.uint8, device=dev)
        t_disp = torch.zeros((h, w), dtype=torch.int32, device=dev)
        t_best = torch.zeros((h, w), dtype=torch.float32, device=dev)
        ctx.set_stream(stream.cuda_stream)
        ctx.fused_box_wta_noalloc(t_ref.data_ptr(), t_mtc.data_ptr(), w, p, win, t_disp.data_ptr(), t_best.data_ptr())
        ctx.sync()                                   # warm-up outside the capture (module load)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            ctx.fused_box_wta_noalloc(t_ref.data_ptr(), t_mtc.data_ptr(), w, p, win, t_disp.data_ptr(),
                                      t_best.data_ptr())
        op = O.make_params(w, h, b, disp_min=0, disp_max=D - 1, aggr_window_size=win, match_fn=1, match_interval=1,
                           match_max=50)
        for seed in (11, 12):
            left, right, _ = synth.scene_pair(w, h, D, n_bands=b, seed=seed)
            t_ref.copy_(torch.from_numpy(left.reshape(h, w)))
            t_mtc.copy_(torch.from_numpy(right.reshape(h, w)))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            odisp, obest = O.compute_correspondence(left, right, op)[:2]
            util.assert_bit_equal(t_disp.cpu().numpy(), odisp, "captured call, seed %d: disparity" % seed)
            util.assert_bit_equal(t_best.cpu().numpy(), obest, "captured call, seed %d: best cost" % seed)
        assert ctx.workspace_bytes() == 0
    finally:
        ctx.set_stream(None)
        ctx.close()
