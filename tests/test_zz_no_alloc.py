"""smhip_fused_box_wta_noalloc -- the entry point that allocates nothing (its slab, interval images and keys
live in a static device array) -- against smhip_fused_box_wta, through the C-ABI, bit for bit: two calls in a
row, the second with a smaller input (the array keeps the first call's contents), then a problem that does not
fit (refused, nothing launched) and a call after it."""
import numpy as np
import pytest

import util
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
