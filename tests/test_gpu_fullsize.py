"""BASELINE configs 4 and 5 at their full sizes, checked through locality: a band of rows of the
full-size result must equal, bit for bit, the oracle run on the matching band of the input
(plus the halo the operator needs), because every operator of the path is local in y apart from
the box filter's vertical running sum, which starts at the top row.  The oracle cannot run the
full volumes (17 GB each) in test time; the bands take a few seconds."""
import numpy as np
import pytest

import util
from stereomatch_amd import capi, sharding, synth

pytestmark = pytest.mark.gpu


def _band(img, y0, y1):
    return np.ascontiguousarray(img[y0:y1])


def test_config4_iterated_aggregation_full_size(gpu_ctx, oracle):
    """3840x2160, D = 512 (config 4): raw costs, then 10 diffusion iterations / 4 binomial iterations
    through the multi-iteration passes; rows 1000..1009 against the oracle."""
    W, H, D = 3840, 2160, 512
    left, right, _ = synth.scene_pair(W, H, 256)
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    re = W * D
    a, b = gpu_ctx.alloc(re * H * 4), gpu_ctx.alloc(re * H * 4)
    y0, y1 = 1000, 1010
    for name, halo, run, ref_iter in (
            ("diffusion x10", 10,
             lambda: gpu_ctx.diffusion_run(a, b, None, W, H, D, re, 0.15, 0.0, 10),
             lambda c: oracle.diffusion_iter(c, 0.15, 0.0, None)),
            ("binomial x4", 8,
             lambda: gpu_ctx.convolve_14641_run(a, b, W, H, D, re, 4),
             oracle.convolve_14641)):
        gpu_ctx.raw_costs(d_ref, d_mtc, W, p, a, re)
        in_b = run()
        res = b if in_b else a
        got = np.empty((y1 - y0, W, D), np.float32)
        gpu_ctx._chk(gpu_ctx.L.smhip_memcpy_d2h(gpu_ctx.h, got.ctypes.data, res.ptr + y0 * re * 4, got.nbytes))
        b0, b1 = y0 - halo, y1 + halo
        po = oracle.make_params(W, b1 - b0, 1, disp_max=D - 1, match_fn=1, match_interval=1)
        want = oracle.raw_costs(_band(left, b0, b1), _band(right, b0, b1), po)
        n_iter = 10 if halo == 10 else 4
        for _ in range(n_iter):
            want = ref_iter(want)
        util.assert_bit_equal(got, want[halo:halo + (y1 - y0)], name)


@pytest.mark.parametrize("n_iter", [4, 8])
def test_config4_fused_iterated_path_full_size(gpu_ctx, oracle, n_iter):
    """3840x2160, D = 512 (config 4) through smhip_fused_iter_wta_keys: raw costs -> 4 / 8 binomial iterations ->
    PadCosts -> WTA keys without a raw or an aggregated DSI (8: between the raw-cost and the WTA end kernels), one
    call per 256 levels as a GPU's share would be; three bands of rows (top border, a segment boundary of the
    keys kernel, bottom border) against the oracle run on the matching band of the input plus the 2-rows-per-
    iteration halo."""
    W, H, D = 3840, 2160, 512
    left, right, _ = synth.scene_pair(W, H, 256)
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    keys = gpu_ctx.alloc(W * H * 8)
    k = None
    for d0 in (0, 256):
        gpu_ctx.fused_iter_wta_keys(d_ref, d_mtc, W, p, 2, n_iter, d0, 256, keys)
        part = gpu_ctx.download(keys, (W, H), np.int64)
        k = part if k is None else np.minimum(k, part)
    halo = 2 * n_iter
    for y0, y1 in ((0, 6), (1020, 1030), (H - 5, H)):
        b0, b1 = max(0, y0 - halo), min(H, y1 + halo)
        po = oracle.make_params(W, b1 - b0, 1, disp_max=D - 1, match_fn=1, match_interval=1, aggr_fn=2,
                                aggr_iter=n_iter)
        wd, wb = oracle.compute_correspondence(_band(left, b0, b1), _band(right, b0, b1), po)[:2]
        gd, gb = sharding.decode_keys(np.ascontiguousarray(k[:, y0:y1].T))
        assert np.array_equal(gd, wd[y0 - b0:y1 - b0]), "rows %d..%d: disparity" % (y0, y1)
        util.assert_bit_equal(gb, wb[y0 - b0:y1 - b0], "rows %d..%d: best cost" % (y0, y1))
    assert len(np.unique(k & 0xFFFFFFFF)) > 100


def test_config5_slab_full_size(gpu_ctx, oracle):
    """7680x4320, one 128-level slab (levels 512..639 of D = 1024: config 5's per-GPU share) through
    the fused path; the top 24 rows against the oracle (the vertical running sum starts at row 0)."""
    W, H, D, d0, n = 7680, 4320, 1024, 512, 128
    left, right, _ = synth.scene_pair(W, H, 256)
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    keys = gpu_ctx.alloc(W * H * 8)
    gpu_ctx.fused_box_wta_keys(d_ref, d_mtc, W, p, 9, d0, n, keys)
    k = gpu_ctx.download(keys, (W, H), np.int64)
    rows, halo = 24, 4
    gd, gb = sharding.decode_keys(np.ascontiguousarray(k[:, :rows].T))
    po = oracle.make_params(W, rows + halo, 1, disp_min=d0, disp_max=d0 + n - 1, match_fn=1,
                            match_interval=1, aggr_window_size=9)
    cost = oracle.aggregate(oracle.raw_costs(_band(left, 0, rows + halo), _band(right, 0, rows + halo), po), po)
    wd, wb = oracle.wta(cost[:rows])
    assert np.array_equal(gd, wd + d0)
    util.assert_bit_equal(gb, wb, "slab best cost")
    # every key of the slab carries a level of the slab
    lv = (k & 0xFFFFFFFF)
    assert lv.min() >= d0 and lv.max() < d0 + n


def test_scanline_optimisers_full_size(gpu_ctx, oracle):
    """1920x1080, D = 256 (config 3's volume): OptSO and OptDP on the device-made raw DSI; scan-lines
    are independent, so a few rows of the full-size result must equal the oracle run on those rows."""
    W, H, D = 1920, 1080, 256
    left, right, _ = synth.scene_pair(W, H, D)
    p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1, match_max=20)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    re = W * D
    cost = gpu_ctx.alloc(re * H * 4)
    disp = gpu_ctx.alloc(W * H * 4)
    gpu_ctx.raw_costs(d_ref, d_mtc, W, p, cost, re)
    rows = [0, 537, 1079]
    dsi = np.empty((len(rows), W, D), np.float32)
    for i, y in enumerate(rows):
        gpu_ctx._chk(gpu_ctx.L.smhip_memcpy_d2h(gpu_ctx.h, dsi[i].ctypes.data, cost.ptr + y * re * 4, dsi[i].nbytes))
    img = np.ascontiguousarray(left.reshape(H, W, 1)[rows])
    q = oracle.make_params(W, len(rows), 1, disp_max=D - 1, opt_smoothness=20.0, opt_grad_thresh=8.0,
                           opt_grad_penalty=2.0, opt_occlusion_cost=20)
    for name, run, ref in (
            ("OptSO", lambda: gpu_ctx.opt_so(cost, W, H, D, re, d_ref, W, 1, 20.0, 8.0, 2.0, disp, None),
             oracle.opt_so),
            ("OptDP", lambda: gpu_ctx.opt_dp(cost, W, H, D, re, d_ref, W, 1, 20.0, 8.0, 2.0, 20, disp, None),
             oracle.opt_dp)):
        run()
        got = gpu_ctx.download(disp, (H, W), np.int32)[rows]
        want = ref(dsi, img, q)
        assert np.array_equal(got, want), "%s: %d pixels differ" % (name, np.sum(got != want))
        assert len(np.unique(got)) > 20                      # a real labelling, not a constant
