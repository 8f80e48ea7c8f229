"""The fused raw-cost -> box -> pad -> WTA path against the oracle pipeline, bit for bit."""
import numpy as np
import pytest

import util
from stereomatch_amd import capi, matcher, synth

pytestmark = pytest.mark.gpu


def oracle_run(oracle, ref, mtc, **kw):
    h, w, b = ref.shape
    p = oracle.make_params(w, h, b, **kw)
    return oracle.compute_correspondence(ref, mtc, p)


CASES = [
    # w, h, b, D, win, params
    (96, 40, 1, 16, 9, dict(match_fn=1, match_interval=1)),
    (96, 40, 4, 16, 9, dict(match_fn=2)),
    (80, 33, 1, 5, 3, dict(match_fn=1)),
    (70, 30, 1, 33, 7, dict(match_fn=2, match_max=5)),
    (120, 24, 1, 64, 9, dict(match_fn=1, match_interval=1)),
    (150, 20, 1, 100, 5, dict(match_fn=1, match_max=2)),
    (160, 24, 1, 128, 9, dict(match_fn=1, match_interval=1)),
    (200, 20, 1, 192, 9, dict(match_fn=2)),
    (280, 22, 1, 256, 9, dict(match_fn=1, match_interval=1)),
    (320, 18, 1, 300, 9, dict(match_fn=1, match_interval=1, match_max=20)),
    (90, 35, 4, 24, 9, dict(match_fn=1, match_interval=1, match_interpolated=1)),
]


@pytest.mark.parametrize("w,h,b,D,win,kw", CASES)
def test_fused_matches_oracle(gpu_ctx, oracle, w, h, b, D, win, kw):
    left, right, _ = synth.scene_pair(w, h, D, n_bands=b)
    odisp, obest, odepth, _ = oracle_run(oracle, left, right, disp_max=D - 1,
                                         aggr_window_size=win, **kw)
    m = matcher.StereoMatcher(ctx=gpu_ctx, disp_max=D - 1, aggr_window_size=win, **kw)
    disp, best, depth = m.ComputeCorrespondence(left, right)
    assert m.used_fused
    util.assert_bit_equal(disp, odisp, "fused disparity")
    util.assert_bit_equal(best, obest, "fused best cost")
    assert np.array_equal(depth, odepth)
    # and the staged path gives the same
    m.allow_fused = False
    disp2, best2, _ = m.ComputeCorrespondence(left, right)
    assert not m.used_fused
    util.assert_bit_equal(disp2, odisp, "staged disparity")
    util.assert_bit_equal(best2, obest, "staged best cost")


def test_fused_tie_stress(gpu_ctx, oracle):
    """match_max 2 on pure noise: almost every window sum ties."""
    left, right = synth.noise_pair(128, 48, 1)
    for mm in (1, 2):
        odisp, obest, _, _ = oracle_run(oracle, left, right, disp_max=63, aggr_window_size=9,
                                        match_fn=1, match_max=mm)
        m = matcher.StereoMatcher(ctx=gpu_ctx, disp_max=63, aggr_window_size=9, match_fn=1,
                                  match_max=mm)
        disp, best, _ = m.ComputeCorrespondence(left, right)
        util.assert_bit_equal(disp, odisp, "tie-stress disparity")
        util.assert_bit_equal(best, obest, "tie-stress best cost")


def test_fused_negative_direction_and_offset(gpu_ctx, oracle):
    left, right, _ = synth.scene_pair(100, 30, 20)
    kw = dict(disp_min=3, disp_max=22, frame_ref=1, frame_match=0, aggr_window_size=5,
              match_fn=2, disp_scale=8.0)
    odisp, obest, odepth, _ = oracle_run(oracle, left, right, **kw)
    m = matcher.StereoMatcher(ctx=gpu_ctx, **kw)
    disp, best, depth = m.ComputeCorrespondence(left, right)
    util.assert_bit_equal(disp, odisp, "disparity")
    util.assert_bit_equal(best, obest, "best")
    assert np.array_equal(depth, odepth)


@pytest.mark.parametrize("D,slabs", [(64, 2), (96, 3), (256, 4), (100, 3)])
def test_disparity_slabs_merge_to_full_result(gpu_ctx, oracle, D, slabs):
    """Slab keys min-merged (what the multi-GPU reduce does) == single-device result."""
    w, h = 140, 26
    left, right, _ = synth.scene_pair(w, h, D)
    odisp, obest, _, _ = oracle_run(oracle, left, right, disp_max=D - 1, aggr_window_size=9,
                                    match_fn=1, match_interval=1)
    p = capi.match_params(w, h, 1, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    keys = gpu_ctx.alloc(w * h * 8)
    merged = None
    bounds = [D * i // slabs for i in range(slabs + 1)]
    for i in range(slabs):
        gpu_ctx.fused_box_wta_keys(d_ref, d_mtc, w, p, 9, bounds[i], bounds[i + 1] - bounds[i], keys)
        k = gpu_ctx.download(keys, (w, h), np.int64)
        merged = k if merged is None else np.minimum(merged, k)
    dk = gpu_ctx.upload(merged)
    disp, best = gpu_ctx.alloc(w * h * 4), gpu_ctx.alloc(w * h * 4)
    gpu_ctx.keys_decode(dk, w, h, disp, best)
    util.assert_bit_equal(gpu_ctx.download(disp, (h, w), np.int32), odisp, "merged disparity")
    util.assert_bit_equal(gpu_ctx.download(best, (h, w), np.float32), obest, "merged best cost")


def test_full_size_properties(gpu_ctx):
    """BASELINE config 3 (1920x1080, D=256, BT-SAD, 9x9 box, WTA) at full size, through
    size-independent properties: slab-merge invariance, determinism, and agreement of the
    fused path with the staged operators (which the small cases tie to the oracle)."""
    w, h, D = 1920, 1080, 256
    left, right, _ = synth.scene_pair(w, h, D)
    m = matcher.StereoMatcher(ctx=gpu_ctx, disp_max=D - 1, aggr_window_size=9, match_fn=1,
                              match_interval=1)
    disp, best, _ = m.ComputeCorrespondence(left, right)
    assert m.used_fused and disp.min() >= 0 and disp.max() < D
    disp_b, best_b, _ = m.ComputeCorrespondence(left, right)
    assert np.array_equal(disp, disp_b) and np.array_equal(best, best_b)
    m.allow_fused = False
    disp2, best2, _ = m.ComputeCorrespondence(left, right)
    util.assert_bit_equal(disp2, disp, "staged vs fused disparity")
    util.assert_bit_equal(best2, best, "staged vs fused best cost")
    # slabs of 128 merged on the host
    p = m.m_params
    keys = gpu_ctx.alloc(w * h * 8)
    merged = None
    for d0 in (0, 128):
        gpu_ctx.fused_box_wta_keys(m.d_ref, m.d_mtc, w, p, 9, d0, 128, keys)
        k = gpu_ctx.download(keys, (w, h), np.int64)
        merged = k if merged is None else np.minimum(merged, k)
    assert np.array_equal((merged & 0xFFFFFFFF).astype(np.int32).T, disp)   # keys are column-major
