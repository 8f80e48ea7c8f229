"""BASELINE configs 1-2 on tsukuba: exp1 (9x9 box) and exp2 (+ 9x9 min filter) grids, exp4-style
binomial / diffusion / membrane runs.  Disparity PGMs must be byte-identical to the reference's
recorded md5s (five anchors) and to the oracle everywhere else."""
import numpy as np
import pytest

import util
from stereomatch_amd import matcher, pnm

pytestmark = pytest.mark.gpu


def run(gpu_ctx, tsukuba, kw, fused=True):
    ref, mtc = tsukuba
    m = matcher.StereoMatcher(ctx=gpu_ctx, **util.TSUKUBA_SEQ, **kw)
    m.allow_fused = fused
    disp, best, depth = m.ComputeCorrespondence(ref, mtc)
    return m, disp, best, depth


def test_reference_anchors(gpu_ctx, tsukuba):
    a = util.anchors()["reference"]["md5"]
    grid = dict(util.exp_grid() + util.exp4_grid())
    for name, want in a.items():
        for fused in (True, False):
            _, _, _, depth = run(gpu_ctx, tsukuba, grid[name], fused)
            assert util.md5(pnm.pgm_bytes(depth)) == want, (name, fused)


def test_ssd09_data_energy(gpu_ctx, tsukuba, oracle):
    """final_energy's data term is the raster-order float sum of the best costs
    (StcGraphCut.cpp:56-59); with it the SSD09 energy matches the reference's 24421442."""
    grid = dict(util.exp_grid())
    m, disp, best, _ = run(gpu_ctx, tsukuba, grid["SSD09"])
    ref, mtc = tsukuba
    h, w, b = ref.shape
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **grid["SSD09"])
    odisp, obest, _, e = oracle.compute_correspondence(ref, mtc, p)
    util.assert_bit_equal(disp, odisp, "SSD09 disparity")
    util.assert_bit_equal(best, obest, "SSD09 best cost")
    assert e == util.anchors()["reference"]["final_energy"]["SSD09"]


@pytest.mark.parametrize("name,kw", util.exp_grid() + util.exp4_grid())
def test_grid_vs_oracle_golden(gpu_ctx, tsukuba, name, kw):
    want = util.anchors()["oracle"]["md5"][name]
    _, _, _, depth = run(gpu_ctx, tsukuba, kw)
    assert util.md5(pnm.pgm_bytes(depth)) == want, name


@pytest.mark.parametrize("name", ["SAD09t02", "SSDmf09b", "SADbf04", "SADdiff10", "SADmem10b05"])
def test_full_dsi_vs_oracle(gpu_ctx, tsukuba, oracle, name):
    """Whole aggregated DSI (not only the winner) bit-identical to the oracle."""
    kw = dict(util.exp_grid() + util.exp4_grid())[name]
    ref, mtc = tsukuba
    h, w, b = ref.shape
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **kw)
    want = oracle.aggregate(oracle.raw_costs(ref, mtc, p), p)
    m, _, _, _ = run(gpu_ctx, tsukuba, kw, fused=False)
    util.assert_bit_equal(m.download_cost(), want, name)


@pytest.mark.parametrize("kw", [
    dict(disp_step=0.5, match_fn=1, match_interval=1, match_interpolated=1, aggr_window_size=9),
    dict(disp_step=0.25, match_fn=2, match_interp=1, match_max=20, aggr_window_size=5,
         aggr_minfilter=5),
    dict(disp_step=0.5, match_fn=1, match_interval=1, aggr_fn=3, aggr_iter=4),
])
def test_subpixel_step_vs_oracle(gpu_ctx, tsukuba, oracle, kw):
    """disp_step = 1/2, 1/4 end to end (staged path: raw costs on up-sampled lines, aggregation,
    PadCosts, WTA, depth scaling with the fractional step).  No reference output exists for
    these settings: checked against the oracle only (parity unpinned)."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    seq = dict(util.TSUKUBA_SEQ)
    p = oracle.make_params(w, h, b, **seq, **kw)
    odisp, obest, odepth, _ = oracle.compute_correspondence(ref, mtc, p)
    m = matcher.StereoMatcher(ctx=gpu_ctx, **seq, **kw)
    disp, best, depth = m.ComputeCorrespondence(ref, mtc)
    assert not m.used_fused and m.m_disp_n == p.disp_n
    util.assert_bit_equal(disp, odisp, "disparity")
    util.assert_bit_equal(best, obest, "best cost")
    assert np.array_equal(depth, odepth.reshape(depth.shape))


@pytest.mark.parametrize("kw", [
    dict(refine_subpix=1, match_fn=2, aggr_window_size=9),
    dict(aggr_subpixel=1, refine_subpix=1, match_fn=1, aggr_window_size=7),
    dict(disp_step=0.5, aggr_subpixel=1, aggr_collapse=1, refine_subpix=1, match_interval=1),
    dict(disp_step=0.25, aggr_collapse=1, refine_subpix=1, match_fn=2, match_max=20),
    dict(disp_step=0.5, aggr_collapse=1, aggr_fn=2, aggr_iter=2),
    dict(disp_step=0.5, refine_subpix=1, aggr_minfilter=3),
])
def test_subpixel_stages_vs_oracle(gpu_ctx, tsukuba, oracle, kw):
    """AggrSubPixelFit / AggrCollapse / Refine end to end, including the float disparities
    (no reference output for these settings: parity unpinned)."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    seq = dict(util.TSUKUBA_SEQ)
    p = oracle.make_params(w, h, b, **seq, **kw)
    odisp, obest, odepth, _, ofd = oracle.compute_correspondence_f(ref, mtc, p)
    m = matcher.StereoMatcher(ctx=gpu_ctx, **seq, **kw)
    disp, best, depth = m.ComputeCorrespondence(ref, mtc)
    assert not m.used_fused
    util.assert_bit_equal(disp, odisp, "disparity")
    util.assert_bit_equal(best, obest, "best cost")
    util.assert_bit_equal(m.m_float_disparity, ofd, "float disparity")
    assert np.array_equal(depth, odepth)
    if kw.get("refine_subpix") and (kw.get("aggr_subpixel") or not kw.get("aggr_collapse")):
        assert np.any(m.m_float_disparity != np.round(m.m_float_disparity * 4) / 4)


def test_preprocess_blur_pipeline(gpu_ctx, tsukuba, oracle):
    """preproc_blur_iter 2 in front of the SSD / 9x9 box / WTA run (oracle only: parity unpinned)."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    kw = dict(preproc_blur_iter=2, match_fn=2, aggr_window_size=9)
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **kw)
    odisp, obest, odepth, _ = oracle.compute_correspondence(ref, mtc, p)
    for fused in (True, False):
        m, disp, best, depth = run(gpu_ctx, tsukuba, kw, fused)
        util.assert_bit_equal(disp, odisp, "disparity")
        util.assert_bit_equal(best, obest, "best cost")
        assert np.array_equal(depth, odepth)


@pytest.mark.parametrize("kw", [
    dict(opt_fn=3, match_fn=1, aggr_window_size=1, aggr_iter=0, opt_smoothness=20.0, opt_grad_thresh=8.0,
         opt_grad_penalty=2.0),
    dict(opt_fn=3, match_fn=1, match_interval=1, aggr_window_size=3, opt_smoothness=5.0),
])
def test_scanline_optimisation_vs_oracle(gpu_ctx, tsukuba, oracle, kw):
    """opt_fn 3 (OptSO) end to end on tsukuba, incl. final_energy's inputs (best cost); oracle
    only: parity unpinned."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **kw)
    odisp, obest, odepth, _ = oracle.compute_correspondence(ref, mtc, p)
    m, disp, best, depth = run(gpu_ctx, tsukuba, kw)
    assert not m.used_fused
    util.assert_bit_equal(disp, odisp, "disparity")
    util.assert_bit_equal(best, obest, "cost at the chosen level")
    assert np.array_equal(depth, odepth)
    assert np.mean(disp != oracle.wta(m.download_cost())[0]) > 0.01      # it is not just WTA


def test_bayesian_diffusion_vs_oracle(gpu_ctx, tsukuba, oracle):
    """aggr_fn 5 (AggrBayesian) with the settings StcDiffusion.cpp:173-178 suggests, end to end on
    tsukuba.  TOLERANCE MODE (device exp / log, see tests/test_gpu_ops.py BAYES_RTOL): energies
    within 1e-4 relative after 10 iterations, winners equal on >= 99.5 % of the pixels.  Oracle
    only: parity unpinned."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    kw = dict(aggr_fn=5, aggr_iter=10, match_fn=2, match_max=12, diff_scale_cost=0.016, diff_mu=0.5,
              diff_sigmaP=0.4, diff_epsP=0.01)
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **kw)
    ocost = oracle.aggregate(oracle.raw_costs(ref, mtc, p), p)
    odisp, _, _, _ = oracle.compute_correspondence(ref, mtc, p)
    m, disp, best, depth = run(gpu_ctx, tsukuba, kw)
    assert not m.used_fused
    cost = m.download_cost()
    err = np.abs(cost.astype(np.float64) - ocost) / np.maximum(1.0, np.abs(ocost))
    assert err.max() <= 1e-4, "max relative deviation %.3g" % err.max()
    agree = np.mean(disp == odisp)
    assert agree >= 0.995, "winners agree on %.5f of the pixels" % agree
    assert np.mean(disp != oracle.wta(oracle.raw_costs(ref, mtc, p))[0]) > 0.05   # it did aggregate


@pytest.mark.parametrize("kw", [
    dict(opt_fn=2, match_fn=1, aggr_iter=0, opt_smoothness=20.0, opt_occlusion_cost=20),       # exp3 DPm1o20s0020
    dict(opt_fn=2, match_fn=1, match_interval=1, match_max=10, aggr_iter=0, opt_smoothness=20.0,
         opt_occlusion_cost=20),                                                               # ...bt10
    dict(opt_fn=2, match_fn=1, aggr_iter=0, opt_smoothness=100.0, opt_occlusion_cost=80),      # exp5 DPm1o80s0100
    dict(opt_fn=2, match_fn=2, aggr_window_size=5, opt_smoothness=50.0, opt_occlusion_cost=50),
])
def test_dynamic_programming_vs_oracle(gpu_ctx, tsukuba, oracle, kw):
    """opt_fn 2 (OptDP) end to end on tsukuba with settings of Scripts/exp3_costGlobal.txt:21-45 and
    exp5_lambda.txt:17-50, plus one on box-filtered (fractional) costs; oracle only: the reference
    holds no result image for these, parity unpinned."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **kw)
    odisp, obest, odepth, _ = oracle.compute_correspondence(ref, mtc, p)
    m, disp, best, depth = run(gpu_ctx, tsukuba, kw)
    assert not m.used_fused
    util.assert_bit_equal(disp, odisp, "disparity")
    util.assert_bit_equal(best, obest, "cost at the chosen level")
    assert np.array_equal(depth, odepth)
    assert np.mean(disp != oracle.wta(m.download_cost())[0]) > 0.01      # it is not just WTA


def test_stage_dsi_checksums(gpu_ctx, tsukuba):
    """The device's aggregated DSI against the committed per-stage checksums (no oracle in the loop)."""
    want = util.anchors()["oracle"]["dsi_crc32"]
    for name, kw in util.stage_cases():
        m, _, _, _ = run(gpu_ctx, tsukuba, kw, fused=False)
        assert util.crc32(m.download_cost()) == want[name], name
