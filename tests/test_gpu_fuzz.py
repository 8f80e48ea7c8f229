"""Seeded differential test of the whole device pipeline against the oracle: random image shapes,
band counts, disparity ranges / steps / directions, matching costs, aggregation operators and
iteration counts, shiftable windows, sub-pixel stages and all three optimisers.  Every case must be
bit-identical (disparity, cost at the winner, float disparity, depth map).  Bayesian aggregation is
left to its own tolerance tests."""
import os

import numpy as np
import pytest

import util
from stereomatch_amd import capi, matcher, synth

pytestmark = pytest.mark.gpu


def draw(seed):
    r = np.random.RandomState(1000 + seed)
    w, h = int(r.randint(2, 97)), int(r.randint(1, 41))
    if seed >= 200 and seed % 3 != 0:                 # larger frames: several tiles / bands per kernel
        w, h = int(r.randint(90, 300)), int(r.randint(33, 120))
    b = int(r.choice([1, 4]))
    step = float(r.choice([1.0, 1.0, 1.0, 2.0, 0.5, 0.25]))
    span = int(r.choice([1, 3, 7, 15, 33, 64, 100, 129, 31, 63, 127, 255, 39]))
    if step < 1.0:
        span = min(span, 33)
    if step > 1.0:
        span = max(span, 3)
    dmin = int(r.choice([0, 0, 0, 2, -3]))
    seq = dict(disp_min=dmin, disp_max=dmin + span, disp_step=step,
               disp_scale=float(r.choice([1.0, 2.0, 16.0])))
    if r.rand() < 0.3:
        seq.update(frame_ref=1, frame_match=0)
    kw = dict(match_fn=int(r.choice([1, 2])), match_max=int(r.choice([1000, 50, 20, 5, 2, 1])),
              match_interval=int(r.rand() < 0.5), match_interp=int(r.choice([1, 3])))
    if step < 1.0 and r.rand() < 0.5:
        kw["match_interpolated"] = 1
    kw["aggr_fn"] = int(r.choice([1, 1, 2, 3, 4]))
    kw["aggr_iter"] = int(r.choice([0, 1, 1, 2, 3, 5, 9]))
    kw["aggr_window_size"] = int(r.choice([1, 3, 5, 7, 9, 11, 13, 4, 6]))
    kw["aggr_minfilter"] = int(r.choice([0, 0, 3, 5, 9, 11, 4]))
    kw["diff_lambda"] = float(r.choice([0.15, 0.1]))
    kw["diff_beta"] = float(r.choice([0.5, 0.25]))
    kw["opt_fn"] = int(r.choice([1, 1, 1, 2, 3]))
    kw["opt_smoothness"] = float(r.choice([1.0, 5.0, 20.0, 50.0]))
    kw["opt_grad_thresh"] = float(r.choice([5.0, 8.0]))
    kw["opt_grad_penalty"] = float(r.choice([1.0, 2.0, 4.0]))
    kw["opt_occlusion_cost"] = int(r.choice([20, 5, 80]))
    kw["preproc_blur_iter"] = int(r.choice([0, 0, 0, 1, 2]))
    if r.rand() < 0.3:
        kw["aggr_subpixel"] = 1
    if step < 1.0 and r.rand() < 0.4:
        kw["aggr_collapse"] = 1
    if r.rand() < 0.4:
        kw["refine_subpix"] = 1
    structured = r.rand() < 0.6
    return w, h, b, seq, kw, structured


@pytest.mark.parametrize("seed", range(int(os.environ.get("SMHIP_FUZZ_N", "280"))))   # more on demand
def test_random_configuration(gpu_ctx, oracle, seed):
    w, h, b, seq, kw, structured = draw(seed)
    if structured:
        ref, mtc, _ = synth.scene_pair(w, h, 16, n_bands=b, seed=77 + seed)
    else:
        ref, mtc = synth.noise_pair(w, h, b, seed=99 + seed)
    p = oracle.make_params(w, h, b, **seq, **kw)
    odisp, obest, odepth, _, ofd = oracle.compute_correspondence_f(ref, mtc, p)
    # lines shorter than the box window are outside the reference's domain (BoxFilter.cpp:114-124
    # reads beyond them): the library refuses, like the reference's asserts
    refused = kw["aggr_fn"] == 1 and kw["aggr_iter"] > 0 and kw["aggr_window_size"] > min(w, h)
    for fused in (True, False):
        m = matcher.StereoMatcher(ctx=gpu_ctx, **seq, **kw)
        m.allow_fused = fused
        if refused:
            with pytest.raises(capi.SmhipError, match="BoxFilter"):
                m.ComputeCorrespondence(ref, mtc)
            continue
        disp, best, depth = m.ComputeCorrespondence(ref, mtc)
        what = "seed %d %s %s fused=%s" % (seed, seq, kw, fused)
        assert np.array_equal(disp, odisp), what
        util.assert_bit_equal(best, obest, what)
        util.assert_bit_equal(np.asarray(m.m_float_disparity, np.float32), ofd, what)
        assert np.array_equal(depth, odepth), what


@pytest.mark.parametrize("seed", range(int(os.environ.get("SMHIP_FUZZ_BAYES_N", "16"))))
def test_random_bayesian_configuration(gpu_ctx, oracle, seed):
    """aggr_fn 5 over random shapes / priors, in its stated tolerance (tests/test_gpu_ops.py):
    energies within 1e-4 * max(1, |E|) after up to 8 iterations, winners equal on >= 99 %."""
    r = np.random.RandomState(5000 + seed)
    w, h, b = int(r.randint(3, 80)), int(r.randint(2, 40)), int(r.choice([1, 4]))
    span = int(r.choice([3, 7, 15, 40, 63, 64, 99, 127]))
    seq = dict(disp_min=0, disp_max=span)
    kw = dict(aggr_fn=5, aggr_iter=int(r.randint(1, 9)), match_fn=2, match_max=int(r.choice([12, 20, 1000])),
              match_interval=int(r.rand() < 0.5), diff_scale_cost=float(r.choice([0.016, 0.01, 0.002])),
              diff_mu=float(r.choice([0.5, 0.25])), diff_sigmaP=float(r.choice([0.4, 0.1, 1.0, 4.0])),
              diff_epsP=float(r.choice([0.01, 0.05, 0.0])))
    ref, mtc, _ = synth.scene_pair(w, h, min(span + 1, 16), n_bands=b, seed=300 + seed)
    p = oracle.make_params(w, h, b, **seq, **kw)
    want = oracle.aggregate(oracle.raw_costs(ref, mtc, p), p)
    m = matcher.StereoMatcher(ctx=gpu_ctx, **seq, **kw)
    disp, _, _ = m.ComputeCorrespondence(ref, mtc)
    got = m.download_cost()
    err = np.abs(got.astype(np.float64) - want) / np.maximum(1.0, np.abs(want))
    assert np.isfinite(got).all() and err.max() <= 1e-4, "seed %d %s: %.3g" % (seed, kw, err.max())
    assert np.mean(disp == oracle.wta(want)[0]) >= 0.99


@pytest.mark.parametrize("seed", range(int(os.environ.get("SMHIP_FUZZ_SLAB_N", "40"))))
def test_random_slab_partition(gpu_ctx, oracle, seed):
    """The multi-GPU decomposition on random shapes: the disparity range cut into random slabs,
    smhip_fused_box_wta_keys per slab, keys merged by MIN (what the all-reduce does), decoded --
    equal to the oracle's single-volume result."""
    from stereomatch_amd import capi as C_, sharding
    r = np.random.RandomState(9000 + seed)
    w, h, b = int(r.randint(12, 200)), int(r.randint(11, 60)), int(r.choice([1, 4]))
    D = int(r.choice([8, 16, 33, 64, 100, 128, 160, 256, 300]))
    win = int(r.choice([3, 5, 7, 9, 11]))
    sign = int(r.choice([0, 1]))
    kw = dict(match_fn=int(r.choice([1, 2])), match_interval=int(r.rand() < 0.6),
              match_max=int(r.choice([1000, 20, 2])))
    # launch-shape knobs drawn last, so that the shapes of the earlier seeds stay what they were
    r2 = np.random.RandomState(77000 + seed)
    if r2.rand() < 0.3:
        win = int(r2.choice([13, 15, 17]))
        h = max(h, win)
        w = max(w, win)
    opts = dict(paired=int(r2.choice([-1, 0, 1])), a_lds_kb=int(r2.choice([-1, 0, 26, 54])),
                b_waves=int(r2.choice([-1, 0, 6, 8, 12])), keybuf=int(r2.choice([-1, -1, 0, 1, 3])),
                b_ahead=int(r2.choice([-1, 1, 2])), b_lred=int(r2.choice([-1, -1, 0, 1])))
    seq = dict(disp_min=0, disp_max=D - 1)
    if sign:
        seq.update(frame_ref=1, frame_match=0)
    left, right, _ = synth.scene_pair(w, h, min(D, 32), n_bands=b, seed=500 + seed)
    p_o = oracle.make_params(w, h, b, **seq, aggr_window_size=win, **kw)
    odisp, obest, _, _ = oracle.compute_correspondence(left, right, p_o)
    p = C_.match_params(w, h, b, D, frame_diff_sign=-1 if sign else 1, **kw)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    n_slabs = int(r.randint(1, min(6, D // 2) + 1))
    cuts = sorted(set([0, D] + [int(x) for x in r.randint(1, D, size=n_slabs - 1)]))
    keys = gpu_ctx.alloc(w * h * 8)
    merged = None
    with gpu_ctx.options(**opts):
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            gpu_ctx.fused_box_wta_keys(d_ref, d_mtc, w * b, p, win, lo, hi - lo, keys)
            k = gpu_ctx.download(keys, (w, h), np.int64)
            merged = k if merged is None else np.minimum(merged, k)
    d, c = sharding.decode_keys(merged)
    assert np.array_equal(d.T, odisp), (seed, cuts, opts)
    util.assert_bit_equal(np.ascontiguousarray(c.T), obest, "best cost %s" % cuts)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SMHIP_FUZZ_PIPE_N", "120"))))
def test_random_slab_pipeline(gpu_ctx, oracle, seed):
    """Disparity-slab sharding of the STAGED pipeline (matcher.ComputeSlabKeys): the random
    configurations above restricted to what shards (WTA, integer steps, per-level aggregators), the
    level range cut into random slabs, each slab through PreProcess / RawCosts / Aggregate /
    smhip_pad_wta_keys (or the fused calls where they apply), keys merged by MIN -- what the int64
    all-reduce does -- and decoded: equal to the oracle's single-volume disparities and costs."""
    from stereomatch_amd import sharding
    w, h, b, seq, kw, structured = draw(3000 + seed)
    if seq["disp_step"] < 1.0:
        seq["disp_step"] = 1.0
    seq["disp_max"] = seq["disp_min"] + max(4, int((seq["disp_max"] - seq["disp_min"]) // seq["disp_step"] * seq["disp_step"]))
    kw.update(opt_fn=1, aggr_subpixel=0, aggr_collapse=0, refine_subpix=0)
    kw.pop("match_interpolated", None)
    if structured:
        ref, mtc, _ = synth.scene_pair(w, h, 16, n_bands=b, seed=77 + seed)
    else:
        ref, mtc = synth.noise_pair(w, h, b, seed=99 + seed)
    p = oracle.make_params(w, h, b, **seq, **kw)
    if kw["aggr_fn"] == 1 and kw["aggr_iter"] > 0 and kw["aggr_window_size"] > min(w, h):
        return                                       # outside the box filter's domain (see above)
    odisp, obest, _, _, _ = oracle.compute_correspondence_f(ref, mtc, p)
    D = p.disp_n
    r = np.random.RandomState(7000 + seed)
    n_slabs = int(r.randint(1, min(5, D // 2) + 1))
    cuts = [0, D]
    while len(cuts) < n_slabs + 1:                   # every slab holds at least two levels
        c = int(r.randint(2, D - 1))
        if all(abs(c - o) >= 2 for o in cuts):
            cuts.append(c)
        elif D < 4 * n_slabs:
            break
    cuts = sorted(cuts)
    for fused in (True, False):
        m = matcher.StereoMatcher(ctx=gpu_ctx, **seq, **kw)
        m.allow_fused = fused
        merged = None
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            keys = m.ComputeSlabKeys(ref, mtc, lo, hi - lo)
            k = gpu_ctx.download(keys, (w, h), np.int64)
            merged = k if merged is None else np.minimum(merged, k)
        d, c = sharding.decode_keys(merged)
        what = "seed %d %s %s cuts=%s fused=%s" % (seed, seq, kw, cuts, fused)
        assert np.array_equal(d.T, odisp), what
        util.assert_bit_equal(np.ascontiguousarray(c.T), obest, what)
