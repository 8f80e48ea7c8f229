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


@pytest.mark.parametrize("w,h,b,D,win,calls,opts", [
    (140, 70, 1, 96, 9, 3, {}),                 # 32-level calls: narrow kernels, merged by keys_min_kernel
    (150, 130, 1, 512, 9, 2, {}),               # 256-level calls: pass B takes the minimum where it stores
    (150, 130, 1, 512, 5, 2, {"keybuf": 0}),    # ... unbuffered key stores
    (151, 64, 1, 256, 9, 2, {"paired": 1}),     # 128-level calls, columns in pairs
    (96, 40, 4, 40, 7, 4, {}),                  # 10-level calls of a 4-band pair
    (120, 50, 1, 1024, 9, 2, {}),               # 512-level calls: two chunks each + the old keys
    (150, 130, 1, 512, 9, 2, {"store_policy": 18}),     # the large-frame policy: non-temporal slab and key stores
    (150, 130, 1, 256, 9, 2, {"store_policy": 18, "b_lred": 0}),
])
def test_keys_merge_entry_point(gpu_ctx, oracle, w, h, b, D, win, calls, opts):
    """smhip_fused_box_wta_keys_merge: a level range walked in several calls, each folding its keys into the
    buffer (min with the key already there), equals the oracle's WTA over the whole range."""
    left, right, _ = synth.scene_pair(w, h, min(D, 64), n_bands=b)
    odisp, obest, _, _ = oracle_run(oracle, left, right, disp_max=D - 1, aggr_window_size=win, match_fn=1,
                                    match_interval=1)
    p = capi.match_params(w, h, b, D, match_fn=1, match_interval=1)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    keys = gpu_ctx.alloc(w * h * 8)
    bounds = [D * i // calls for i in range(calls + 1)]
    with gpu_ctx.options(**opts):
        for i in range(calls):
            fn = gpu_ctx.fused_box_wta_keys if i == 0 else gpu_ctx.fused_box_wta_keys_merge
            fn(d_ref, d_mtc, w * b, p, win, bounds[i], bounds[i + 1] - bounds[i], keys)
    disp, best = gpu_ctx.alloc(w * h * 4), gpu_ctx.alloc(w * h * 4)
    gpu_ctx.keys_decode(keys, w, h, disp, best)
    util.assert_bit_equal(gpu_ctx.download(disp, (h, w), np.int32), odisp, "merged disparity")
    util.assert_bit_equal(gpu_ctx.download(best, (h, w), np.float32), obest, "merged best cost")


@pytest.mark.parametrize("pair,kw", [
    ("scene", dict(match_fn=1, match_interval=1, match_max=1000)),     # BASELINE config 3 as bench.py runs it
    ("noise", dict(match_fn=1, match_interval=1, match_max=1000)),     # the no-structure pair
    ("scene", dict(match_fn=1, match_interval=0, match_max=2)),        # the tie-stress run
    ("noise", dict(match_fn=1, match_interval=0, match_max=2)),
])
def test_config3_full_size_vs_oracle(gpu_ctx, oracle, pair, kw):
    """BASELINE config 3 at FULL size (1920x1080, D = 256, 9x9 box, WTA), device result against the
    oracle's whole frame, bit for bit: disparity and best cost (OptWTA, StcOptimize.cpp:26-56, after the
    running-sum box filter of BoxFilter.cpp:70-170).  The oracle runs its OpenMP build (bit-identical to
    the serial one: test_oracle_pins.py)."""
    from oracle import oracle as O
    w, h, D = 1920, 1080, 256
    if pair == "scene":
        left, right, _ = synth.scene_pair(w, h, D)
    else:
        left, right = synth.noise_pair(w, h, 1, seed=0x5EED0002)
    O.use_openmp(True)
    try:
        odisp, obest, odepth, _ = oracle_run(O, left, right, disp_max=D - 1, aggr_window_size=9, **kw)
    finally:
        O.use_openmp(False)
    # float-lane pass A (store policies; one-wave workgroups; four levels per lane with 1-KiB dwordx4 stores and
    # the wait states the compiler leaves out behind them), integer kernels
    for float_a, pol, a_nw in ((1, -1, -1), (1, 0, -1), (1, 18, -1), (0, -1, -1), (1, -1, 1), (1, -1, 4), (1, 0, 4)):
        with gpu_ctx.options(float_a=float_a, store_policy=pol, a_nw=a_nw):
            m = matcher.StereoMatcher(ctx=gpu_ctx, disp_max=D - 1, aggr_window_size=9, **kw)
            disp, best, depth = m.ComputeCorrespondence(left, right)
        assert m.used_fused
        util.assert_bit_equal(disp, odisp, "disparity (float_a=%d)" % float_a)
        util.assert_bit_equal(best, obest, "best cost (float_a=%d)" % float_a)
        assert np.array_equal(depth, odepth)


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


@pytest.mark.parametrize("w,h,b,D,win,sign,fn,mmax,interval", [
    (70, 40, 1, 36, 9, 1, 1, 1000, 1),
    (50, 33, 4, 64, 5, 1, 2, 30, 0),
    (45, 70, 1, 100, 4, -1, 1, 1000, 1),     # even window: generic passes
    (40, 20, 1, 260, 9, 1, 1, 7, 0),         # two 256-level chunks
    (64, 36, 1, 128, 13, 1, 2, 1000, 1),     # static windows beyond 11 (integer pass A)
    (33, 47, 1, 64, 3, 1, 1, 9, 1),          # blocks of 9 rows for window 3
    (40, 31, 1, 96, 5, -1, 1, 1000, 1),      # blocks of 10 rows for window 5
    (48, 19, 1, 68, 17, 1, 1, 12, 0),
    (30, 52, 4, 40, 15, 1, 2, 40, 0),
])
def test_fused_box_dsi(gpu_ctx, oracle, w, h, b, D, win, sign, fn, mmax, interval):
    """smhip_fused_box_dsi == smhip_raw_costs + smhip_box_filter (== oracle), whole DSI, bit for
    bit, including the positions PadCosts will overwrite later; then min filter + pad + WTA as
    smhip_min_pad_wta_keys == the staged oracle pipeline of exp2."""
    ref, mtc = synth.noise_pair(w, h, b, seed=synth.SEED_NOISE + D)
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    po = oracle.make_params(w, h, b, disp_max=D - 1, frame_ref=fr, frame_match=fm, match_fn=fn,
                            match_max=mmax, match_interval=interval, aggr_window_size=win)
    want = oracle.box_filter(oracle.raw_costs(ref, mtc, po), win)
    p = capi.match_params(w, h, b, D, frame_diff_sign=sign, match_fn=fn, match_max=mmax,
                          match_interval=interval)
    d_ref, d_mtc = gpu_ctx.upload(ref), gpu_ctx.upload(mtc)
    cost = gpu_ctx.alloc(w * h * D * 4)
    gpu_ctx.fused_box_dsi(d_ref, d_mtc, w * b, p, win, 0, D, cost, w * D)
    util.assert_bit_equal(gpu_ctx.download(cost, (h, w, D), np.float32), want, "fused box DSI")
    for mf in (3, 9):
        wd, wb = oracle.wta(oracle.pad_costs(oracle.min_filter(want, mf), po))
        keys = gpu_ctx.alloc(w * h * 8)
        disp, best = gpu_ctx.alloc(w * h * 4), gpu_ctx.alloc(w * h * 4)
        for tseg in (-1, 8, 64):                 # the library's segment length, the shortest, the longest
            with gpu_ctx.options(tseg=tseg):
                gpu_ctx.min_pad_wta_keys(cost, w * D, p, mf, 0, D, keys)
            gpu_ctx.keys_decode(keys, w, h, disp, best)
            assert np.array_equal(gpu_ctx.download(disp, (h, w), np.int32), wd)
            util.assert_bit_equal(gpu_ctx.download(best, (h, w), np.float32), wb, "min+pad+wta best")
    # a disparity slab of the same volume
    d0, n = 8, D - 12 - (D - 12) % 4
    if n >= 33:
        slab = gpu_ctx.alloc(w * h * n * 4)
        gpu_ctx.fused_box_dsi(d_ref, d_mtc, w * b, p, win, d0, n, slab, w * n)
        util.assert_bit_equal(gpu_ctx.download(slab, (h, w, n), np.float32), want[:, :, d0:d0 + n], "slab")
        keys = gpu_ctx.alloc(w * h * 8)
        gpu_ctx.min_pad_wta_keys(slab, w * n, p, 5, d0, n, keys)
        k = gpu_ctx.download(keys, (w, h), np.int64).T
        padded = oracle.pad_costs(oracle.min_filter(want, 5), po)[:, :, d0:d0 + n]
        wd, wb = oracle.wta(padded)
        from stereomatch_amd import sharding
        gd, gb = sharding.decode_keys(k)
        assert np.array_equal(gd, wd + d0)
        util.assert_bit_equal(gb, wb, "slab keys")


@pytest.mark.parametrize("w, h, D, sign", [(130, 150, 40, 1), (57, 65, 96, -1), (56, 64, 4, 1), (300, 17, 68, 1)])
@pytest.mark.parametrize("mf", [3, 5, 7, 9, 11])
def test_min_pad_wta_keys_windows(gpu_ctx, oracle, w, h, D, sign, mf):
    """smhip_min_pad_wta_keys for every window on shapes that cut its tiles: several pixel tiles and row
    segments, level counts that are not a multiple of the 32-level tile, a frame exactly one tile big,
    both frame directions; costs with many ties (small integers), so the lowest-level rule decides."""
    rng = np.random.default_rng(1000 * mf + D)
    cost = rng.integers(0, 6, (h, w, D)).astype(np.float32)
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    po = oracle.make_params(w, h, 1, disp_max=D - 1, frame_ref=fr, frame_match=fm)
    wd, wb = oracle.wta(oracle.pad_costs(oracle.min_filter(cost, mf), po))
    p = capi.match_params(w, h, 1, D, frame_diff_sign=sign)
    d_cost, keys = gpu_ctx.upload(cost), gpu_ctx.alloc(w * h * 8)
    disp, best = gpu_ctx.alloc(w * h * 4), gpu_ctx.alloc(w * h * 4)
    # the library's tile and grid, then the launch shapes it can be told to take: one group of workgroups for all
    # level tiles / one per tile (merged by keys_min_kernel), 128 staged columns, short segments
    for opts in ({}, {"iter_groups": 1}, {"iter_groups": 16}, {"multi_txt": 128}, {"multi_txt": 128, "iter_groups": 2, "tseg": 16}):
        with gpu_ctx.options(**opts):
            gpu_ctx.min_pad_wta_keys(d_cost, w * D, p, mf, 0, D, keys)
        gpu_ctx.keys_decode(keys, w, h, disp, best)
        assert np.array_equal(gpu_ctx.download(disp, (h, w), np.int32), wd), opts
        util.assert_bit_equal(gpu_ctx.download(best, (h, w), np.float32), wb, "best cost %r" % (opts,))


def test_exp2_pipeline_fused_equals_staged(gpu_ctx, oracle):
    """The exp2 configuration (box 9x9 + 9x9 min filter + WTA) at D = 64 through the matcher:
    fused route (smhip_fused_box_dsi + smhip_min_pad_wta_keys) == staged route == oracle."""
    w, h, D = 200, 120, 64
    left, right, _ = synth.scene_pair(w, h, D)
    kw = dict(disp_min=0, disp_max=D - 1, match_fn=2, match_max=20, aggr_window_size=9, aggr_minfilter=9)
    po = oracle.make_params(w, h, 1, **kw)
    odisp, obest, odepth, _ = oracle.compute_correspondence(left, right, po)
    for fused in (True, False):
        m = matcher.StereoMatcher(ctx=gpu_ctx, **kw)
        m.allow_fused = fused
        disp, best, depth = m.ComputeCorrespondence(left, right)
        assert m.m_keys_done == fused and m.m_raw_deferred == fused
        assert np.array_equal(disp, odisp), fused
        util.assert_bit_equal(best, obest, "best cost")
        assert np.array_equal(depth, odepth)


def test_general_shape_narrow_kernels(gpu_ctx, oracle):
    """The general-shape fused kernels (narrow slabs of 4-band images whose rows are not 4-byte
    aligned) stay bit-identical to the oracle; the `narrow` option forces them on an aligned input."""
    for w, h, b, D, win in ((61, 23, 1, 16, 9), (40, 17, 4, 7, 5), (33, 12, 1, 32, 4)):
        left, right = synth.noise_pair(w, h, b, seed=31 + D)
        kw = dict(disp_max=D - 1, match_fn=1, match_max=3, match_interval=1, aggr_window_size=win)
        odisp, obest, _, _ = oracle_run(oracle, left, right, **kw)
        with gpu_ctx.options(narrow=1):
            m = matcher.StereoMatcher(ctx=gpu_ctx, **kw)
            disp, best, _ = m.ComputeCorrespondence(left, right)
        assert m.used_fused
        assert np.array_equal(disp, odisp)
        util.assert_bit_equal(best, obest, "best cost")


@pytest.mark.parametrize("w,h,b,D,win,sign", [
    (41, 30, 1, 128, 9, 1), (40, 64, 1, 100, 5, -1), (7, 12, 4, 65, 4, 1), (130, 35, 1, 96, 11, 1),
    (33, 9, 1, 128, 3, -1), (24, 131, 1, 128, 9, 1), (19, 200, 1, 66, 7, -1), (50, 65, 4, 127, 11, 1),
    (30, 5, 1, 80, 5, 1), (30, 10, 1, 128, 9, 1), (12, 17, 1, 128, 11, -1), (9, 449, 1, 90, 3, 1),
    (21, 33, 1, 128, 7, 1), (16, 97, 1, 70, 5, 1), (7, 11, 1, 128, 5, 1), (301, 55, 1, 128, 9, -1),
    (64, 28, 1, 97, 11, 1), (5, 5, 1, 65, 5, -1), (11, 27, 1, 128, 9, 1), (10, 54, 1, 128, 9, 1), (9, 9, 1, 128, 9, 1),
])
def test_slabs_of_65_to_128_levels(gpu_ctx, oracle, w, h, b, D, win, sign):
    """65..128 levels (config 5's per-GPU share is 128): pass B with two levels per lane; heights around the
    key groups of 63 rows and the LDS key buffer (buffered, one group, unbuffered), odd and even heights;
    match_max 4 makes ties frequent, so the lowest-level rule decides."""
    left, right = synth.noise_pair(w, h, b, seed=77 + D)
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    kw = dict(disp_max=D - 1, frame_ref=fr, frame_match=fm, match_fn=1, match_max=4, match_interval=1,
              aggr_window_size=win)
    odisp, obest, _, _ = oracle_run(oracle, left, right, **kw)
    # columns in pairs (one band, windows 3..11; forced: the library pairs from 3584 columns on) / one column per
    # wave with the minimum over the lanes per group of rows through LDS (the library's choice up to 2048 columns)
    # / ... per row with the DPP chain
    for paired, lred in ((1, -1), (0, -1), (0, 0)):
        for keybuf in (-1, 0, 1):
            with gpu_ctx.options(keybuf=keybuf, paired=paired, b_lred=lred):
                m = matcher.StereoMatcher(ctx=gpu_ctx, **kw)
                disp, best, _ = m.ComputeCorrespondence(left, right)
            assert m.used_fused
            assert np.array_equal(disp, odisp), (paired, lred, keybuf)
            util.assert_bit_equal(best, obest, "best cost (paired %d, lred %d, keybuf %d)" % (paired, lred, keybuf))


@pytest.mark.parametrize("win", [3, 5, 9, 13, 15, 17])
@pytest.mark.parametrize("D", [40, 100, 200, 300])
def test_pass_b_blocks_of_other_windows(gpu_ctx, oracle, win, D):
    """Pass B's static windows whose blocks are not one window long (3 and 5: blocks of 9 and 10 rows, the leaving
    row comes from this block or the one before) and the wide ones (13..17: two waves per SIMD): heights of one
    window, one block more or less, several key groups; 1, 2 and 4 levels per lane and two chunks."""
    blk = {3: 9, 5: 10}.get(win, win)
    for h in (win, win + 1, blk - 1, blk + 1, 2 * blk, 3 * blk + 2, 131):
        if h < win:
            continue
        w = 24 + (h % 5)
        left, right = synth.noise_pair(w, h, 1, seed=win * 100 + h)
        kw = dict(disp_max=D - 1, match_fn=1, match_max=6, match_interval=1, aggr_window_size=win)
        odisp, obest, _, _ = oracle_run(oracle, left, right, **kw)
        # (ahead: levels <= 128 keep four / one / two blocks of rows in flight; lred: their wave minimum per group of
        # rows through LDS -- the default on frames this narrow, windows up to 11 -- or per row with DPP)
        for keybuf, ahead, lred in ((-1, -1, -1), (0, -1, -1), (-1, 1, -1), (-1, 2, -1), (-1, -1, 0), (0, 2, 1), (2, 1, 1)):
            with gpu_ctx.options(keybuf=keybuf, b_ahead=ahead, b_lred=lred):
                m = matcher.StereoMatcher(ctx=gpu_ctx, **kw)
                disp, best, _ = m.ComputeCorrespondence(left, right)
            assert m.used_fused
            assert np.array_equal(disp, odisp), (h, keybuf, ahead, lred)
            util.assert_bit_equal(best, obest, "best cost (h %d, keybuf %d, ahead %d, lred %d)" % (h, keybuf, ahead, lred))


@pytest.mark.parametrize("w,h,D,win", [(70, 40, 256, 9), (41, 33, 128, 9), (300, 21, 500, 7), (64, 140, 200, 5)])
def test_occupancy_options(gpu_ctx, oracle, w, h, D, win):
    """The bounds on resident workgroups (LDS floors: `a_lds_kb` for the float-lane pass A, `b_waves` for pass B;
    the library's own policy picks among them by shape) change the launch shape only, never a result."""
    left, right = synth.noise_pair(w, h, 1, seed=5 + D)
    kw = dict(disp_max=D - 1, match_fn=1, match_max=5, match_interval=1, aggr_window_size=win)
    odisp, obest, _, _ = oracle_run(oracle, left, right, **kw)
    for opts in (dict(a_lds_kb=0, b_waves=0), dict(a_lds_kb=54, b_waves=8), dict(a_lds_kb=26, b_waves=6),
                 dict(a_lds_kb=80, b_waves=16), dict(b_waves=10, keybuf=2), dict(a_lds_kb=41, paired=1),
                 dict(b_ahead=1), dict(b_ahead=2, paired=0), dict(b_lred=0), dict(b_lred=1, b_waves=8)):
        with gpu_ctx.options(**opts):
            m = matcher.StereoMatcher(ctx=gpu_ctx, **kw)
            disp, best, _ = m.ComputeCorrespondence(left, right)
        assert m.used_fused
        assert np.array_equal(disp, odisp), opts
        util.assert_bit_equal(best, obest, "best cost (%r)" % (opts,))
    for name, bad in (("a_lds_kb", 81), ("b_waves", 4), ("b_waves", 18)):
        with pytest.raises(Exception, match="LDS floor|waves per CU"):
            gpu_ctx.set_option(name, bad)


@pytest.mark.parametrize("w,h,b,D,win,sign,step,kw", [
    (1100, 20, 1, 64, 9, 1, 1, dict(match_fn=1, match_interval=1)),           # three chunks
    (1000, 20, 4, 33, 9, 1, 1, dict(match_fn=2, match_max=20)),
    (700, 20, 4, 128, 5, -1, 1, dict(match_fn=1, match_interval=1, match_max=4)),
    (600, 20, 1, 300, 17, 1, 1, dict(match_fn=1)),                             # two level chunks of 256
    (530, 20, 1, 40, 19, 1, 1, dict(match_fn=2)),                              # generic window
    (481, 20, 1, 16, 3, -1, 1, dict(match_fn=1, match_max=2)),                 # chunk boundary at the last column
    (482, 20, 1, 16, 3, 1, 1, dict(match_fn=1, match_max=2)),
    (960, 20, 4, 60, 7, 1, 2, dict(match_fn=1, match_interval=1)),             # disp_step 2
    (90, 20, 4, 24, 9, 1, 1, dict(match_fn=1, match_interval=1)),              # a row shorter than one chunk
    (3000, 19, 4, 64, 9, 1, 1, dict(match_fn=1, match_interval=1)),            # chosen without the knob (48 KB row)
])
def test_chunked_row_staging(gpu_ctx, oracle, w, h, b, D, win, sign, step, kw):
    """Pass A with the row staged in chunks (smhip_fused_chunk.h: rows above 40 KB, e.g. 4-band 8K
    frames; forced here through the `chunk` option) equals the oracle, and so do the whole-row kernel
    and the library's own choice (the float-lane kernel where it applies)."""
    left, right = synth.noise_pair(w, h, b, seed=400 + w)
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    seq = dict(disp_max=(D - 1) * step, disp_step=float(step), frame_ref=fr, frame_match=fm, aggr_window_size=win)
    odisp, obest, _, _ = oracle_run(oracle, left, right, **seq, **kw)
    for force in (1, 0, -1):
        with gpu_ctx.options(chunk=force):
            m = matcher.StereoMatcher(ctx=gpu_ctx, **seq, **kw)
            disp, best, _ = m.ComputeCorrespondence(left, right)
        assert m.used_fused
        assert np.array_equal(disp, odisp), force
        util.assert_bit_equal(best, obest, "best cost (chunk=%s)" % force)


@pytest.mark.parametrize("d_first", [0, 1024, 1664, 1792])
def test_bench_geometry_of_eight_ranks(gpu_ctx, oracle, d_first):
    """The slabs `bench.py --gpus 8` hands out (1920 wide, D = 2048, 256 levels per rank) at reduced
    height: the last ranks' disparities reach beyond the image width, where every cost is
    match_outside.  Keys of one slab against the oracle run on that slab as a DSI of its own."""
    from stereomatch_amd import sharding
    w, h, D, n = 1920, 12, 2048, 256
    left, right, _ = synth.scene_pair(w, h, 256)
    p = capi.match_params(w, h, 1, D, match_fn=1, match_interval=1, match_max=1000)
    d_ref, d_mtc = gpu_ctx.upload(left), gpu_ctx.upload(right)
    keys = gpu_ctx.alloc(w * h * 8)
    gpu_ctx.fused_box_wta_keys(d_ref, d_mtc, w, p, 9, d_first, n, keys)
    d, c = sharding.decode_keys(gpu_ctx.download(keys, (w, h), np.int64))
    odisp, obest, _, _ = oracle_run(oracle, left, right, disp_min=d_first, disp_max=d_first + n - 1,
                                    aggr_window_size=9, match_fn=1, match_interval=1, match_max=1000)
    assert np.array_equal(d.T, odisp + d_first)
    util.assert_bit_equal(np.ascontiguousarray(c.T), obest, "best cost of slab %d" % d_first)


@pytest.mark.parametrize("w,h,D,win", [(96, 40, 16, 9), (1, 9, 4, 1), (2, 11, 8, 1), (130, 20, 64, 5), (300, 12, 256, 9)])
def test_bt_interval_in_kernel_and_as_prepass(gpu_ctx, oracle, w, h, D, win):
    """One-band frames: pass A forms the matching row's Birchfield-Tomasi interval while staging the row
    (first and last pixel replicate, StcRawCosts.cpp:107-128); the `bt_prepass` option keeps the separate
    pre-pass kernel that every other shape uses.  Both equal the oracle, with the float-lane and with
    the integer kernels."""
    left, right = synth.noise_pair(w, h, 1, seed=w + D)
    kw = dict(disp_max=D - 1, aggr_window_size=win, match_fn=1, match_interval=1, match_max=1000)
    odisp, obest, _, _ = oracle_run(oracle, left, right, **kw)
    for prepass in (0, 1):
        for float_a in (1, 0):
            with gpu_ctx.options(bt_prepass=prepass, float_a=float_a):
                m = matcher.StereoMatcher(ctx=gpu_ctx, **kw)
                disp, best, _ = m.ComputeCorrespondence(left, right)
            assert m.used_fused
            util.assert_bit_equal(disp, odisp, "disparity prepass=%s float_a=%s" % (prepass, float_a))
            util.assert_bit_equal(best, obest, "best cost prepass=%s float_a=%s" % (prepass, float_a))
