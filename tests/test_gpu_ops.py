"""Parity of every stage operator of the C-ABI against the oracle, bit for bit."""
import numpy as np
import pytest

import util
from stereomatch_amd import capi, synth

pytestmark = pytest.mark.gpu


def up(ctx, a):
    return ctx.upload(np.ascontiguousarray(a))


def run_inplace(ctx, cost, fn, row_pad=0):
    """Uploads cost [H,W,D] with `row_pad` extra floats per row, runs fn(buf, w, h, d, row_elems)."""
    h, w, d = cost.shape
    re = w * d + row_pad
    host = np.full((h, re), np.float32(-12345.0))
    host[:, :w * d] = cost.reshape(h, w * d)
    buf = up(ctx, host)
    fn(buf, w, h, d, re)
    out = ctx.download(buf, (h, re), np.float32)
    if row_pad:
        assert np.all(out[:, w * d:] == np.float32(-12345.0)), "row padding was overwritten"
    return out[:, :w * d].reshape(h, w, d)


RAW_CASES = [
    # w, h, b, D, dmin, sign, fn, mmax, interval, interpolated
    (37, 11, 1, 9, 0, 1, 1, 1000, 0, 0),
    (37, 11, 4, 9, 0, 1, 2, 1000, 0, 0),
    (64, 9, 4, 16, 0, 1, 1, 20, 1, 0),
    (50, 7, 1, 33, 2, 1, 2, 5, 1, 0),
    (41, 8, 1, 12, 0, -1, 1, 1000, 1, 0),
    (41, 8, 4, 12, 1, -1, 2, 50, 0, 0),
    (45, 6, 4, 8, 0, 1, 1, 1000, 1, 1),
    (130, 5, 1, 100, 0, 1, 1, 2, 1, 0),
    (300, 4, 1, 260, 0, 1, 1, 1000, 1, 0),
    (90, 3, 4, 1028, 3, -1, 2, 1000, 1, 1),
    (70, 5, 1, 20, -7, 1, 1, 9, 0, 0),
]


@pytest.mark.parametrize("w,h,b,D,dmin,sign,fn,mmax,interval,interp", RAW_CASES)
def test_raw_costs(gpu_ctx, oracle, w, h, b, D, dmin, sign, fn, mmax, interval, interp):
    ref, mtc = synth.noise_pair(w, h, b, seed=synth.SEED_NOISE + D)
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    po = oracle.make_params(w, h, b, disp_min=dmin, disp_max=dmin + D - 1, frame_ref=fr,
                            frame_match=fm, match_fn=fn, match_max=mmax,
                            match_interval=interval, match_interpolated=interp)
    want = oracle.raw_costs(ref, mtc, po)
    p = capi.match_params(w, h, b, D, disp_min=dmin, frame_diff_sign=sign, match_fn=fn,
                          match_max=mmax, match_interval=interval, match_interpolated=interp)
    assert gpu_ctx.match_outside(p) == oracle.match_outside(po)
    d_ref, d_mtc = up(gpu_ctx, ref), up(gpu_ctx, mtc)
    cost = gpu_ctx.alloc(w * h * D * 4)
    gpu_ctx.raw_costs(d_ref, d_mtc, w * b, p, cost, w * D)
    util.assert_bit_equal(gpu_ctx.download(cost, (h, w, D), np.float32), want, "raw costs")
    # PadCosts on the same volume
    gpu_ctx.pad_costs(cost, w * D, p)
    util.assert_bit_equal(gpu_ctx.download(cost, (h, w, D), np.float32),
                          oracle.pad_costs(want, po), "pad costs")


SUBPIX_CASES = [
    # w, h, b, dmin, dmax, den, sign, fn, mmax, interp(1 linear / 3 cubic), interval, interpolated
    (37, 7, 1, 0, 4, 2, 1, 1, 1000, 3, 0, 0),
    (37, 7, 1, 0, 4, 2, 1, 2, 1000, 1, 0, 0),
    (40, 6, 4, 0, 3, 4, 1, 1, 20, 3, 1, 0),
    (40, 6, 4, 1, 4, 3, -1, 2, 30, 1, 1, 1),
    (33, 5, 1, 0, 5, 4, 1, 1, 1000, 3, 1, 1),
    (33, 5, 1, 0, 5, 2, -1, 1, 1000, 3, 0, 1),
    (29, 4, 4, 0, 2, 5, 1, 2, 1000, 3, 0, 1),
    (64, 3, 1, 2, 9, 8, 1, 1, 7, 1, 1, 1),
]


@pytest.mark.parametrize("w,h,b,dmin,dmax,den,sign,fn,mmax,minterp,interval,interp", SUBPIX_CASES)
def test_raw_costs_subpixel(gpu_ctx, oracle, w, h, b, dmin, dmax, den, sign, fn, mmax, minterp,
                            interval, interp):
    """disp_step = 1/den: InterpolateLine (linear / cubic), BT on the up-sampled lines and the
    interpolated-cost average (StcRawCosts.cpp:67-218), then PadCosts with den > 1.  The
    reference holds no golden output for disp_step < 1: parity is against the oracle only
    (parity unpinned)."""
    ref, mtc = synth.noise_pair(w, h, b, seed=synth.SEED_NOISE + den)
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    po = oracle.make_params(w, h, b, disp_min=dmin, disp_max=dmax, disp_step=1.0 / den,
                            frame_ref=fr, frame_match=fm, match_fn=fn, match_max=mmax,
                            match_interp=minterp, match_interval=interval,
                            match_interpolated=interp)
    D = po.disp_n
    assert po.disp_den == den and D == den * (dmax - dmin) + 1
    want = oracle.raw_costs(ref, mtc, po)
    p = capi.match_params(w, h, b, D, disp_min=dmin, disp_den=den, frame_diff_sign=sign,
                          match_fn=fn, match_interp=minterp, match_max=mmax,
                          match_interval=interval, match_interpolated=interp)
    cost = gpu_ctx.alloc(w * h * D * 4)
    gpu_ctx.raw_costs(up(gpu_ctx, ref), up(gpu_ctx, mtc), w * b, p, cost, w * D)
    util.assert_bit_equal(gpu_ctx.download(cost, (h, w, D), np.float32), want, "sub-pixel raw costs")
    gpu_ctx.pad_costs(cost, w * D, p)
    util.assert_bit_equal(gpu_ctx.download(cost, (h, w, D), np.float32),
                          oracle.pad_costs(want, po), "sub-pixel pad costs")


def test_raw_costs_pitched_images(gpu_ctx, oracle):
    """Images whose rows are padded to 8 bytes (Image.cpp:97-98): W*b = 37 -> pitch 40."""
    w, h, b, D = 37, 6, 1, 8
    ref, mtc = synth.noise_pair(w, h, b)
    pitch = (w * b + 7) & ~7
    def padded(img):
        out = np.full((h, pitch), 255, np.uint8)
        out[:, :w * b] = img.reshape(h, w * b)
        return out
    po = oracle.make_params(w, h, b, disp_max=D - 1, match_interval=1)
    p = capi.match_params(w, h, b, D, match_interval=1)
    cost = gpu_ctx.alloc(w * h * D * 4)
    gpu_ctx.raw_costs(up(gpu_ctx, padded(ref)), up(gpu_ctx, padded(mtc)), pitch, p, cost, w * D)
    util.assert_bit_equal(gpu_ctx.download(cost, (h, w, D), np.float32),
                          oracle.raw_costs(ref, mtc, po), "pitched raw costs")


SHAPES = [(13, 17, 5, 0), (20, 33, 16, 0), (9, 21, 3, 1), (24, 40, 64, 0), (12, 19, 100, 0),
          (10, 15, 7, 1), (31, 29, 8, 4), (27, 30, 12, 2), (13, 13, 4, 0), (1, 1, 3, 0)]


@pytest.mark.parametrize("h,w,d,pad", SHAPES)
@pytest.mark.parametrize("win", [3, 4, 5, 9, 11, 13])
def test_box_filter(gpu_ctx, oracle, h, w, d, pad, win):
    c = util.rand_dsi(h, w, d, 10 + win)
    if win > min(h, w):
        # outside the reference's domain: its running sum reads src[x + pr] unchecked up to index
        # win - 1 (BoxFilter.cpp:114-124) -- the library refuses instead of inventing a border rule
        with pytest.raises(capi.SmhipError, match="too large"):
            run_inplace(gpu_ctx, c, lambda b, W, H, D, re: gpu_ctx.box_filter(b, W, H, D, re, win), pad)
        return
    got = run_inplace(gpu_ctx, c, lambda b, W, H, D, re: gpu_ctx.box_filter(b, W, H, D, re, win), pad)
    util.assert_bit_equal(got, oracle.box_filter(c, win), "box %d" % win)


def test_box_filter_large_magnitude_drift(gpu_ctx, oracle):
    """SSD magnitudes (outside = 260100) next to small costs: the running sum's rounding
    drift must be reproduced exactly (hard parts)."""
    c = util.rand_dsi(16, 48, 16, 3)
    c[:, :5, :] = 260100.0
    got = run_inplace(gpu_ctx, c, lambda b, W, H, D, re: gpu_ctx.box_filter(b, W, H, D, re, 9))
    util.assert_bit_equal(got, oracle.box_filter(c, 9), "box drift")


def test_box_filter_sum_mode(gpu_ctx, oracle):
    c = util.rand_dsi(11, 14, 6, 4)
    got = run_inplace(gpu_ctx, c,
                      lambda b, W, H, D, re: gpu_ctx.box_filter(b, W, H, D, re, 5, average=False))
    util.assert_bit_equal(got, oracle.box_filter(c, 5, average=False), "box sum")


@pytest.mark.parametrize("h,w,d,pad", SHAPES)
@pytest.mark.parametrize("win", [2, 3, 7, 9])
def test_min_filter(gpu_ctx, oracle, h, w, d, pad, win):
    c = util.rand_dsi(h, w, d, 20 + win, "float")
    got = run_inplace(gpu_ctx, c, lambda b, W, H, D, re: gpu_ctx.min_filter(b, W, H, D, re, win), pad)
    util.assert_bit_equal(got, oracle.min_filter(c, win), "min %d" % win)


@pytest.mark.parametrize("h,w,d,pad", SHAPES)
def test_binomial(gpu_ctx, oracle, h, w, d, pad):
    c = util.rand_dsi(h, w, d, 30, "float")
    got = run_inplace(gpu_ctx, c, lambda b, W, H, D, re: gpu_ctx.convolve_14641(b, W, H, D, re), pad)
    util.assert_bit_equal(got, oracle.convolve_14641(c), "binomial")
    # iterate, as aggr_iter does (StcAggregate.cpp:265)
    want = c
    for _ in range(3):
        want = oracle.convolve_14641(want)
    def three(b, W, H, D, re):
        for _ in range(3):
            gpu_ctx.convolve_14641(b, W, H, D, re)
    util.assert_bit_equal(run_inplace(gpu_ctx, c, three, pad), want, "binomial x3")


@pytest.mark.parametrize("h,w,d,pad", SHAPES)
@pytest.mark.parametrize("beta", [0.0, 0.5])
def test_diffusion(gpu_ctx, oracle, h, w, d, pad, beta):
    c = util.rand_dsi(h, w, d, 40, "float")
    c0 = util.rand_dsi(h, w, d, 41, "float")
    re = w * d + pad
    def padded(a):
        out = np.zeros((h, re), np.float32)
        out[:, :w * d] = a.reshape(h, w * d)
        return out
    src, dst, b0 = up(gpu_ctx, padded(c)), gpu_ctx.alloc(h * re * 4), up(gpu_ctx, padded(c0))
    gpu_ctx.diffusion_iter(src, dst, b0 if beta > 0 else None, w, h, d, re, 0.15, beta)
    got = gpu_ctx.download(dst, (h, re), np.float32)[:, :w * d].reshape(h, w, d)
    util.assert_bit_equal(got, oracle.diffusion_iter(c, 0.15, beta, c0), "diffusion")


@pytest.mark.parametrize("h,w,d", [(6, 20, 7), (5, 9, 64), (4, 11, 100), (3, 7, 256), (3, 5, 300),
                                   (3, 37, 128), (5, 19, 512), (3, 9, 1024), (3, 9, 640), (3, 5, 1100),
                                   (4, 7, 132), (4, 70, 8), (5, 130, 16), (4, 33, 20), (4, 41, 32),
                                   (4, 29, 40), (4, 70, 60), (4, 23, 96), (4, 19, 124), (4, 50, 4)])
def test_wta(gpu_ctx, oracle, h, w, d):
    c = util.rand_dsi(h, w, d, 50)
    c[0, 0, :] = 5.0
    c[1, 3, 2] = c[1, 3, d - 1] = -1.0
    c[2, 1, d - 1] = -0.0
    c[2, 1, :d - 1] = 0.0
    wd, wb = oracle.wta(c)
    disp, best = gpu_ctx.alloc(h * w * 4), gpu_ctx.alloc(h * w * 4)
    gpu_ctx.wta(up(gpu_ctx, c), w, h, d, w * d, disp, best)
    assert np.array_equal(gpu_ctx.download(disp, (h, w), np.int32), wd)
    assert np.array_equal(gpu_ctx.download(best, (h, w), np.float32), wb)


def test_error_reporting(gpu_ctx):
    """Host misuse comes back as SMHIP_EINVAL with the reference's message (BoxFilter.cpp:179-180)."""
    buf = gpu_ctx.alloc(64 * 64 * 4 * 4)
    with pytest.raises(capi.SmhipError) as e:
        gpu_ctx.box_filter(buf, 64, 64, 4, 256, 5, 7)
    assert e.value.code == -1 and "xWidth != yWidth" in str(e.value)
    with pytest.raises(capi.SmhipError):
        gpu_ctx.box_filter(buf, 64, 64, 4, 100, 5)            # row_elems < w*d
    p = capi.match_params(64, 64, 3, 8)
    with pytest.raises(capi.SmhipError) as e:
        gpu_ctx.raw_costs(buf, buf, 192, p, buf, 512)
    assert "number of bands (3) not supported" in str(e.value)    # StereoIO.cpp:76-81
    p = capi.match_params(64, 64, 1, 1)
    with pytest.raises(capi.SmhipError) as e:
        gpu_ctx.raw_costs(buf, buf, 64, p, buf, 64)
    assert "too few disparity levels" in str(e.value)             # StereoMatcher.cpp:158-159
    p = capi.match_params(64, 64, 1, 8, disp_den=2)
    with pytest.raises(capi.SmhipError) as e:                     # fused path: integer steps only
        gpu_ctx.fused_box_wta(buf, buf, 64, p, 5, buf, buf)
    assert "disp_step < 1" in str(e.value)
    with pytest.raises(capi.SmhipError, match="rows"):            # SMHIP_MAX_ROWS: checked, nothing launched
        gpu_ctx.wta(buf, 1, 70000, 4, 4, buf, buf)
    p = capi.match_params(8, 70000, 1, 8)
    with pytest.raises(capi.SmhipError, match="rows"):
        gpu_ctx.raw_costs(buf, buf, 8, p, buf, 64)
    p = capi.match_params(16, 16, 1, 8)                           # PadCosts + WTA -> keys on a slab
    with pytest.raises(capi.SmhipError, match="outside"):
        gpu_ctx.pad_wta_keys(buf, 16 * 4, p, 6, 4, buf)           # levels [6, 10) of 8
    with pytest.raises(capi.SmhipError, match="outside"):
        gpu_ctx.pad_wta_keys(buf, 16 * 4, p, -1, 4, buf)
    with pytest.raises(capi.SmhipError, match="row_elems"):
        gpu_ctx.pad_wta_keys(buf, 16 * 4 - 1, p, 0, 4, buf)
    with pytest.raises(capi.SmhipError, match="disp_step < 1"):
        gpu_ctx.pad_wta_keys(buf, 16 * 4, capi.match_params(16, 16, 1, 8, disp_den=2), 0, 4, buf)


@pytest.mark.parametrize("h,w,d,pad", SHAPES + [(140, 30, 8, 0), (70, 21, 12, 4)])
def test_out_of_place_filters(gpu_ctx, oracle, h, w, d, pad):
    """smhip_min_filter_to / smhip_convolve_14641_to: one pass src -> dst (column-walk kernels);
    tall shapes cross the row-segment boundaries of those kernels."""
    c = util.rand_dsi(h, w, d, 60, "float")
    re = w * d + pad
    host = np.zeros((h, re), np.float32)
    host[:, :w * d] = c.reshape(h, w * d)
    src, dst = up(gpu_ctx, host), gpu_ctx.alloc(h * re * 4)
    def get():
        return gpu_ctx.download(dst, (h, re), np.float32)[:, :w * d].reshape(h, w, d)
    for win in (9, 5, 3, 4, 7, 11):
        if win - win // 2 >= min(w, h):
            continue
        gpu_ctx.min_filter_to(src, dst, w, h, d, re, win)
        util.assert_bit_equal(get(), oracle.min_filter(c, win), "min_to %d" % win)
    gpu_ctx.convolve_14641_to(src, dst, w, h, d, re)
    util.assert_bit_equal(get(), oracle.convolve_14641(c), "binomial_to")
    # the source is untouched
    util.assert_bit_equal(gpu_ctx.download(src, (h, re), np.float32), host, "source")
    with pytest.raises(capi.SmhipError):
        gpu_ctx.convolve_14641_to(src, src, w, h, d, re)


@pytest.mark.parametrize("h,w,d", [(5, 9, 3), (6, 17, 16), (4, 11, 31), (3, 7, 100)])
def test_subpixel_fit_collapse_refine(gpu_ctx, oracle, h, w, d):
    """AggrSubPixelFit / AggrCollapse / Refine (StcAggregate.cpp:66-230, StcRefine.cpp:28-100)
    against the oracle on random volumes with invalid levels, plateaus and ties.  The reference
    holds no output for these stages: parity unpinned."""
    outside = 255.0
    rng = np.random.RandomState(70 + d)
    c = rng.randint(0, 12, size=(h, w, d)).astype(np.float32)     # many ties and plateaus
    f = (rng.rand(h, w, d) * 30.0).astype(np.float32)
    c = np.where(rng.rand(h, w, d) < 0.5, f, c).astype(np.float32)
    c[rng.rand(h, w, d) < 0.05] = outside
    wc, wm, we = oracle.subpixel_fit(c, outside)
    n = h * w * d * 4
    src, dst, dmin, dcert = up(gpu_ctx, c), gpu_ctx.alloc(n), gpu_ctx.alloc(n), gpu_ctx.alloc(n)
    gpu_ctx.subpixel_fit(src, dst, dmin, dcert, w, h, d, w * d, outside)
    util.assert_bit_equal(gpu_ctx.download(dst, (h, w, d), np.float32), wc, "fit cost")
    util.assert_bit_equal(gpu_ctx.download(dmin, (h, w, d), np.float32), wm, "fit min")
    util.assert_bit_equal(gpu_ctx.download(dcert, (h, w, d), np.float32), we, "fit cert")
    # collapse, with and without the fitted offsets
    for df in (2, 3, 5):
        if (d - 1) % df:
            continue
        d2 = (d - 1) // df + 1
        step = float(np.float32(1.0) / np.float32(df))
        for mind, dm in ((wm, dmin), (None, None)):
            oc2, om2 = oracle.collapse(wc, mind, df, step)
            c2, m2 = gpu_ctx.alloc(h * w * d2 * 4), gpu_ctx.alloc(h * w * d2 * 4)
            gpu_ctx.collapse(dst, dm, w, h, d, w * d, df, d2, w * d2, step, c2, m2)
            util.assert_bit_equal(gpu_ctx.download(c2, (h, w, d2), np.float32), oc2, "collapse cost")
            util.assert_bit_equal(gpu_ctx.download(m2, (h, w, d2), np.float32), om2, "collapse min")
    # refine in the three modes, winners from the oracle's WTA
    disp, _ = oracle.wta(wc)
    ddisp, fd = up(gpu_ctx, disp), gpu_ctx.alloc(h * w * 4)
    for mode in (0, 1, 2):
        for step, dmn in ((1.0, 0), (0.5, 3)):
            gpu_ctx.refine(ddisp, dst, dmin, w, h, d, w * d, step, float(dmn), mode, fd)
            util.assert_bit_equal(gpu_ctx.download(fd, (h, w), np.float32),
                                  oracle.refine(wc, wm, disp, step, dmn, mode), "refine %d" % mode)
    with pytest.raises(capi.SmhipError):
        gpu_ctx.collapse(dst, dmin, w, h, d, w * d, 2, d, w * d, 0.5, dst, dmin)
    # ComputeMatchQuality (StcEvaluate.cpp:760-833) on the fitted volume: parabola at the refined winner,
    # no interpolation (eval_match_quality 2), the aggr_subpixel route, 1 / 3 bands, with and without sqrt
    fcost, fcert = gpu_ctx.alloc(h * w * 4), gpu_ctx.alloc(h * w * 4)
    for step, dmn in ((1.0, 0), (0.5, 3)):
        fdisp = oracle.refine(wc, wm, disp, step, dmn, 2)
        fd = up(gpu_ctx, fdisp)
        inv = float(np.float32(1.0) / np.float32(step))
        for qmode, use_fit, nbands, sq in ((1, 0, 1, 0), (2, 0, 3, 1), (1, 1, 3, 0), (1, 0, 3, 1)):
            gpu_ctx.match_quality(dst, dcert if use_fit else None, fd, w, h, d, w * d, w * d, float(dmn), inv,
                                  qmode, use_fit, nbands, sq, fcost, fcert)
            oc, oe = oracle.match_quality(wc, we if use_fit else None, fdisp, dmn, inv, qmode, use_fit, nbands, sq)
            util.assert_bit_equal(gpu_ctx.download(fcost, (h, w), np.float32), oc, "final cost %r" % ((qmode, use_fit),))
            util.assert_bit_equal(gpu_ctx.download(fcert, (h, w), np.float32), oe, "certainty %r" % ((qmode, use_fit),))
    with pytest.raises(capi.SmhipError, match="certainty volume"):
        gpu_ctx.match_quality(dst, None, fd, w, h, d, w * d, w * d, 0.0, 1.0, 1, 1, 1, 0, fcost, fcert)


@pytest.mark.parametrize("h,w,d,pad", [(9, 21, 8, 0), (150, 70, 12, 0), (70, 150, 40, 4), (3, 5, 4, 0),
                                       (40, 130, 7, 1)])
@pytest.mark.parametrize("n_iter", [1, 2, 3, 4, 5, 7, 10])
@pytest.mark.parametrize("maxk", ["", "4", "3"])
def test_multi_iteration_passes(gpu_ctx, oracle, h, w, d, pad, n_iter, maxk, request):
    """smhip_diffusion_run / smhip_convolve_14641_run: up to four iterations per pass (temporal
    blocking across tiles of 64 pixels and row segments) == n_iter single iterations of the
    oracle, bit for bit; unaligned shapes take the one-iteration-per-pass route.  The depth per pass
    follows the shape (small volumes: 1 or 2); the `maxk` option forces the deeper kernels onto these shapes."""
    if maxk:
        gpu_ctx.set_option("maxk", int(maxk))
        request.addfinalizer(lambda: gpu_ctx.set_option("maxk", -1))
    c = util.rand_dsi(h, w, d, 80 + n_iter, "float")
    c0 = util.rand_dsi(h, w, d, 81, "float")
    re = w * d + pad
    def padded(a):
        out = np.zeros((h, re), np.float32)
        out[:, :w * d] = a.reshape(h, w * d)
        return out
    def result(bufs, in_b):
        return gpu_ctx.download(bufs[1] if in_b else bufs[0], (h, re), np.float32)[:, :w * d].reshape(h, w, d)
    b0 = up(gpu_ctx, padded(c0))
    for beta in (0.0, 0.5):
        want = c
        for _ in range(n_iter):
            want = oracle.diffusion_iter(want, 0.15, beta, c0)
        bufs = (up(gpu_ctx, padded(c)), gpu_ctx.alloc(h * re * 4))
        in_b = gpu_ctx.diffusion_run(bufs[0], bufs[1], b0 if beta > 0 else None, w, h, d, re, 0.15, beta, n_iter)
        util.assert_bit_equal(result(bufs, in_b), want, "diffusion x%d beta %g" % (n_iter, beta))
    want = c
    for _ in range(n_iter):
        want = oracle.convolve_14641(want)
    bufs = (up(gpu_ctx, padded(c)), gpu_ctx.alloc(h * re * 4))
    in_b = gpu_ctx.convolve_14641_run(bufs[0], bufs[1], w, h, d, re, n_iter)
    util.assert_bit_equal(result(bufs, in_b), want, "binomial x%d" % n_iter)


@pytest.mark.parametrize("kind", ["tiny", "subnormal", "negative", "late_tiny", "signed_zero", "plain"])
@pytest.mark.parametrize("opts", [{}, {"tap_slow": 1}, {"multi_px": 0, "multi_txt": 128}, {"maxk": 3}, {"multi_px": 0},
                                  {"multi_px": 32}, {"multi_px": 0, "tap_slow": 1}])
def test_binomial_tap_guard(gpu_ctx, oracle, kind, opts, request):
    """The multi-iteration binomial pass takes the 2^-4 and 2^-2 taps as exact-product FMAs (same bits as the
    reference's multiply-then-add, Convolve.cpp:100-107) only while every input of the workgroup is zero or at
    least 2^-80 and carries no sign bit; otherwise it falls back to the ten-operation sum.  Inputs that would
    expose a difference -- values whose products are subnormal (a rounded product loses bits an FMA keeps),
    negative values, -0 -- must therefore still match the oracle bit for bit, wherever in the volume they sit."""
    for k, v in dict({"maxk": 4}, **opts).items():
        gpu_ctx.set_option(k, v)
        request.addfinalizer(lambda k=k: gpu_ctx.set_option(k, -1))
    h, w, d = 90, 152, 40
    rng = np.random.RandomState(7)
    c = (rng.rand(h, w, d) * 1000.0).astype(np.float32)
    c[rng.rand(h, w, d) < 0.3] = 0.0
    if kind == "tiny":              # products around 2^-100 .. 2^-126 after a few iterations
        c *= np.float32(1e-30)
    elif kind == "subnormal":       # the boundary itself: inputs in the subnormal range and just above
        c[:, :w // 2] *= np.float32(1e-36)
        c[:, w // 2:] = (rng.rand(h, w - w // 2, d) * 3e-38).astype(np.float32)
    elif kind == "negative":
        c[40:50, 60:70] -= np.float32(500.0)
    elif kind == "late_tiny":       # one tiny value deep inside a segment, in the halo column of a tile
        c[77, 47, 13] = np.float32(3.1e-33)
        c[78, 48, 14] = np.float32(1.7e-39)
    elif kind == "signed_zero":
        c[rng.rand(h, w, d) < 0.1] = np.float32(-0.0)
    for n_iter in (4, 8):
        want = c
        for _ in range(n_iter):
            want = oracle.convolve_14641(want)
        bufs = (up(gpu_ctx, c), gpu_ctx.alloc(h * w * d * 4))
        in_b = gpu_ctx.convolve_14641_run(bufs[0], bufs[1], w, h, d, w * d, n_iter)
        util.assert_bit_equal(gpu_ctx.download(bufs[1] if in_b else bufs[0], (h, w, d), np.float32), want,
                              "binomial x%d on %s inputs" % (n_iter, kind))


@pytest.mark.parametrize("h,w,d", [(40, 128, 8), (150, 252, 40), (70, 364, 12), (3, 8, 4)])
@pytest.mark.parametrize("n_iter", [1, 3, 6, 7, 10, 20])
def test_diffusion_run_pixels_on_lanes(gpu_ctx, oracle, h, w, d, n_iter):
    """smhip_diffusion_run with the pixels-on-lanes passes (up to six iterations each, horizontal neighbours by
    DPP; the library's choice from 64 MiB on, forced here) == n_iter single iterations of the oracle."""
    c = util.rand_dsi(h, w, d, 60 + n_iter, "float")
    want = c
    for _ in range(n_iter):
        want = oracle.diffusion_iter(want, 0.15, 0.0, None)
    with gpu_ctx.options(multi_px=32):
        bufs = (up(gpu_ctx, c), gpu_ctx.alloc(h * w * d * 4))
        in_b = gpu_ctx.diffusion_run(bufs[0], bufs[1], None, w, h, d, w * d, 0.15, 0.0, n_iter)
    util.assert_bit_equal(gpu_ctx.download(bufs[1] if in_b else bufs[0], (h, w, d), np.float32), want,
                          "diffusion x%d" % n_iter)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SMHIP_FUZZ_OPS_N", "40"))))   # more on demand
def test_random_iterated_operators(gpu_ctx, oracle, seed):
    """The iterated stage operators and the min filter + WTA kernel on random volumes and launch shapes: every form
    the library can be told to take (LDS tiles of 64 / 128 pixels, pixels on lanes with 16 / 32 levels per workgroup,
    short and long row segments, any depth per pass) gives the oracle's bits."""
    from stereomatch_amd import capi, sharding
    r = np.random.RandomState(6100 + seed)
    w = int(r.randint(2, 90)) * 4 if r.rand() < 0.8 else int(r.randint(5, 300))
    h = int(r.choice([int(r.randint(2, 30)), int(r.randint(30, 180))]))
    d = int(r.choice([4, 8, 12, 20, 36, 40, 68]))
    n_iter = int(r.randint(1, 12))
    opts = dict(multi_px=int(r.choice([-1, 0, 16, 32])), multi_txt=int(r.choice([-1, 64, 128])),
                tseg=int(r.choice([-1, -1, 16, 128])), maxk=int(r.choice([-1, -1, 1, 2, 3])))
    kind = "float" if r.rand() < 0.7 else "int"
    c = util.rand_dsi(h, w, d, 700 + seed, kind)
    op = int(r.choice([0, 1, 2]))
    if op == 0:                                        # binomial 14641
        want = c
        for _ in range(n_iter):
            want = oracle.convolve_14641(want)
        with gpu_ctx.options(**opts):
            bufs = (up(gpu_ctx, c), gpu_ctx.alloc(h * w * d * 4))
            in_b = gpu_ctx.convolve_14641_run(bufs[0], bufs[1], w, h, d, w * d, n_iter)
        util.assert_bit_equal(gpu_ctx.download(bufs[1] if in_b else bufs[0], (h, w, d), np.float32), want,
                              "binomial x%d %r" % (n_iter, opts))
    elif op == 1:                                      # diffusion
        lam = float(r.choice([0.15, 0.05, 0.25]))
        want = c
        for _ in range(n_iter):
            want = oracle.diffusion_iter(want, lam, 0.0, None)
        with gpu_ctx.options(**opts):
            bufs = (up(gpu_ctx, c), gpu_ctx.alloc(h * w * d * 4))
            in_b = gpu_ctx.diffusion_run(bufs[0], bufs[1], None, w, h, d, w * d, lam, 0.0, n_iter)
        util.assert_bit_equal(gpu_ctx.download(bufs[1] if in_b else bufs[0], (h, w, d), np.float32), want,
                              "diffusion x%d %r" % (n_iter, opts))
    else:                                              # min filter + PadCosts + WTA keys
        mf = int(r.choice([3, 5, 7, 9, 11]))
        if mf > min(w, h):
            mf = 3
        if min(w, h) < 3:
            return
        sign = int(r.choice([1, -1]))
        fr, fm = (0, 1) if sign > 0 else (1, 0)
        po = oracle.make_params(w, h, 1, disp_max=d - 1, frame_ref=fr, frame_match=fm)
        wd, wb = oracle.wta(oracle.pad_costs(oracle.min_filter(c, mf), po))
        p = capi.match_params(w, h, 1, d, frame_diff_sign=sign)
        keys = gpu_ctx.alloc(w * h * 8)
        with gpu_ctx.options(iter_groups=int(r.choice([-1, 1, 3])), multi_txt=opts["multi_txt"],
                             tseg=int(r.choice([-1, 8, 32, 64]))):
            gpu_ctx.min_pad_wta_keys(up(gpu_ctx, c), w * d, p, mf, 0, d, keys)
        gd, gb = sharding.decode_keys(gpu_ctx.download(keys, (w, h), np.int64).T)
        assert np.array_equal(gd, wd), (seed, mf, opts)
        util.assert_bit_equal(np.ascontiguousarray(gb), wb, "min %d + WTA best cost" % mf)


def test_tile_kernels_with_long_row_segments(gpu_ctx, oracle, request):
    """Large volumes use 128-row segments in the LDS-tiled kernels; force that length on a small
    volume (the `tseg` option) so the segment logic of the production configuration is covered."""
    gpu_ctx.set_option("tseg", 128)
    request.addfinalizer(lambda: gpu_ctx.set_option("tseg", -1))
    h, w, d = 300, 40, 8
    c = util.rand_dsi(h, w, d, 90, "float")
    src, dst = up(gpu_ctx, c), gpu_ctx.alloc(h * w * d * 4)
    gpu_ctx.min_filter_to(src, dst, w, h, d, w * d, 9)
    util.assert_bit_equal(gpu_ctx.download(dst, (h, w, d), np.float32), oracle.min_filter(c, 9), "min 9")
    gpu_ctx.convolve_14641_to(src, dst, w, h, d, w * d)
    util.assert_bit_equal(gpu_ctx.download(dst, (h, w, d), np.float32), oracle.convolve_14641(c), "binomial")
    want = c
    for _ in range(4):
        want = oracle.convolve_14641(want)
    in_b = gpu_ctx.convolve_14641_run(src, dst, w, h, d, w * d, 4)
    util.assert_bit_equal(gpu_ctx.download(dst if in_b else src, (h, w, d), np.float32), want, "binomial x4")
    src = up(gpu_ctx, c)
    want = c
    for _ in range(4):
        want = oracle.diffusion_iter(want, 0.15, 0.0, None)
    in_b = gpu_ctx.diffusion_run(src, dst, None, w, h, d, w * d, 0.15, 0.0, 4)
    util.assert_bit_equal(gpu_ctx.download(dst if in_b else src, (h, w, d), np.float32), want, "diffusion x4")


@pytest.mark.parametrize("w,h,b,iters", [(37, 11, 1, 1), (40, 9, 4, 3), (5, 3, 1, 2)])
def test_preprocess_blur(gpu_ctx, oracle, w, h, b, iters):
    """PreProcess (StcPreProcess.cpp:49-55): uchar 121 x 14641 blur, truncating, replicate border;
    rows padded to 8 bytes keep their padding.  Parity unpinned (oracle only)."""
    img, _ = synth.noise_pair(w, h, b, seed=7 + iters)
    pitch = (w * b + 7) & ~7
    host = np.full((h, pitch), 77, np.uint8)
    host[:, :w * b] = img.reshape(h, w * b)
    d = up(gpu_ctx, host)
    gpu_ctx.preprocess_blur(d, pitch, w, h, b, iters)
    got = gpu_ctx.download(d, (h, pitch), np.uint8)
    assert np.array_equal(got[:, :w * b].reshape(h, w, b), oracle.preprocess_blur(img, iters))
    assert np.all(got[:, w * b:] == 77)


@pytest.mark.parametrize("h,w,d,b", [(5, 40, 7, 1), (4, 33, 16, 4), (3, 50, 100, 1), (3, 20, 200, 1),
                                     (2, 12, 300, 1), (2, 9, 1, 1), (2, 1, 5, 1), (2, 10, 520, 1),
                                     (1, 6, 1024, 1)])
@pytest.mark.parametrize("smooth", [(1.0, 5.0, 1.0), (20.0, 8.0, 2.0), (0.0, 5.0, 1.0)])
def test_opt_so(gpu_ctx, oracle, h, w, d, b, smooth):
    """smhip_opt_so (O(D) per pixel, wave minima) == the reference's O(D^2) scan restated in the
    oracle, including ties (integer-valued costs, zero smoothness) -- parity unpinned."""
    rng = np.random.RandomState(100 + d)
    c = rng.randint(0, 12, size=(h, w, d)).astype(np.float32)
    c = np.where(rng.rand(h, w, d) < 0.3, (rng.rand(h, w, d) * 40).astype(np.float32), c)
    img, _ = synth.noise_pair(w, h, b, seed=200 + d)
    p = oracle.make_params(w, h, b, disp_max=max(d - 1, 1), opt_smoothness=smooth[0],
                           opt_grad_thresh=smooth[1], opt_grad_penalty=smooth[2])
    want = oracle.opt_so(c, img, p)
    disp, best = gpu_ctx.alloc(h * w * 4), gpu_ctx.alloc(h * w * 4)
    gpu_ctx.opt_so(up(gpu_ctx, c), w, h, d, w * d, up(gpu_ctx, img), w * b, b, smooth[0], smooth[1],
                   smooth[2], disp, best)
    got = gpu_ctx.download(disp, (h, w), np.int32)
    assert np.array_equal(got, want)
    gb = gpu_ctx.download(best, (h, w), np.float32)
    assert np.array_equal(gb, np.take_along_axis(c, want[..., None], axis=2)[..., 0])


@pytest.mark.parametrize("h,w,d,win", [(12, 17, 5, 5), (9, 9, 16, 9), (20, 31, 7, 4), (6, 40, 33, 3)])
@pytest.mark.parametrize("bx,by", [(2, 2), (3, 3), (2, 1), (1, 3), (1, 1)])
def test_box_filter_border_modes(gpu_ctx, oracle, h, w, d, win, bx, by):
    """smhip_box_filter_border: the reference's reflect / cyclic border modes (BoxFilter.cpp:29-65), per pass,
    against the oracle's restatement (parity unpinned: the hot path, and every reference output, uses
    replicate); zero padding is refused as the reference refuses it."""
    c = util.rand_dsi(h, w, d, 900 + w, "float")
    for average in (True, False):
        got = run_inplace(gpu_ctx, c, lambda b, W, H, D, re: gpu_ctx.box_filter_border(b, W, H, D, re, win, average, bx, by))
        util.assert_bit_equal(got, oracle.box_filter_border(c, win, bx, by, average), "box %d/%d" % (bx, by))
    if (bx, by) == (1, 1):
        util.assert_bit_equal(got, oracle.box_filter(c, win, False), "replicate = smhip_box_filter")
    with pytest.raises(capi.SmhipError, match="border modes"):
        gpu_ctx.box_filter_border(up(gpu_ctx, c), w, h, d, w * d, win, 1, 0, 1)


# AggrBayesian is the one operator in TOLERANCE MODE: exp / log come from the device's math
# library, about 1 ulp from the host libm the oracle (and the reference) calls, and an energy is
# -log of a sum of such values.  Stated tolerance: |E_gpu - E_oracle| <= BAYES_RTOL * max(1, |E|)
# per iteration chain of up to 5 iterations.
BAYES_RTOL = 1e-5


@pytest.mark.parametrize("h,w,d,pad", [(6, 20, 7, 0), (5, 9, 64, 3), (4, 11, 99, 0), (3, 7, 128, 0),
                                       (3, 13, 65, 1), (1, 9, 16, 0), (7, 1, 5, 0), (2, 2, 2, 0),
                                       (3, 6, 129, 0), (2, 5, 200, 1), (2, 7, 256, 0), (2, 4, 300, 0),
                                       (1, 3, 512, 0)])
@pytest.mark.parametrize("n_iter", [1, 5])
@pytest.mark.parametrize("prior", [(0.4, 0.01), (3.0, 0.05), (1.5, 0.0), (30.0, 0.01)])
def test_bayesian_iter(gpu_ctx, oracle, h, w, d, pad, n_iter, prior):
    """smhip_bayesian_iter vs the oracle's restatement of StcDiffusion.cpp:139-308 -- parity
    unpinned (the reference holds no fixture for aggr_fn 5), tolerance stated above."""
    c = util.rand_dsi(h, w, d, 60 + d, "float") * np.float32(0.4)      # energies 0 .. 6.4 after scaling
    c[::2, ::3, ::4] *= np.float32(40.0)                               # and some whose exp underflows
    sp, ep = prior        # narrow band + constant (the reference's default), wide bands, full matrix
    want = oracle.bayesian_iters(c, n_iter, 0.016, 0.5, sp, ep)
    re = w * d + pad
    es, e0 = gpu_ctx.alloc(h * re * 4), gpu_ctx.alloc(h * re * 4)
    def run(b, W, H, D, r):
        for it in range(n_iter):
            gpu_ctx.bayesian_iter(b, es, e0, W, H, D, r, 0.016, 0.5, sp, ep, it)
    got = run_inplace(gpu_ctx, c, run, pad)
    err = np.abs(got.astype(np.float64) - want) / np.maximum(1.0, np.abs(want))
    assert np.isfinite(got).all()
    assert err.max() <= BAYES_RTOL, "max relative deviation %.3g" % err.max()
    same = np.mean(np.argmin(got, axis=2) == np.argmin(want, axis=2))
    assert same >= 0.99, "winners agree on %.4f of the pixels" % same


def test_bayesian_underflow_and_limits(gpu_ctx, oracle):
    """Energies so large that every exp underflows take the uniform branch (:238-241); more than
    SMHIP_BAYES_MAX_LEVELS levels and aliased volumes are refused."""
    h, w, d = 3, 5, 9
    c = np.full((h, w, d), 1.0e4, np.float32) + util.rand_dsi(h, w, d, 3, "float")
    want = oracle.bayesian_iters(c, 2, 1.0, 0.5, 0.4, 0.01)
    es, e0 = gpu_ctx.alloc(c.nbytes), gpu_ctx.alloc(c.nbytes)
    def run(b, W, H, D, r):
        for it in range(2):
            gpu_ctx.bayesian_iter(b, es, e0, W, H, D, r, 1.0, 0.5, 0.4, 0.01, it)
    got = run_inplace(gpu_ctx, c, run)
    assert np.abs(got - want).max() <= BAYES_RTOL * np.abs(want).max()
    big = gpu_ctx.alloc(2 * 2 * 513 * 4)
    with pytest.raises(capi.SmhipError, match="at most 512"):
        gpu_ctx.bayesian_iter(big, es, e0, 2, 2, 513, 2 * 513, 1.0, 0.5, 0.4, 0.01, 0)
    with pytest.raises(capi.SmhipError, match="distinct"):
        gpu_ctx.bayesian_iter(es, es, e0, w, h, d, w * d, 1.0, 0.5, 0.4, 0.01, 0)


@pytest.mark.parametrize("h,w,d,b", [(5, 40, 7, 1), (4, 33, 16, 4), (3, 50, 64, 1), (3, 50, 100, 1),
                                     (3, 70, 65, 1), (3, 20, 200, 1), (2, 12, 300, 1), (2, 9, 2, 1),
                                     (2, 1, 5, 1), (2, 30, 128, 1), (1, 150, 129, 4), (2, 10, 520, 1),
                                     (1, 6, 1024, 1)])
@pytest.mark.parametrize("par", [(20.0, 8.0, 2.0, 20), (5.0, 5.0, 1.0, 3), (0.0, 5.0, 1.0, 0),
                                 (50.0, 8.0, 4.0, 80)])
def test_opt_dp(gpu_ctx, oracle, h, w, d, b, par):
    """smhip_opt_dp (wave per scan-line, systolic R chain) == the oracle's cell-by-cell restatement
    of StcOptDP.cpp, bit for bit, with integer-valued costs (ties everywhere) and with fractional
    ones (the additions of the occlusion cost do not commute) -- parity unpinned."""
    rng = np.random.RandomState(300 + d)
    c = rng.randint(0, 30, size=(h, w, d)).astype(np.float32)
    c = np.where(rng.rand(h, w, d) < 0.4, (rng.rand(h, w, d) * 60).astype(np.float32), c)
    img, _ = synth.noise_pair(w, h, b, seed=400 + d)
    p = oracle.make_params(w, h, b, disp_max=max(d - 1, 1), opt_smoothness=par[0], opt_grad_thresh=par[1],
                           opt_grad_penalty=par[2], opt_occlusion_cost=par[3])
    want = oracle.opt_dp(c, img, p)
    disp, best = gpu_ctx.alloc(h * w * 4), gpu_ctx.alloc(h * w * 4)
    gpu_ctx.opt_dp(up(gpu_ctx, c), w, h, d, w * d, up(gpu_ctx, img), w * b, b, par[0], par[1], par[2], par[3],
                   disp, best)
    got = gpu_ctx.download(disp, (h, w), np.int32)
    assert np.array_equal(got, want), "%d of %d pixels differ" % (np.sum(got != want), got.size)
    gb = gpu_ctx.download(best, (h, w), np.float32)
    util.assert_bit_equal(gb, np.take_along_axis(c, want[:, :, None], 2)[:, :, 0], "cost at the chosen level")


@pytest.mark.parametrize("h,w,d,pad", [(6, 20, 7, 0), (5, 9, 64, 3), (4, 11, 99, 0), (3, 7, 128, 0), (1, 9, 16, 0),
                                       (7, 1, 5, 0), (40, 70, 33, 1), (3, 9, 200, 0), (2, 5, 400, 1)])
@pytest.mark.parametrize("n_iter", [0, 1, 2, 3, 6])
def test_bayesian_run_equals_iterations(gpu_ctx, h, w, d, pad, n_iter):
    """smhip_bayesian_run (energies formed inside the smoothing kernel between iterations) gives the
    bits of n_iter calls of smhip_bayesian_iter: the same float operations, fewer passes."""
    c = util.rand_dsi(h, w, d, 70 + d, "float") * np.float32(0.4)
    re = w * d + pad
    host = np.full((h, re), np.float32(-7.0))
    host[:, :w * d] = c.reshape(h, w * d)
    a, a2, a0 = up(gpu_ctx, host), gpu_ctx.alloc(h * re * 4), gpu_ctx.alloc(h * re * 4)
    for it in range(n_iter):
        gpu_ctx.bayesian_iter(a, a2, a0, w, h, d, re, 0.016, 0.5, 0.4, 0.01, it)
    want = gpu_ctx.download(a, (h, re), np.float32)
    b, b2, b0 = up(gpu_ctx, host), up(gpu_ctx, np.zeros((h, re), np.float32)), gpu_ctx.alloc(h * re * 4)
    in_b = gpu_ctx.bayesian_run(b, b2, b0, w, h, d, re, 0.016, 0.5, 0.4, 0.01, n_iter)
    assert in_b == (n_iter > 0 and n_iter % 2 == 0)
    got = gpu_ctx.download(b2 if in_b else b, (h, re), np.float32)
    util.assert_bit_equal(got[:, :w * d], want[:, :w * d], "energies after %d iterations" % n_iter)
    if pad and not in_b:
        assert np.all(got[:, w * d:] == np.float32(-7.0)), "row padding was overwritten"


@pytest.mark.parametrize("w,h,D,lo,n,sign,dmin", [(40, 9, 16, 0, 16, 1, 0), (33, 7, 64, 20, 13, 1, -3),
                                                     (50, 6, 200, 64, 136, -1, 2), (21, 5, 12, 11, 1, 1, 0),
                                                     (300, 4, 384, 128, 256, 1, 0)])
def test_pad_wta_keys_on_a_slab(gpu_ctx, oracle, w, h, D, lo, n, sign, dmin):
    """smhip_pad_wta_keys: PadCosts + OptWTA restricted to levels [lo, lo + n) of a padded volume, keys
    with global levels; the slab buffer ends up padded like the oracle's."""
    from stereomatch_amd import sharding
    full = util.rand_dsi(h, w, D, 50, "float")
    seq = dict(disp_min=dmin, disp_max=dmin + D - 1)
    if sign < 0:
        seq.update(frame_ref=1, frame_match=0)
    po = oracle.make_params(w, h, 1, **seq)
    padded = oracle.pad_costs(full, po)[:, :, lo:lo + n]
    wd, wb = oracle.wta(np.ascontiguousarray(padded))
    slab = np.ascontiguousarray(full[:, :, lo:lo + n])
    d_slab, keys = up(gpu_ctx, slab), gpu_ctx.alloc(w * h * 8)
    p = capi.match_params(w, h, 1, D, disp_min=dmin, frame_diff_sign=sign)
    gpu_ctx.pad_wta_keys(d_slab, w * n, p, lo, n, keys)
    d, c = sharding.decode_keys(gpu_ctx.download(keys, (w, h), np.int64))
    assert np.array_equal(d.T, wd + lo)
    util.assert_bit_equal(np.ascontiguousarray(c.T), wb, "best cost")
    util.assert_bit_equal(gpu_ctx.download(d_slab, (h, w, n), np.float32), padded, "padded slab")
