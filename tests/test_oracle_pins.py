"""The oracle against the reference's recorded outputs, and against independent numpy
restatements of each operator (pure definitions, small sizes)."""
import numpy as np
import pytest

import util
from stereomatch_amd import pnm

F = np.float32


def test_reference_md5_anchors(oracle, tsukuba):
    """Five disparity maps + SSD09 final_energy produced by the reference CPU path
    (BASELINE.md section 2).  This is what pins the oracle."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    a = util.anchors()["reference"]
    grid = dict(util.exp_grid() + util.exp4_grid())
    assert set(a["md5"]) <= set(grid)
    for name, want in a["md5"].items():
        p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **grid[name])
        _, _, depth, e = oracle.compute_correspondence(ref, mtc, p)
        assert util.md5(pnm.pgm_bytes(depth)) == want, name
        if name in a["final_energy"]:
            assert e == a["final_energy"][name], name


def test_oracle_golden_regression(oracle, tsukuba):
    """The committed oracle vectors are what the oracle still produces (subset, for speed)."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    a = util.anchors()["oracle"]["md5"]
    grid = dict(util.exp_grid() + util.exp4_grid())
    for name in ("SAD09t02", "SSDmf09bt05", "SADmf09", "SADbf02", "SADmem10b05"):
        p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **grid[name])
        _, _, depth, _ = oracle.compute_correspondence(ref, mtc, p)
        assert util.md5(pnm.pgm_bytes(depth)) == a[name], name


def np_raw_costs(ref, mtc, D, disp_min, sign, fn, mmax, interval):
    """Definition of SURVEY appendix A1/A2 in numpy integers."""
    h, w, b = ref.shape
    R, M = ref.astype(np.int64), mtc.astype(np.int64)
    cutoff = mmax * mmax if fn == 2 else abs(mmax)
    outside = min(b * (255 * 255 if fn == 2 else 255), cutoff)
    if interval:
        nxt = np.concatenate([M[:, 1:], M[:, -1:]], 1)
        hr = (M + nxt + 1) // 2
        hl = np.concatenate([M[:, :1], hr[:, :-1]], 1)
        mn = np.minimum(hl, np.minimum(M, hr))
        mx = np.maximum(hl, np.maximum(M, hr))
    out = np.empty((h, w, D), F)
    for k in range(D):
        disp = -sign * (disp_min + k)
        for x in range(w):
            xm = x + disp
            if xm < 0 or xm >= w:
                out[:, x, k] = outside
                continue
            if interval:
                d1 = np.maximum(0, np.maximum(mn[:, xm] - R[:, x], R[:, x] - mx[:, xm]))
            else:
                d1 = M[:, xm] - R[:, x]
            e = (d1 * d1 if fn == 2 else np.abs(d1)).sum(1)
            out[:, x, k] = np.minimum(e, cutoff)
    return out


@pytest.mark.parametrize("fn,mmax,interval,b,sign,dmin", [
    (1, 1000, 0, 1, 1, 0), (2, 1000, 0, 4, 1, 0), (1, 20, 1, 4, 1, 0), (2, 5, 1, 1, 1, 2),
    (1, 1000, 1, 1, -1, 0), (2, 50, 0, 4, -1, 1)])
def test_raw_costs_definition(oracle, fn, mmax, interval, b, sign, dmin):
    from stereomatch_amd import synth
    ref, mtc = synth.noise_pair(37, 11, b)
    D = 9
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    p = oracle.make_params(37, 11, b, disp_min=dmin, disp_max=dmin + D - 1, frame_ref=fr,
                           frame_match=fm, match_fn=fn, match_max=mmax, match_interval=interval)
    assert p.disp_n == D
    util.assert_bit_equal(oracle.raw_costs(ref, mtc, p),
                          np_raw_costs(ref, mtc, D, dmin, sign, fn, mmax, interval), "raw costs")


def np_box_1d(s, w):
    """Appendix A3 on axis 0 of a float32 array."""
    n = s.shape[0]
    pr = w // 2
    pl = pr - w
    scale = F(1.0 / w)
    c = lambda i: min(max(i, 0), n - 1)
    t = np.empty_like(s)
    acc = np.zeros(s.shape[1:], F)
    for i in range(pl + 1, pr + 1):
        acc = acc + scale * s[c(i)]
    t[0] = acc
    for x in range(1, n):
        t[x] = t[x - 1] + scale * (-s[c(x + pl)] + s[c(x + pr)])
    return t


@pytest.mark.parametrize("w", [3, 4, 9])
def test_box_filter_definition(oracle, w):
    c = util.rand_dsi(13, 17, 5, 1)
    hp = np.stack([np_box_1d(c[y], w) for y in range(c.shape[0])])
    want = np_box_1d(hp, w)
    util.assert_bit_equal(oracle.box_filter(c, w), want, "box")


@pytest.mark.parametrize("w", [2, 3, 9])
def test_min_filter_definition(oracle, w):
    c = util.rand_dsi(12, 15, 4, 2, "float")
    pr = w // 2
    pl = pr - w
    def mn(a, axis):
        n = a.shape[axis]
        idx = lambda i: np.clip(i, 0, n - 1)
        out = None
        for k in range(pl + 1, pr + 1):
            v = np.take(a, idx(np.arange(n) + k), axis=axis)
            out = v if out is None else np.minimum(out, v)
        return out
    util.assert_bit_equal(oracle.min_filter(c, w), mn(mn(c, 1), 0), "min")


def test_binomial_definition(oracle):
    c = util.rand_dsi(9, 11, 3, 3, "float")
    k = [F(0.0625), F(0.25), F(0.375), F(0.25), F(0.0625)]
    def conv(a, axis):
        n = a.shape[axis]
        acc = np.zeros_like(a)
        for l in range(5):
            acc = acc + k[l] * np.take(a, np.clip(np.arange(n) + l - 2, 0, n - 1), axis=axis)
        return acc
    util.assert_bit_equal(oracle.convolve_14641(c), conv(conv(c, 1), 0), "binomial")


@pytest.mark.parametrize("beta", [0.0, 0.5])
def test_diffusion_definition(oracle, beta):
    c = util.rand_dsi(8, 10, 3, 4, "float")
    c0 = util.rand_dsi(8, 10, 3, 5, "float")
    lam, be = F(0.15), F(beta)
    a = F(1) - lam * (be + F(4))
    def sh(arr, axis, k):
        n = arr.shape[axis]
        return np.take(arr, np.clip(np.arange(n) + k, 0, n - 1), axis=axis)
    want = a * c + lam * (((sh(c, 1, -1) + sh(c, 1, 1)) + sh(c, 0, -1)) + sh(c, 0, 1))
    if beta > 0:
        want = want + (lam * be) * c0
    util.assert_bit_equal(oracle.diffusion_iter(c, 0.15, beta, c0), want, "diffusion")


def test_wta_and_pad(oracle):
    c = util.rand_dsi(6, 20, 7, 6)
    c[0, 0, :] = 5.0                      # all-equal => first wins
    c[1, 3, 2] = c[1, 3, 5] = -1.0        # tie => lowest d
    disp, best = oracle.wta(c)
    assert disp[0, 0] == 0 and disp[1, 3] == 2
    assert np.array_equal(disp, np.argmin(c, axis=2))
    assert np.array_equal(best, np.min(c, axis=2))
    p = oracle.make_params(20, 6, 1, disp_min=1, disp_max=7)
    out = oracle.pad_costs(c, p)
    o = oracle.match_outside(p)
    for k in range(7):
        assert np.all(out[:, :1 + k, k] == o)
        assert np.array_equal(out[:, 1 + k:, k], c[:, 1 + k:, k])


def test_openmp_build_is_identical(oracle, tsukuba):
    """oracle/_build/libsm_oracle_omp.so (bench.py's all-cores CPU line) spreads rows and column
    blocks over threads; every chain keeps its own serial order, so the bits do not change."""
    import util
    ref, mtc = tsukuba
    h, w, b = ref.shape
    grid = dict(util.exp_grid())
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **grid["SSD09"])
    want = oracle.compute_correspondence(ref, mtc, p)
    oracle.use_openmp(True)
    try:
        assert oracle.threads() >= 1
        got = oracle.compute_correspondence(ref, mtc, p)
    finally:
        oracle.use_openmp(False)
    assert oracle.threads() == 1
    for a, g in zip(want[:3], got[:3]):
        assert a.tobytes() == g.tobytes()
    assert want[3] == got[3] == util.anchors()["reference"]["final_energy"]["SSD09"]


def test_evaluator_reproduces_recorded_statistics(oracle, tsukuba):
    """SSD09 with tsukuba/param_in.txt's eval_ignore_border 18 / eval_disp_gap 2: the three
    evaluator numbers recorded from the reference come out digit for digit;
    bad_pixels_nonocc pins the occlusion map (both ForwardWarp passes) as well."""
    import os
    import util
    from stereomatch_amd import pnm
    ref, mtc = tsukuba
    h, w, b = ref.shape
    truth = pnm.read_image(os.path.join(util.GOLDEN, "tsukuba", "truedisp.row3.col3.pgm"))
    p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **dict(util.exp_grid())["SSD09"])
    _, _, _, _, fd = oracle.compute_correspondence_f(ref, mtc, p)
    ev = oracle.evaluate(fd, truth, ref, p, **util.TSUKUBA_EVAL)
    for k, v in util.anchors()["reference"]["evaluator"]["SSD09"].items():
        assert "%f" % ev[k] == v, k


def test_stage_dsi_checksums(oracle, tsukuba):
    """Per-stage DSI checksums (CRC-32 of the float bytes after Aggregate) committed with the golden
    vectors: the oracle keeps producing the same volumes, bit for bit."""
    ref, mtc = tsukuba
    h, w, b = ref.shape
    want = util.anchors()["oracle"]["dsi_crc32"]
    assert set(want) == {n for n, _ in util.stage_cases()}
    for name, kw in util.stage_cases():
        p = oracle.make_params(w, h, b, **util.TSUKUBA_SEQ, **kw)
        assert util.crc32(oracle.aggregate(oracle.raw_costs(ref, mtc, p), p)) == want[name], name
