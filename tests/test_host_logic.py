"""Host-side logic that needs neither GPU nor oracle."""
import hashlib
import os

import numpy as np

from stereomatch_amd import pnm, synth, matcher


def test_synth_is_deterministic():
    l, r, d = synth.scene_pair(96, 64, 16)
    l2, r2, d2 = synth.scene_pair(96, 64, 16)
    assert np.array_equal(l, l2) and np.array_equal(r, r2) and np.array_equal(d, d2)
    assert hashlib.md5(synth.splitmix64(synth.SEED_SCENE, 4).tobytes()).hexdigest() == \
        hashlib.md5(synth.splitmix64(synth.SEED_SCENE, 8)[:4].tobytes()).hexdigest()
    # splitmix64 reference value: seed 0, first output
    assert int(synth.splitmix64(0, 1)[0]) == 0xE220A8397B1DCDAF
    a, b = synth.noise_pair(8, 4, 4)
    assert a.shape == (4, 8, 4) and not np.array_equal(a, b)
    assert d.min() >= 0 and d.max() <= 15


def test_pnm_roundtrip_and_bgra(tmp_path):
    img = (np.arange(6 * 5, dtype=np.uint8) * 7).reshape(5, 6)
    p = str(tmp_path / "a.pgm")
    pnm.write_pgm(p, img)
    assert open(p, "rb").read().startswith(b"P5\n6 5\n255\n")      # ImageIO.cpp:431-432
    assert np.array_equal(pnm.read_image(p)[..., 0], img)
    rgb = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    q = str(tmp_path / "c.ppm")
    with open(q, "wb") as f:
        f.write(b"P6\n# comment\n3 2\n255\n" + rgb.tobytes())
    got = pnm.read_image(q)
    assert got.shape == (2, 3, 4)
    assert np.array_equal(got[..., 0], rgb[..., 2]) and np.array_equal(got[..., 2], rgb[..., 0])
    assert np.all(got[..., 3] == 255)                                # ImageIO.cpp:390-396


def test_tsukuba_fixture_shapes(tsukuba):
    ref, mtc = tsukuba
    assert ref.shape == mtc.shape == (288, 384, 4)


def test_parameter_defaults():
    p = matcher.StereoParameters()
    # StereoParameters.cpp:39-45, 65-79
    assert (p.frame_ref, p.frame_match, p.disp_min, p.disp_max, p.disp_step) == (0, 1, 0, 7, 1.0)
    assert (p.match_fn, p.match_interp, p.match_max, p.match_interval) == (1, 3, 1000, 0)
    assert (p.aggr_fn, p.aggr_window_size, p.aggr_iter, p.aggr_minfilter) == (1, 7, 1, 0)
    assert abs(p.diff_lambda - 0.15) < 1e-9 and p.diff_beta == 0.5 and p.opt_fn == 1


def test_pmf_roundtrip(tmp_path):
    """PMF (P9): width, height, bands, native float rows (ImageIO.cpp:352-376, :428-441)."""
    rng = np.random.RandomState(5)
    a = rng.randn(7, 5, 3).astype(np.float32)
    a[0, 0, 0] = -0.0
    path = str(tmp_path / "x.pmf")
    pnm.write_pmf(path, a)
    raw = open(path, "rb").read()
    assert raw.startswith(b"P9\n5 7\n3\n") and len(raw) == 9 + a.nbytes
    b = pnm.read_image(path)
    assert b.dtype == np.float32 and b.shape == a.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
