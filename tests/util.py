"""Shared helpers of the parity tests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tsukuba/param_in.txt:1-6
TSUKUBA_SEQ = dict(disp_min=0, disp_max=15, frame_ref=2, frame_match=3, disp_scale=16.0)
TSUKUBA_EVAL = dict(ignore_border=18, disp_gap=2.0)      # tsukuba/param_in.txt:11-12


def anchors():
    with open(os.path.join(GOLDEN, "tsukuba_anchors.json")) as f:
        return json.load(f)


def md5(b):
    return hashlib.md5(b).hexdigest()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    if a.dtype == np.float32:
        a, b = bits(a), bits(b)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError("%s: %d of %d elements differ, first at %s" %
                             (what, len(bad), a.size, tuple(bad[0])))


def rand_dsi(h, w, d, seed, kind="int"):
    """Seeded DSI: 'int' = integer-valued costs (what raw costs produce), 'float' = arbitrary."""
    rng = np.random.RandomState(seed)
    if kind == "int":
        return rng.randint(0, 1021, size=(h, w, d)).astype(np.float32)
    return (rng.rand(h, w, d) * 1000.0).astype(np.float32)


def exp_grid():
    """(name, params) of Scripts/exp1_costSSD9.txt:25-64 and exp2_costSSDMF9.txt:25-64."""
    out = []
    for mf, tag in ((0, ""), (9, "mf")):
        for fn, iv, pre, suf in ((1, 0, "SAD", ""), (2, 0, "SSD", ""), (1, 1, "SAD", "b"), (2, 1, "SSD", "b")):
            for mm, t in ((1000, ""), (50, "t50"), (20, "t20"), (10, "t10"), (5, "t05"), (2, "t02"), (1, "t01")):
                name = "%s%s09%s%s" % (pre, tag, suf, t)
                out.append((name, dict(match_fn=fn, match_interval=iv, match_max=mm,
                                       aggr_window_size=9, aggr_minfilter=mf)))
    return out


def exp4_grid():
    """Scripts/exp4_aggr.txt:56-69 (binomial) plus diffusion / membrane runs in its style."""
    out = [("SADbf%02d" % it, dict(match_fn=1, aggr_fn=2, aggr_iter=it, aggr_minfilter=0))
           for it in (2, 4, 6, 8)]
    out += [("SADdiff%02d" % it, dict(match_fn=1, aggr_fn=3, aggr_iter=it)) for it in (1, 10)]
    out += [("SADmem10b05", dict(match_fn=1, aggr_fn=4, aggr_iter=10, diff_beta=0.5))]
    return out


def stage_cases():
    """Configurations whose aggregated DSI (after PadCosts) is pinned by a CRC-32 of its float bytes
    in tests/golden/tsukuba_anchors.json ("dsi_crc32"): one per stage operator of the path."""
    return [
        ("raw_SSD", dict(match_fn=2, aggr_iter=0)),
        ("raw_SADb_t20", dict(match_fn=1, match_interval=1, match_max=20, aggr_iter=0)),
        ("raw_SAD_interp_half", dict(match_fn=1, match_interpolated=1, match_interval=1, disp_step=0.5, aggr_iter=0)),
        ("box9_SSD", dict(match_fn=2, aggr_window_size=9)),
        ("box5x2_SAD", dict(match_fn=1, aggr_window_size=5, aggr_iter=2)),
        ("box9_min9_SSD_t20", dict(match_fn=2, match_max=20, aggr_window_size=9, aggr_minfilter=9)),
        ("binomial4_SAD", dict(match_fn=1, aggr_fn=2, aggr_iter=4)),
        ("diffusion10_SAD", dict(match_fn=1, aggr_fn=3, aggr_iter=10, diff_lambda=0.15)),
        ("membrane3_SADb", dict(match_fn=1, match_interval=1, aggr_fn=4, aggr_iter=3, diff_lambda=0.15, diff_beta=0.5)),
    ]


def crc32(a):
    import zlib
    import numpy as np
    return "%08x" % (zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)


def free_port():
    """A TCP port nobody listens on right now (two sessions may share a box: no fixed rendezvous ports)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
