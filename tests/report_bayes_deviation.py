"""Prints how far smhip_bayesian_iter (tolerance mode) is from the oracle: run on the GPU box as
`python tests/report_bayes_deviation.py`.  Not collected by pytest; the bound itself is asserted in
tests/test_gpu_ops.py::test_bayesian_iter."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
from stereomatch_amd import capi  # noqa: E402
from oracle import oracle  # noqa: E402
import util  # noqa: E402

ctx = capi.Context(0)
for (h, w, d, n) in [(64, 96, 16, 1), (64, 96, 16, 5), (64, 96, 16, 20), (32, 40, 99, 5), (32, 40, 128, 10)]:
    c = util.rand_dsi(h, w, d, 5, "float") * np.float32(0.4)
    want = oracle.bayesian_iters(c, n, 0.016, 0.5, 0.4, 0.01)
    b = ctx.upload(c)
    es, e0 = ctx.alloc(c.nbytes), ctx.alloc(c.nbytes)
    for it in range(n):
        ctx.bayesian_iter(b, es, e0, w, h, d, w * d, 0.016, 0.5, 0.4, 0.01, it)
    got = ctx.download(b, c.shape, np.float32)
    err = np.abs(got.astype(np.float64) - want) / np.maximum(1.0, np.abs(want))
    print("%dx%dx%d x%d: max rel dev %.3g, mean %.3g, equal %.4f, winners equal %.5f" % (
        w, h, d, n, err.max(), err.mean(), np.mean(got == want),
        np.mean(got.argmin(2) == want.argmin(2))), flush=True)
