"""AggrBayesian on the device: deviation from the oracle (tolerance mode) and time per iteration."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from stereomatch_amd import capi
from oracle import oracle
import util

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
for (h, w, d) in [(288, 384, 16), (1080, 1920, 64), (1080, 1920, 128)]:
    nbytes = h * w * d * 4
    b, es, e0 = ctx.alloc(nbytes), ctx.alloc(nbytes), ctx.alloc(nbytes)
    ctx.L.smhip_memcpy_d2d  # noqa
    for it in range(3):
        ctx.bayesian_iter(b, es, e0, w, h, d, w * d, 0.016, 0.5, 0.4, 0.01, it)
    ctx.sync()
    t = time.perf_counter()
    n = 10
    for it in range(1, n + 1):
        ctx.bayesian_iter(b, es, e0, w, h, d, w * d, 0.016, 0.5, 0.4, 0.01, it)
    ctx.sync()
    ms = (time.perf_counter() - t) / n * 1e3
    print("%dx%dx%d: %.3f ms / iteration (%.1f GB/s of 20 B/voxel, %.2f TFLOP/s f64 of the DxD product)" % (
        w, h, d, ms, 20.0 * h * w * d / ms / 1e6, 2.0 * h * w * d * d / ms / 1e9), flush=True)
    for x in (b, es, e0):
        x.free()
