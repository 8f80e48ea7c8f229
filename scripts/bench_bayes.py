"""AggrBayesian on the device: time per iteration (the deviation from the oracle is measured by
tests/report_bayes_deviation.py and bounded by tests/test_gpu_ops.py::test_bayesian_iter)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from stereomatch_amd import capi

ctx = capi.Context(0)
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
    ctx.bayesian_run(b, es, e0, w, h, d, w * d, 0.016, 0.5, 0.4, 0.01, 10)
    ctx.sync()
    t = time.perf_counter()
    ctx.bayesian_run(b, es, e0, w, h, d, w * d, 0.016, 0.5, 0.4, 0.01, 10)
    ctx.sync()
    ms = (time.perf_counter() - t) / 10 * 1e3
    print("%dx%dx%d: %.3f ms / iteration inside a run of 10 (smhip_bayesian_run, 12 B/voxel: %.1f GB/s)" % (
        w, h, d, ms, 12.0 * h * w * d / ms / 1e6), flush=True)
    for x in (b, es, e0):
        x.free()
