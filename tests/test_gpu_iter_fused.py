"""smhip_fused_iter_wta_keys -- the fused path of the iterated aggregators (raw costs -> n iterations of the
binomial 14641 / diffusion / membrane -> PadCosts -> OptWTA -> keys; reference sequence StcRawCosts.cpp:229-435,
StcAggregate.cpp:280-318 and :365, StcOptimize.cpp:26-56) -- against the CPU oracle, through the C-ABI, bit for
bit: disparity and best cost.  Binomial with <= 4 iterations is ONE kernel without any DSI (smhip_iter_px.h),
5 and more bracket the multi-iteration passes between its raw-source and its keys-sink kernel; everything else
takes the stage calls on a private DSI.  Parity unpinned beyond the tsukuba anchors (SADbf04 pins 4 binomial
iterations), as for every operator: the oracle is the restatement."""
import numpy as np
import pytest

import util
from stereomatch_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_keys(ctx, left, right, p, aggr_fn, n_iter, slabs=1, **kw):
    h, w, b = left.shape
    d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
    keys = ctx.alloc(w * h * 8)
    D = p.disp_n
    bounds = [D * i // slabs for i in range(slabs + 1)]
    merged = None
    for i in range(slabs):
        ctx.fused_iter_wta_keys(d_ref, d_mtc, w * b, p, aggr_fn, n_iter, bounds[i], bounds[i + 1] - bounds[i], keys, **kw)
        k = ctx.download(keys, (w, h), np.int64)
        merged = k if merged is None else np.minimum(merged, k)
    dk = ctx.upload(merged)
    disp, best = ctx.alloc(w * h * 4), ctx.alloc(w * h * 4)
    ctx.keys_decode(dk, w, h, disp, best)
    return ctx.download(disp, (h, w), np.int32), ctx.download(best, (h, w), np.float32)


def check(ctx, oracle, w, h, b, D, aggr_fn, n_iter, slabs=1, dmin=0, step=1, sign=1, seed=None, noise=False, **kw):
    if noise:
        left, right = synth.noise_pair(w, h, b, seed=seed or 77)
    else:
        left, right, _ = synth.scene_pair(w, h, D, n_bands=b, **({"seed": seed} if seed else {}))
    fr, fm = (0, 1) if sign > 0 else (1, 0)
    op = oracle.make_params(w, h, b, disp_min=dmin, disp_max=dmin + (D - 1) * step, disp_step=float(step), frame_ref=fr,
                            frame_match=fm, aggr_fn=aggr_fn, aggr_iter=n_iter, **kw)
    assert op.disp_n == D
    odisp, obest = oracle.compute_correspondence(left, right, op)[:2]
    p = capi.match_params(w, h, b, D, disp_min=dmin, disp_num=step, frame_diff_sign=sign,
                          **{k: v for k, v in kw.items() if k.startswith("match_")})
    extra = {k: kw[k] for k in ("diff_lambda", "diff_beta") if k in kw}
    disp, best = run_keys(ctx, left, right, p, aggr_fn, n_iter, slabs, **extra)
    what = "%dx%dx%d b=%d fn=%d x%d %r" % (w, h, D, b, aggr_fn, n_iter, kw)
    util.assert_bit_equal(disp, odisp, "disparity " + what)
    util.assert_bit_equal(best, obest, "best cost " + what)


@pytest.mark.parametrize("n_iter", [1, 2, 3, 4])
@pytest.mark.parametrize("w,h,b,D,kw", [
    (128, 40, 1, 16, dict(match_fn=1, match_interval=1)),
    (252, 70, 1, 40, dict(match_fn=1, match_interval=1, match_max=20)),     # several tiles, levels % 16 != 0
    (120, 33, 1, 37, dict(match_fn=2, match_max=50)),                        # levels % 4 != 0
    (96, 40, 4, 16, dict(match_fn=2)),
    (132, 45, 4, 21, dict(match_fn=1, match_interval=1, match_interpolated=1)),
    (8, 3, 1, 4, dict(match_fn=1)),
    (364, 150, 1, 64, dict(match_fn=1, match_interval=1)),                   # three row segments at 64 rows
])
def test_binomial_one_kernel(gpu_ctx, oracle, w, h, b, D, kw, n_iter):
    check(gpu_ctx, oracle, w, h, b, D, 2, n_iter, **kw)


@pytest.mark.parametrize("n_iter", [5, 6, 7, 8, 9, 13])
@pytest.mark.parametrize("w,h,b,D,kw", [
    (252, 70, 1, 40, dict(match_fn=1, match_interval=1)),
    (96, 40, 4, 16, dict(match_fn=2)),
    (128, 150, 1, 64, dict(match_fn=1, match_max=7)),
])
def test_binomial_two_ends(gpu_ctx, oracle, w, h, b, D, kw, n_iter):
    check(gpu_ctx, oracle, w, h, b, D, 2, n_iter, **kw)


@pytest.mark.parametrize("n_iter", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("w,h,b,D,kw", [
    (128, 40, 1, 16, dict(match_fn=1, match_interval=1)),
    (252, 70, 1, 40, dict(match_fn=1, match_interval=1, match_max=20, diff_lambda=0.2)),
    (120, 33, 1, 37, dict(match_fn=2, match_max=50)),
    (96, 40, 4, 16, dict(match_fn=2)),
    (8, 3, 1, 4, dict(match_fn=1)),
    (364, 150, 1, 64, dict(match_fn=1, match_interval=1, diff_lambda=0.05)),
])
def test_diffusion_one_kernel(gpu_ctx, oracle, w, h, b, D, kw, n_iter):
    check(gpu_ctx, oracle, w, h, b, D, 3, n_iter, **kw)


@pytest.mark.parametrize("n_iter", [7, 10, 12, 13, 25, 50])
@pytest.mark.parametrize("w,h,b,D,kw", [
    (252, 70, 1, 40, dict(match_fn=1, match_interval=1)),
    (96, 40, 4, 16, dict(match_fn=2, diff_lambda=0.1)),
    (128, 150, 1, 64, dict(match_fn=1, match_max=7)),
])
def test_diffusion_several_passes(gpu_ctx, oracle, w, h, b, D, kw, n_iter):
    check(gpu_ctx, oracle, w, h, b, D, 3, n_iter, **kw)


@pytest.mark.parametrize("n_iter", [2, 4, 6])
@pytest.mark.parametrize("dmin,step,sign", [(3, 1, 1), (0, 1, -1), (2, 2, 1), (-5, 3, -1), (0, 16, 1)])
def test_directions_and_steps(gpu_ctx, oracle, dmin, step, sign, n_iter):
    for fn in (2, 3):
        check(gpu_ctx, oracle, 200, 36, 1, 24, fn, n_iter, dmin=dmin, step=step, sign=sign, match_fn=1, match_interval=1)
        check(gpu_ctx, oracle, 100, 20, 4, 12, fn, n_iter, dmin=dmin, step=step, sign=sign, match_fn=1)


@pytest.mark.parametrize("slabs", [2, 3, 5])
@pytest.mark.parametrize("n_iter", [3, 4, 8])
def test_slabs_merge_to_the_whole_range(gpu_ctx, oracle, slabs, n_iter):
    check(gpu_ctx, oracle, 236, 50, 1, 72, 2, n_iter, slabs=slabs, match_fn=1, match_interval=1, match_max=30)
    check(gpu_ctx, oracle, 236, 50, 1, 72, 3, n_iter + 2, slabs=slabs, match_fn=1, match_interval=1, match_max=30)


@pytest.mark.parametrize("opts", [{"iter_groups": 1}, {"iter_groups": 3}, {"iter_groups": 16}, {"tseg": 16}, {"tseg": 8},
                                  {"iter_fused": 0}, {"tap_slow": 1}, {"iter_direct": 1}, {"iter_direct": 1, "tseg": 64}])
@pytest.mark.parametrize("n_iter", [4, 7])
def test_launch_shapes(gpu_ctx, oracle, opts, n_iter):
    with gpu_ctx.options(**opts):
        check(gpu_ctx, oracle, 248, 90, 1, 80, 2, n_iter, match_fn=1, match_interval=1)
        check(gpu_ctx, oracle, 124, 70, 4, 20, 2, n_iter, match_fn=2, match_max=12)
        check(gpu_ctx, oracle, 248, 90, 1, 80, 3, n_iter + 3, match_fn=1, match_interval=1)


@pytest.mark.parametrize("n_iter", [2, 4, 6, 9])
@pytest.mark.parametrize("h,tseg", [(33, -1), (64, -1), (97, -1), (200, 64), (300, 128), (161, 96)])
def test_direct_key_ring(gpu_ctx, oracle, h, tseg, n_iter):
    """One level tile per workgroup, keys through the two-chunk LDS ring (iter_direct): heights around the chunks of
    32 rows, several segments, segments that end inside a chunk."""
    with gpu_ctx.options(iter_direct=1, tseg=tseg):
        check(gpu_ctx, oracle, 244, h, 1, 40, 2, min(n_iter, 8), match_fn=1, match_interval=1, match_max=9)
        check(gpu_ctx, oracle, 120, h, 4, 36, 3, n_iter, match_fn=2, slabs=2)


@pytest.mark.parametrize("n_iter", [1, 4, 8])
def test_tie_stress_on_noise(gpu_ctx, oracle, n_iter):
    """Truncated costs on a pair without structure: many exact ties, the lowest level must win."""
    check(gpu_ctx, oracle, 256, 60, 1, 48, 2, n_iter, noise=True, match_fn=1, match_max=2)
    check(gpu_ctx, oracle, 256, 60, 1, 48, 2, n_iter, noise=True, match_fn=1, match_max=1, match_interval=1)
    check(gpu_ctx, oracle, 256, 60, 1, 48, 3, n_iter + 1, noise=True, match_fn=1, match_max=2)


@pytest.mark.parametrize("w,h,b,D,aggr_fn,n_iter,kw", [
    (130, 40, 1, 16, 2, 4, dict(match_fn=1)),                                # width % 4 != 0: stage calls
    (128, 40, 1, 16, 2, 0, dict(match_fn=1, match_interval=1)),              # no iteration at all
    (130, 40, 1, 16, 3, 5, dict(match_fn=1, match_interval=1)),              # diffusion, width % 4 != 0
    (96, 30, 4, 14, 3, 10, dict(match_fn=2, diff_lambda=0.2)),               # ... several passes, levels % 4 != 0
    (128, 40, 1, 16, 4, 3, dict(match_fn=1, diff_beta=0.5)),                 # membrane
    (100, 30, 1, 20, 4, 7, dict(match_fn=1, match_interval=1, diff_lambda=0.1, diff_beta=0.25)),
    (128, 40, 1, 16, 2, 3, dict(match_fn=1, match_interval=1, match_interpolated=1)),   # reference intervals
])
def test_every_other_case_takes_the_stage_calls(gpu_ctx, oracle, w, h, b, D, aggr_fn, n_iter, kw):
    check(gpu_ctx, oracle, w, h, b, D, aggr_fn, n_iter, **kw)
    check(gpu_ctx, oracle, w, h, b, D, aggr_fn, n_iter, slabs=2, **kw)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SMHIP_FUZZ_ITER_N", "60"))))   # more on demand
def test_random_shapes(gpu_ctx, oracle, seed):
    """Random frames (several pixel tiles and row segments, widths % 4 == 0 mostly), level counts, iteration counts
    on both sides of the one-kernel limits, directions, steps, slabs and launch shapes."""
    r = np.random.RandomState(4100 + seed)
    w = int(r.randint(4, 110)) * 4 if r.rand() < 0.85 else int(r.randint(9, 300))
    h = int(r.choice([int(r.randint(3, 40)), int(r.randint(40, 200))]))
    b = int(r.choice([1, 1, 4]))
    D = int(r.choice([int(r.randint(2, 20)), int(r.randint(20, 100))]))
    fn = int(r.choice([2, 3]))
    n_iter = int(r.choice([int(r.randint(0, 5)), int(r.randint(5, 14))]))
    step = int(r.choice([1, 1, 2, 3]))
    sign = int(r.choice([1, -1]))
    dmin = int(r.randint(-6, 7))
    kw = dict(match_fn=int(r.choice([1, 2])), match_interval=int(r.rand() < 0.6), match_max=int(r.choice([1000, 30, 3])))
    if fn == 3:
        kw["diff_lambda"] = float(r.choice([0.15, 0.05, 0.2, 0.25]))
    slabs = int(r.randint(1, 4)) if D >= 6 else 1
    opts = dict(iter_groups=int(r.choice([-1, -1, 1, 2, 5])), tseg=int(r.choice([-1, -1, 16, 64])),
                iter_direct=int(r.choice([-1, 0, 1, 1])))
    with gpu_ctx.options(**opts):
        check(gpu_ctx, oracle, w, h, b, D, fn, n_iter, slabs=slabs, dmin=dmin, step=step, sign=sign, seed=900 + seed,
              noise=bool(r.rand() < 0.3), **kw)


def test_argument_errors(gpu_ctx):
    w, h, D = 64, 16, 8
    left, right, _ = synth.scene_pair(w, h, D)
    p = capi.match_params(w, h, 1, D)
    d_ref, d_mtc, keys = gpu_ctx.upload(left), gpu_ctx.upload(right), gpu_ctx.alloc(w * h * 8)
    with pytest.raises(capi.SmhipError, match="aggr_fn"):
        gpu_ctx.fused_iter_wta_keys(d_ref, d_mtc, w, p, 1, 2, 0, D, keys)
    with pytest.raises(capi.SmhipError, match="slab"):
        gpu_ctx.fused_iter_wta_keys(d_ref, d_mtc, w, p, 2, 2, 4, D, keys)
    with pytest.raises(capi.SmhipError, match="negative"):
        gpu_ctx.fused_iter_wta_keys(d_ref, d_mtc, w, p, 2, -1, 0, D, keys)
    ph = capi.match_params(w, h, 1, 2 * D - 1, disp_den=2)
    with pytest.raises(capi.SmhipError, match="disp_step"):
        gpu_ctx.fused_iter_wta_keys(d_ref, d_mtc, w, ph, 2, 2, 0, D, keys)
