#!/usr/bin/env python3
"""bench.py -- cost-volume throughput of the fused raw-cost -> 9x9 box -> WTA hot path.

N = 1 (BASELINE.json configs[2], the configuration the metric's target is quoted on):
    synthetic rectified pair 1920x1080, b = 1, D = 256, BT-interpolated SAD
    (match_fn 1, match_interval 1, match_max 1000), 9x9 box aggregation, PadCosts, WTA.
N > 1: weak scaling by disparity slab -- every rank runs the same 256-level slab size on the
    same image pair, the problem has D = 256*N levels, and the per-pixel (cost, d) keys are
    min-reduced with ONE RCCL all-reduce (int64 MIN) per step, then decoded.  Steps are frames
    of a stream: the all-reduce of frame i runs under the kernels of frame i + 1
    (sharding.KeyReducePipeline, two key buffers); every frame is reduced and decoded inside
    the timed region.

A step = one pass of the hot path over the resident image pair (images already in HBM).
value = (W*H*D_total voxels of all ranks) / (max-over-ranks wall time of K steps).

The JSON line also carries `roofline` (dominant kernel, HIP events recorded inside the timed
region on the launch stream), `roofline_fused` (both passes against the 8N + images + outputs
algorithmic bytes of SURVEY 8d) and `cpu_baseline` (the CPU oracle on one host core, on a
bounded sample of the same workload; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disp-per-gpu", type=int, default=256)
    ap.add_argument("--window", type=int, default=9)
    ap.add_argument("--bands", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the short config 4 / config 5 measurements reported under 'extra'")
    ap.add_argument("--cpu-rows", type=int, default=1080,
                    help="rows of the workload the CPU baseline is timed on")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "N > 1 code path where the ranks cannot have a GPU each)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from rocprofv3 --pmc passes (see profiles/README.md)")
    args = ap.parse_args()
    if os.environ.get("BENCH_WATCHDOG"):      # rehearsal aid: dump the stacks and exit instead of hanging
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)

    import numpy as np
    import torch
    import torch.distributed as dist
    from stereomatch_amd import capi, sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    W, H, b, win = args.width, args.height, args.bands, args.window
    Dg = args.disp_per_gpu
    D = Dg * world
    left, right, _ = synth.scene_pair(W, H, min(D, 256), n_bands=b)

    ctx = capi.Context(local_rank)
    # an explicit stream for the library's kernels and torch's collectives alike: the legacy null
    # stream synchronises implicitly with every blocking stream of the process, which neither the
    # overlap of the all-reduce with the next frame nor torch's helper threads want
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    t_ref = torch.from_numpy(left.reshape(H, W * b)).to(dev)
    t_mtc = torch.from_numpy(right.reshape(H, W * b)).to(dev)
    keys = torch.empty((W, H), dtype=torch.int64, device=dev)   # column-major key image
    keys2 = torch.empty_like(keys) if world > 1 else None
    disp = torch.empty((H, W), dtype=torch.int32, device=dev)
    best = torch.empty((H, W), dtype=torch.float32, device=dev)
    p = capi.match_params(W, H, b, D, match_fn=1, match_interval=1, match_max=1000)
    d_first = rank * Dg

    def produce(buf, _frame):
        ctx.fused_box_wta_keys(t_ref.data_ptr(), t_mtc.data_ptr(), W * b, p, win, d_first, Dg,
                               buf.data_ptr())

    def consume(buf, _frame):
        ctx.keys_decode(buf.data_ptr(), W, H, disp.data_ptr(), best.data_ptr())

    # N > 1: the one exchange step (int64 MIN all-reduce over xGMI) overlaps the next frame
    pipe = sharding.KeyReducePipeline([keys, keys2], produce, consume) if world > 1 else None

    trace = bool(os.environ.get("BENCH_TRACE"))

    def step():
        if trace:
            print("rank %d step" % rank, file=sys.stderr, flush=True)
        if pipe is not None:
            pipe.submit(0)
        else:
            produce(keys, 0)
            consume(keys, 0)

    def fence():
        if pipe is not None:
            pipe.flush()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    profiling = not os.environ.get("BENCH_NO_PROFILE")       # rehearsal aid
    if profiling:
        ctx.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    n_prof, ms_a, ms_b = ctx.profile_end() if profiling else (1, 0.001, 0.001)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    n_vox_gpu = W * H * Dg
    n_vox = n_vox_gpu * world
    value = n_vox * args.steps / dt / 1e9
    ms_step = dt / args.steps * 1e3

    if rank == 0:
        # algorithmic bytes per launch: pass A writes the slab once and reads the
        # images; pass B reads the slab once and writes one 8-byte key per pixel.
        n_bt = 3 if p.match_interval else 2          # ref + (mn, mx) or ref + match
        bytes_a = 4.0 * n_vox_gpu + n_bt * b * W * H
        bytes_b = 4.0 * n_vox_gpu + 8.0 * W * H
        bytes_fused = 8.0 * n_vox_gpu + 2.0 * b * W * H + 8.0 * W * H
        ta, tb = ms_a / n_prof * 1e-3, ms_b / n_prof * 1e-3
        traffic = {}
        if os.path.exists(args.traffic_json):
            with open(args.traffic_json) as f:
                traffic = json.load(f)

        def roof(name, nbytes, secs):
            ach = nbytes / secs / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic.get(name), "avg_ms": round(secs * 1e3, 4),
                    "algorithmic_bytes": int(nbytes)}
        ka = roof("fused_rawbox_h_lds_kernel", bytes_a, ta)
        kb = roof("fused_boxv_wta_vpl_kernel", bytes_b, tb)
        dominant = ka if ta >= tb else kb
        fused = roof("fused pair (pass A + pass B)", bytes_fused, ta + tb)

        out = {
            "metric": "cost-volume throughput (G pixel*disparity/s)",
            "value": round(value, 2), "unit": "G pixel*disparity/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "fused BT-SAD raw cost + %dx%d box + PadCosts + WTA, %dx%d, b=%d, "
                            "D=%d (%d levels per GPU%s)" % (
                                win, win, W, H, b, D, Dg,
                                ", int64-MIN %s all-reduce of keys" % ("RCCL" if args.backend == "nccl" else args.backend)
                                if world > 1 else ""),
                "width": W, "height": H, "bands": b, "disp_total": D, "disp_per_gpu": Dg,
                "window": win, "match_fn": "AD", "match_interval": 1, "match_max": 1000,
                "parallelism": "disparity slabs x%d" % world,
                "collective_backend": (args.backend if world > 1 else None)},
            "roofline": {k: dominant[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")},
            "roofline_kernel": dominant["kernel"],
            "roofline_fused": fused,
            "kernels": [ka, kb],
        }
        if world == 1 and not args.no_extra:
            out["extra"] = extra_configs(np, capi, synth, ctx)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, left, right, args, Dg, win)
            out["cpu_baseline_all_cores"] = cpu_baseline(np, left, right, args, Dg, win, openmp=True)
            if "extra" in out:
                out["extra"]["config4_cpu"] = cpu_config4(np, synth)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def extra_configs(np, capi, synth, ctx):
    """BASELINE configs 4 and 5 (per-GPU share), a few repetitions each, reported next to the
    headline line (not part of `value`): iterated aggregation on a resident 3840x2160x512 DSI and
    the fused path on a 7680x4320 slab of 128 levels.  frac = SURVEY 8d algorithmic bytes / time /
    8 TB/s.  Failures are reported, they do not stop the bench."""
    res = {}

    def timed(fn, reps):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps

    try:
        W, H, D = 3840, 2160, 512
        N = W * H * D
        left, right, _ = synth.scene_pair(W, H, 256)
        p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
        d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
        a, b = ctx.alloc(N * 4), ctx.alloc(N * 4)
        ctx.raw_costs(d_ref, d_mtc, W, p, a, W * D)
        for name, iters, fn in (
                ("binomial_x4", 4, lambda: ctx.convolve_14641_run(a, b, W, H, D, W * D, 4)),
                ("binomial_x8", 8, lambda: ctx.convolve_14641_run(a, b, W, H, D, W * D, 8)),
                ("diffusion_x10", 10, lambda: ctx.diffusion_run(a, b, None, W, H, D, W * D, 0.15, 0.0, 10)),
                ("diffusion_x50", 50, lambda: ctx.diffusion_run(a, b, None, W, H, D, W * D, 0.15, 0.0, 50))):
            dt = timed(fn, 2)
            res["config4_" + name] = {
                "workload": "3840x2160, D=512, %d iterations on the resident DSI" % iters,
                "ms": round(dt * 1e3, 3), "ms_per_iteration": round(dt * 1e3 / iters, 3),
                "G_voxel_iterations_per_s": round(N * iters / dt / 1e9, 1),
                "frac_of_8N_per_iteration": round(8.0 * N * iters / dt / 1e9 / HBM_PEAK_GBS, 3)}
        a.free()
        b.free()
        # one GPU's share of config 4 under slab sharding (D = 512 over 8 GPUs): the staged pipeline on
        # levels 256..319 -- raw costs, 4 binomial iterations, PadCosts + WTA -> keys
        n, d0 = 64, 256
        ps = capi.match_params(W, H, 1, n, disp_min=d0, match_fn=1, match_interval=1)
        a, b, keys = ctx.alloc(W * H * n * 4), ctx.alloc(W * H * n * 4), ctx.alloc(W * H * 8)

        def slab_step():
            ctx.raw_costs(d_ref, d_mtc, W, ps, a, W * n)
            in_b = ctx.convolve_14641_run(a, b, W, H, n, W * n, 4)
            ctx.pad_wta_keys(b if in_b else a, W * n, p, d0, n, keys)
        dt = timed(slab_step, 3)
        res["config4_slab_pipeline"] = {
            "workload": "3840x2160, levels 256..319 of D=512 (one GPU's share of 8): raw costs + 4 binomial "
                        "iterations + PadCosts + WTA -> keys",
            "ms": round(dt * 1e3, 3), "G_pixel_disparity_per_s": round(W * H * n / dt / 1e9, 1)}
        a.free()
        b.free()
        keys.free()
    except Exception as e:      # noqa: BLE001 - reported, the headline line must still print
        res["config4_error"] = repr(e)
    try:
        W, H, D, n = 7680, 4320, 1024, 128
        left, right, _ = synth.scene_pair(W, H, 256)
        p = capi.match_params(W, H, 1, D, match_fn=1, match_interval=1)
        d_ref, d_mtc = ctx.upload(left), ctx.upload(right)
        keys = ctx.alloc(W * H * 8)
        dt = timed(lambda: ctx.fused_box_wta_keys(d_ref, d_mtc, W, p, 9, 512, n, keys), 5)
        nb = 8.0 * W * H * n + 2.0 * W * H + 8.0 * W * H
        res["config5_slab"] = {
            "workload": "7680x4320, levels 512..639 of D=1024 (one GPU's share of 8), fused path, keys out",
            "ms": round(dt * 1e3, 3), "G_pixel_disparity_per_s": round(W * H * n / dt / 1e9, 1),
            "frac_of_8TBs": round(nb / dt / 1e9 / HBM_PEAK_GBS, 3)}
    except Exception as e:      # noqa: BLE001
        res["config5_error"] = repr(e)
    return res


def cpu_baseline(np, left, right, args, D, win, openmp=False):
    """The CPU oracle (a port of the reference's single-threaded CPU path, bit-identical to it on
    the tsukuba anchors) on the top `cpu_rows` rows of the same image pair: one core (like the
    reference), or -- openmp=True -- the -fopenmp build of the same source on the host cores this
    process may use (OMP_NUM_THREADS)."""
    from oracle import oracle as O
    O.use_openmp(openmp)
    rows = min(args.cpu_rows, left.shape[0])
    l, r = np.ascontiguousarray(left[:rows]), np.ascontiguousarray(right[:rows])
    h, w, b = l.shape
    p = O.make_params(w, h, b, disp_min=0, disp_max=D - 1, match_fn=1, match_interval=1,
                      match_max=1000, aggr_window_size=win)
    O.lib()
    cores = O.threads()
    t0 = time.perf_counter()
    O.compute_correspondence(l, r, p)
    dt = time.perf_counter() - t0
    O.use_openmp(False)
    return {"value": round(w * h * D / dt / 1e9, 5), "unit": "G pixel*disparity/s", "cores": cores,
            "kind": "port",
            "sample": "top %d of %d rows of the same %dx%d D=%d pair (%.1f s wall); "
                      "oracle/sm_oracle.c, gcc -O2%s" % (
                          rows, left.shape[0], w, left.shape[0], D, dt,
                          " -fopenmp, rows / column blocks over %d threads" % cores if openmp
                          else ", single thread like the reference")}


def cpu_config4(np, synth):
    """CPU side of config 4 (size-normalised, on a sub-volume the CPU path can hold): one
    binomial and one diffusion iteration of the oracle on 3840x2160 x 4 levels, single thread."""
    from oracle import oracle as O
    W, H, D = 3840, 2160, 4
    left, right, _ = synth.scene_pair(W, H, 256)
    p = O.make_params(W, H, 1, disp_min=0, disp_max=D - 1, match_fn=1, match_interval=1)
    c = O.raw_costs(left, right, p)
    res = {"sub_volume": "3840x2160 x %d levels (of 512), oracle/sm_oracle.c, one thread" % D}
    for name, fn in (("binomial", lambda: O.convolve_14641(c)), ("diffusion", lambda: O.diffusion_iter(c, 0.15, 0.0, None))):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        res[name + "_G_voxel_iterations_per_s"] = round(W * H * D / dt / 1e9, 4)
    return res


if __name__ == "__main__":
    main()
