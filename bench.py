#!/usr/bin/env python3
"""bench.py -- cost-volume throughput of the fused raw-cost -> 9x9 box -> WTA hot path.

N = 1 (BASELINE.json configs[2], the configuration the metric's target is quoted on):
    synthetic rectified pair 1920x1080, b = 1, D = 256, BT-interpolated SAD
    (match_fn 1, match_interval 1, match_max 1000), 9x9 box aggregation, PadCosts, WTA.
N > 1 (BASELINE.json configs[4], the volume north_star asks the 1 -> 8 GPU scaling on):
    synthetic 7680x4320 pair, D = 1024 in all, STRONG scaling by disparity slab -- rank r owns levels
    sharding.slab_bounds(1024, N, r), runs them in fused calls of at most 256 levels (their keys are
    min-merged locally), and the per-pixel (cost, d) keys of all ranks are min-reduced with ONE RCCL
    all-reduce (int64 MIN, 265 MB) per frame, then decoded.  Steps are frames of a stream: the
    all-reduce of frame i runs under the kernels of frame i + 1 (sharding.KeyReducePipeline, two key
    buffers); every frame is reduced and decoded inside the timed region.
    (--weak restores round 1's weak-scaling mode: config 3's frame, 256 levels per rank.)

A step = one pass of the hot path over the resident image pair (images already in HBM).
value = (W*H*D_total voxels of all ranks) / (max-over-ranks wall time of K steps).

The JSON line also carries `roofline` (dominant kernel, HIP events recorded inside the timed
region on the launch stream), `roofline_fused` (both passes against the 8N + images + outputs
algorithmic bytes of SURVEY 8d), at N = 1 `cpu_baseline` (the CPU oracle on one host core, on the same
workload; rank 0) and `verified_vs_oracle` (the frame left by the timed loop equals the oracle's), at
N > 1 the per-rank compute time and the collective time left exposed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (RCCL's buffer exchange fails with
# hipIpcGetMemHandle otherwise); the launcher's environment normally carries it already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MAX_CALL_LEVELS = 256   # levels per fused call at N > 1 (bounds the slab workspace: 34 GB at 8K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=0, help="default: 1920 at N = 1, 7680 at N > 1")
    ap.add_argument("--height", type=int, default=0, help="default: 1080 at N = 1, 4320 at N > 1")
    ap.add_argument("--disp-total", type=int, default=0, help="levels of the whole problem (default 256 / 1024)")
    ap.add_argument("--disp-per-gpu", type=int, default=256, help="levels per rank in --weak mode")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling of config 3 (256 levels per rank)")
    ap.add_argument("--window", type=int, default=9)
    ap.add_argument("--bands", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the short config 4 / config 5 measurements reported under 'extra'")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="N = 1: after the timed steps, run the same step for this long and report it under "
                         "extra.sustained (0: skip)")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="rows of the workload the CPU baseline is timed on (default: all)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "N > 1 code path where the ranks cannot have a GPU each: keys are then reduced "
                         "through pinned host memory)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--one-rank-rehearsal", action="store_true",
                    help="run the N > 1 code path in a process group of one rank (no scaling figure)")
    ap.add_argument("--check-oracle", action="store_true",
                    help="rehearsal / test: rank 0 compares EVERY decoded frame with the CPU oracle")
    ap.add_argument("--max-call-levels", type=int, default=MAX_CALL_LEVELS,
                    help="N > 1: levels per fused call (tests lower it to exercise the local key merge)")
    ap.add_argument("--collective", choices=("allreduce", "reduce_scatter"), default="allreduce",
                    help="N > 1: how the per-pixel keys of the ranks are merged -- one int64 MIN all-reduce (every "
                         "rank ends with the merged keys and decodes them), or a MIN reduce-scatter whose chunks are "
                         "gathered and decoded on rank 0 only (half the xGMI traffic)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="experiments: a launch-shape option of the library (capi.OPTIONS), e.g. b_lred=0; never a result")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from rocprofv3 --pmc passes (see profiles/README.md)")
    args = ap.parse_args()
    if os.environ.get("BENCH_WATCHDOG"):      # rehearsal aid: dump the stacks and exit instead of hanging
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)

    # stdout carries the ONE JSON line and nothing else: libraries that print there (RCCL's version banner at the
    # first collective) are sent to stderr -- file descriptor 1 is pointed at stderr for the run, the line goes to
    # a duplicate of the original
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

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
    # --one-rank-rehearsal: the N > 1 code path (process group, config 5 in slabs and calls, key pipeline with a
    # real collective, timing all-gather) in a group of ONE rank -- what a one-GPU box can run of it over RCCL
    multi = world > 1 or args.one_rank_rehearsal
    if args.one_rank_rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # a port nobody listens on (two sessions may share a box)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    sharded = multi and not args.weak          # config 5, strong scaling
    max_call = args.max_call_levels
    b, win = args.bands, args.window
    W = args.width or (7680 if sharded else 1920)
    H = args.height or (4320 if sharded else 1080)
    if sharded:
        D = args.disp_total or 1024
        d_first, Dg = sharding.slab_bounds(D, world, rank)
    else:
        Dg = args.disp_total or args.disp_per_gpu
        D = Dg * world
        d_first = rank * Dg
    left, right, _ = synth.scene_pair(W, H, min(D, 256), n_bands=b)

    ctx = capi.Context(local_rank)
    for nv in args.option:
        name, value = nv.split("=")
        ctx.set_option(name, int(value))
    # an explicit stream for the library's kernels and torch's collectives alike: the legacy null
    # stream synchronises implicitly with every blocking stream of the process, which neither the
    # overlap of the all-reduce with the next frame nor torch's helper threads want
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    extra = None
    if not multi and not args.no_extra and rank == 0:
        # the side measurements (configs 4 and 5 per GPU) run first: they also bring the device out of
        # its idle power state before the W warm-up steps
        extra = extra_configs(np, capi, synth, ctx)

    t_ref = torch.from_numpy(left.reshape(H, W * b)).to(dev)
    t_mtc = torch.from_numpy(right.reshape(H, W * b)).to(dev)
    # column-major key image; for the reduce-scatter a flat buffer padded to a multiple of the ranks (the padding
    # keeps the largest key: the kernels write the first W * H elements only)
    n_keys = sharding.padded_key_count(W * H, world) if multi else W * H
    keys = torch.full((n_keys,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    keys2 = torch.full_like(keys, torch.iinfo(torch.int64).max) if multi else None
    disp = torch.empty((H, W), dtype=torch.int32, device=dev)
    best = torch.empty((H, W), dtype=torch.float32, device=dev)
    p = capi.match_params(W, H, b, D, match_fn=1, match_interval=1, match_max=1000)
    calls_per_step = 1 if not multi else (Dg + max_call - 1) // max_call

    want = None
    if args.check_oracle and rank == 0:
        want = oracle_frame(np, left, right, D, win)
    frames_checked = [0]
    produce_events = []

    def produce(buf, _frame):
        if not multi:
            ctx.fused_box_wta_keys(t_ref.data_ptr(), t_mtc.data_ptr(), W * b, p, win, d_first, Dg, buf.data_ptr())
            return
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        if True:
            for i, s0 in enumerate(range(d_first, d_first + Dg, max_call)):
                n = min(max_call, d_first + Dg - s0)
                # the keys of one rank's calls merge like those of ranks: from the second call on pass B stores
                # min(old key, new key) (smhip_fused_box_wta_keys_merge) -- no pass over two key images
                fn = ctx.fused_box_wta_keys if i == 0 else ctx.fused_box_wta_keys_merge
                fn(t_ref.data_ptr(), t_mtc.data_ptr(), W * b, p, win, s0, n, buf.data_ptr())
        ev1.record()
        produce_events.append((ev0, ev1))

    def consume(buf, _frame):
        if buf is None:             # reduce-scatter mode: the merged keys exist on rank 0 only
            return
        ctx.keys_decode(buf.data_ptr(), W, H, disp.data_ptr(), best.data_ptr())
        if want is not None:
            got_d, got_b = disp.cpu().numpy(), best.cpu().numpy()
            if not (np.array_equal(got_d, want[0]) and np.array_equal(got_b.view(np.uint32), want[1].view(np.uint32))):
                raise SystemExit("frame %d differs from the oracle" % frames_checked[0])
            frames_checked[0] += 1

    # N > 1: the one exchange step (int64 MIN all-reduce over xGMI) overlaps the next frame
    pipe = sharding.KeyReducePipeline([keys, keys2], produce, consume, force_collective=(world == 1),
                                      host_staging=(args.backend != "nccl"),
                                      collective=args.collective) if multi else None

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

    if not multi and not args.check_oracle:
        # clock ramp: a cold GPU runs the first ~100 ms of launches below its sustained clocks (pass B 0.358
        # instead of 0.3375 ms), and W warm-up steps of 0.73 ms do not get it there; untimed, like the warm-up
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.3:
            for _ in range(20):
                step()
            fence()
    for _ in range(args.warmup):
        step()
    fence()
    del produce_events[:]
    profiling = not os.environ.get("BENCH_NO_PROFILE")       # rehearsal aid
    if profiling:
        ctx.profile_begin(args.steps * calls_per_step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    per_call = ctx.profile_get(0, args.steps * calls_per_step) if profiling else ([0.001], [0.001])
    n_prof, ms_a, ms_b = ctx.profile_end() if profiling else (1, 0.001, 0.001)
    compute_ms = (sum(e0.elapsed_time(e1) for e0, e1 in produce_events) / len(produce_events) if produce_events
                  else (ms_a + ms_b) / max(1, n_prof))

    sustained = None
    if not multi and not args.check_oracle and args.sustained_seconds > 0:
        # the same step for >= 2 s, after the timed region (the headline is the K steps above): what the part holds
        # once clocks and temperature have settled, with the spread of the per-launch times
        n_s = max(args.steps, int(args.sustained_seconds / (dt / args.steps)) + 1)
        if profiling:
            ctx.profile_begin(n_s)
        t0s = time.perf_counter()
        for _ in range(n_s):
            step()
        fence()
        dts = time.perf_counter() - t0s
        sustained = {"seconds": round(dts, 3), "steps": n_s, "ms_per_step": round(dts / n_s * 1e3, 4),
                     "value": round(W * H * D * n_s / dts / 1e9, 2), "unit": "G pixel*disparity/s"}
        if profiling:
            pa, pb = ctx.profile_get(0, n_s)
            ctx.profile_end()
            for name, xs in (("pass_a_ms", pa), ("pass_b_ms", pb)):
                xs = sorted(xs)
                sustained[name] = {"min": round(xs[0], 4), "median": round(xs[len(xs) // 2], 4),
                                   "p99": round(xs[int(len(xs) * 0.99)], 4), "max": round(xs[-1], 4)}

    t = torch.tensor([dt, compute_ms], dtype=torch.float64, device="cpu" if args.backend != "nccl" and multi else dev)
    per_rank_compute = [compute_ms]
    if multi:
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        dt = max(float(g[0]) for g in gathered)
        per_rank_compute = [round(float(g[1]), 4) for g in gathered]

    n_vox_gpu = W * H * Dg
    n_vox = W * H * D
    value = n_vox * args.steps / dt / 1e9
    ms_step = dt / args.steps * 1e3

    if rank == 0:
        # algorithmic bytes per launch: pass A writes the slab once and reads the
        # images; pass B reads the slab once and writes one 8-byte key per pixel.
        n_call = n_vox_gpu / calls_per_step             # voxels of one fused call
        n_bt = 3 if p.match_interval else 2          # ref + (mn, mx) or ref + match
        bytes_a = 4.0 * n_call + n_bt * b * W * H
        bytes_b = 4.0 * n_call + 8.0 * W * H
        bytes_fused = 8.0 * n_call + 2.0 * b * W * H + 8.0 * W * H
        ta, tb = ms_a / n_prof * 1e-3, ms_b / n_prof * 1e-3
        traffic, traffic_source = {}, None
        config3 = not multi and (W, H, Dg, b, win) == (1920, 1080, 256, 1, 9)
        if config3 and os.path.exists(args.traffic_json):    # counters were collected on config 3 only
            with open(args.traffic_json) as f:
                traffic = json.load(f)
            traffic_source = traffic.get("source", "rocprofv3 --pmc passes of an earlier run of this command "
                                                   "(profiles/README.md), not of this run")

        def stats(xs):
            xs = sorted(xs)
            return {"min": round(xs[0], 4), "median": round(xs[len(xs) // 2], 4), "max": round(xs[-1], 4)}

        def roof(name, nbytes, secs, per=None):
            ach = nbytes / secs / 1e9
            r = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                 "traffic": traffic.get(name), "avg_ms": round(secs * 1e3, 4),
                 "algorithmic_bytes": int(nbytes)}
            if per:
                r["ms_per_launch"] = stats(per)
            return r
        name_a = ctx.last_pass_a_kernel()            # the library names the pass A kernel its dispatcher took
        ka = roof(name_a, bytes_a, ta, per_call[0])
        kb = roof("fused_boxv_wta_vpl_kernel", bytes_b, tb, per_call[1])
        dominant = ka if ta >= tb else kb
        fused = roof("fused pair (pass A + pass B)", bytes_fused, ta + tb)

        if sharded:
            workload = ("fused BT-SAD raw cost + %dx%d box + PadCosts + WTA, %dx%d, b=%d, D=%d over %d GPUs by "
                        "disparity slab (%d levels per GPU in calls of <= %d), int64-MIN %s %s of the keys" % (
                            win, win, W, H, b, D, world, Dg, max_call,
                            "RCCL" if args.backend == "nccl" else args.backend + " (host-staged)",
                            "all-reduce" if args.collective == "allreduce" else "reduce-scatter + gather on rank 0"))
        else:
            workload = "fused BT-SAD raw cost + %dx%d box + PadCosts + WTA, %dx%d, b=%d, D=%d (%d levels per GPU%s)" % (
                win, win, W, H, b, D, Dg,
                ", int64-MIN %s %s of keys" % ("RCCL" if args.backend == "nccl" else args.backend, args.collective)
                if multi else "")
        out = {
            "metric": "cost-volume throughput (G pixel*disparity/s)",
            "value": round(value, 2), "unit": "G pixel*disparity/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workload,
                "width": W, "height": H, "bands": b, "disp_total": D, "disp_per_gpu": Dg,
                "window": win, "match_fn": "AD", "match_interval": 1, "match_max": 1000,
                "parallelism": "disparity slabs x%d" % world,
                "collective_backend": (args.backend if multi else None),
                "collective": (args.collective if multi else None)},
            "roofline": {k: dominant[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")},
            "roofline_kernel": dominant["kernel"],
            "traffic_source": traffic_source,
            "roofline_fused": fused,
            "kernels": [ka, kb],
        }
        if multi:
            out["per_rank_compute_ms"] = per_rank_compute
            out["exposed_collective_ms"] = round(ms_step - max(per_rank_compute), 4)
            # bytes a rank sends per frame: ring all-reduce 2 (N-1)/N of the key image; reduce-scatter (N-1)/N
            # plus its 1/N chunk to the root
            kb = n_keys * 8
            out["collective_bytes_per_step"] = kb
            out["collective_bytes_sent_per_rank"] = int(
                2 * kb * (world - 1) / world if args.collective == "allreduce" else kb * (world - 1) / world + kb / world)
        if args.check_oracle:
            out["frames_checked_vs_oracle"] = frames_checked[0]
        if extra is not None:
            out["extra"] = extra
        if sustained is not None:
            out.setdefault("extra", {})["sustained"] = sustained
        if not multi and not args.no_cpu_baseline:
            got = (disp.cpu().numpy(), best.cpu().numpy())         # the frame the timed loop left behind
            out["cpu_baseline"], ok = cpu_baseline(np, left, right, args, Dg, win, got=got)
            out["verified_vs_oracle"] = ok
            out["cpu_baseline_all_cores"], _ = cpu_baseline(np, left, right, args, Dg, win, openmp=True)
            if "extra" in out:
                out["extra"]["config4_cpu"] = cpu_config4(np, synth)
        print(json.dumps(out), file=json_out, flush=True)

    if multi:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def oracle_frame(np, left, right, D, win):
    from oracle import oracle as O
    O.use_openmp(True)
    h, w, b = left.shape
    p = O.make_params(w, h, b, disp_min=0, disp_max=D - 1, match_fn=1, match_interval=1, match_max=1000,
                      aggr_window_size=win)
    d, c = O.compute_correspondence(left, right, p)[:2]
    O.use_openmp(False)
    return d, c


def extra_configs(np, capi, synth, ctx):
    """BASELINE configs 4 and 5 (per-GPU share), a few repetitions each, reported next to the
    headline line (not part of `value`): iterated aggregation on a resident 3840x2160x512 DSI and
    the fused path on a 7680x4320 slab of 128 levels.  The iterated runs keep several iterations in LDS
    per pass over the DSI (temporal blocking), so "8 bytes per voxel and iteration" is not what they move:
    they are reported with the HBM bytes the PMC counters measured for their kernels at this very shape
    (profiles/r03_stage_ops_summary.json, scripts/prof_ops_pmc.sh), the rate those bytes give over the time
    measured here, and the share of SIMD cycles that had a vector instruction in flight.  Failures are
    reported, they do not stop the bench."""
    res = {}
    ops_prof = {}
    ops_prof_path = os.path.join(ROOT, "profiles", "r03_stage_ops_summary.json")
    if os.path.exists(ops_prof_path):
        with open(ops_prof_path) as f:
            ops_prof = json.load(f)

    def run_counters(kernel_fmt, n_iter, dt, maxk=4):
        """Launches of smhip_*_run for n_iter iterations (depths as multi_depth of smhip_api.hip splits
        them) x the counters of each depth's kernel."""
        left, passes, nbytes, valu_t, t = n_iter, (n_iter + maxk - 1) // maxk, 0, 0.0, 0.0
        while left > 0:
            k = (left + passes - 1) // passes
            c = ops_prof.get(kernel_fmt % k)
            if c is None:                       # a depth the profile run did not launch: the nearest one it did
                near = [ops_prof[kernel_fmt % j] for j in range(1, maxk + 1) if kernel_fmt % j in ops_prof]
                c = near[-1] if near else None
            if c is None:
                return {"measured_bytes": None}
            nbytes += c["measured_bytes"]
            valu_t += c["valu_active_pct"] * c["avg_us"]
            t += c["avg_us"]
            left, passes = left - k, passes - 1
        return {"measured_bytes": nbytes, "achieved_GBs": round(nbytes / dt / 1e9, 1),
                "frac_of_8TBs": round(nbytes / dt / 1e9 / HBM_PEAK_GBS, 3), "valu_active_pct": round(valu_t / t, 1),
                "counters_source": "profiles/r03_stage_ops_summary.json"}

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
        # kernels of the _run calls at this shape: both in the pixels-on-lanes form, the binomial up to 4 iterations
        # per pass, the diffusion up to 6 (smhip_api.hip)
        bk, dk = "smk::iter_px_kernel<0, %d, 32, 0, 0, false, 2>", "smk::iter_px_kernel<1, %d, 32, 0, 0, false, 2>"
        for name, iters, kfmt, maxk, fn in (
                ("binomial_x4", 4, bk, 4, lambda: ctx.convolve_14641_run(a, b, W, H, D, W * D, 4)),
                ("binomial_x8", 8, bk, 4, lambda: ctx.convolve_14641_run(a, b, W, H, D, W * D, 8)),
                ("diffusion_x10", 10, dk, 6, lambda: ctx.diffusion_run(a, b, None, W, H, D, W * D, 0.15, 0.0, 10)),
                ("diffusion_x50", 50, dk, 6, lambda: ctx.diffusion_run(a, b, None, W, H, D, W * D, 0.15, 0.0, 50))):
            dt = timed(fn, 2)
            res["config4_" + name] = {
                "workload": "3840x2160, D=512, %d iterations on the resident DSI" % iters,
                "ms": round(dt * 1e3, 3), "ms_per_iteration": round(dt * 1e3 / iters, 3),
                "G_voxel_iterations_per_s": round(N * iters / dt / 1e9, 1)}
            res["config4_" + name].update(run_counters(kfmt, iters, dt, maxk))
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
        dt_staged = timed(slab_step, 3)
        a.free()
        b.free()
        # the same through smhip_fused_iter_wta_keys: one kernel, no DSI (smhip_iter_px.h)
        dt = timed(lambda: ctx.fused_iter_wta_keys(d_ref, d_mtc, W, p, 2, 4, d0, n, keys), 5)
        res["config4_slab_pipeline"] = {
            "workload": "3840x2160, levels 256..319 of D=512 (one GPU's share of 8): raw costs + 4 binomial "
                        "iterations + PadCosts + WTA -> keys, fused (smhip_fused_iter_wta_keys: no DSI traffic, "
                        "VALU-bound)",
            "ms": round(dt * 1e3, 3), "G_pixel_disparity_per_s": round(W * H * n / dt / 1e9, 1),
            "ms_stage_calls": round(dt_staged * 1e3, 3)}
        for fn, name, its in ((2, "binomial", (4, 8)), (3, "diffusion", (10, 50))):
            for iters in its:
                dt = timed(lambda: [ctx.fused_iter_wta_keys(d_ref, d_mtc, W, p, fn, iters, s0, 256, keys)
                                    for s0 in (0, 256)], 2)
                res["config4_fused_%s_x%d" % (name, iters)] = {
                    "workload": "3840x2160, D=512 in two calls of 256 levels: raw costs + %d %s iterations + "
                                "PadCosts + WTA -> keys, fused (smhip_fused_iter_wta_keys)" % (iters, name),
                    "ms": round(dt * 1e3, 3), "G_pixel_disparity_per_s": round(N / dt / 1e9, 1)}
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


def cpu_baseline(np, left, right, args, D, win, openmp=False, got=None):
    """The CPU oracle (a port of the reference's single-threaded CPU path, bit-identical to it on
    the tsukuba anchors) on the top `cpu_rows` rows of the same image pair: one core (like the
    reference), or -- openmp=True -- the -fopenmp build of the same source on the host cores this
    process may use (OMP_NUM_THREADS)."""
    from oracle import oracle as O
    O.use_openmp(openmp)
    rows = min(args.cpu_rows or left.shape[0], left.shape[0])
    l, r = np.ascontiguousarray(left[:rows]), np.ascontiguousarray(right[:rows])
    h, w, b = l.shape
    p = O.make_params(w, h, b, disp_min=0, disp_max=D - 1, match_fn=1, match_interval=1,
                      match_max=1000, aggr_window_size=win)
    O.lib()
    cores = O.threads()
    t0 = time.perf_counter()
    res = O.compute_correspondence(l, r, p)
    dt = time.perf_counter() - t0
    O.use_openmp(False)
    verified = None
    if got is not None and rows == left.shape[0]:    # the device's frame against the one just computed here
        verified = bool(np.array_equal(got[0], res[0]) and
                        np.array_equal(got[1].view(np.uint32), res[1].view(np.uint32)))
    return {"value": round(w * h * D / dt / 1e9, 5), "unit": "G pixel*disparity/s", "cores": cores,
            "kind": "port",
            "sample": "top %d of %d rows of the same %dx%d D=%d pair (%.1f s wall); "
                      "oracle/sm_oracle.c, gcc -O2%s" % (
                          rows, left.shape[0], w, left.shape[0], D, dt,
                          " -fopenmp, rows / column blocks over %d threads" % cores if openmp
                          else ", single thread like the reference")}, verified


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
