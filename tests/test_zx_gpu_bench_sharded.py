"""The N > 1 code path of bench.py (config 5's strong scaling by disparity slab) end to end, as far as one
GPU allows: two fresh processes launched the way the driver launches them, both on cuda:0, the gloo backend
with the keys reduced through pinned host memory (gloo's own device-tensor path stalls when two ranks share a
GPU: scripts/gloo_probe.py), a reduced frame, two fused calls per rank (so the local key merge runs), 30
pipelined frames -- and rank 0 compares EVERY decoded frame with the CPU oracle (bench.py --check-oracle)."""
import json
import os
import subprocess
import sys

import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("collective", ["allreduce", "reduce_scatter"])
def test_two_ranks_share_one_gpu_every_frame_equals_the_oracle(collective):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_WATCHDOG="150")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(util.free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
           "--width", "960", "--height", "540", "--disp-total", "64", "--max-call-levels", "16",
           "--steps", "30", "--warmup", "2", "--check-oracle", "--no-extra", "--no-cpu-baseline",
           "--collective", collective]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["collective"] == collective
    assert out["frames_checked_vs_oracle"] == 32          # warm-up frames are checked as well
    assert out["config"]["disp_total"] == 64 and out["config"]["disp_per_gpu"] == 32
    assert len(out["per_rank_compute_ms"]) == 2


@pytest.mark.parametrize("collective", ["allreduce", "reduce_scatter"])
def test_one_rank_over_rccl_every_frame_equals_the_oracle(collective):
    """The same code path over the backend the driver uses -- RCCL (torch `nccl`) -- in a process group of one
    rank, which is all a one-GPU box can run of it: process-group set-up on the device, the asynchronous int64
    MIN all-reduce (or reduce-scatter + gather on rank 0) of the key image on the library's stream, the timing
    all-gather, four fused calls per frame merged by key minimum inside pass B (smhip_fused_box_wta_keys_merge);
    12 frames, every one compared with the oracle."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_WATCHDOG="150",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(util.free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--one-rank-rehearsal",
           "--width", "960", "--height", "540", "--disp-total", "64", "--max-call-levels", "16",
           "--steps", "10", "--warmup", "2", "--check-oracle", "--no-extra", "--no-cpu-baseline",
           "--collective", collective]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["collective_backend"] == "nccl" and out["scaling"] == "strong"
    assert out["config"]["collective"] == collective
    assert out["frames_checked_vs_oracle"] == 12
    assert out["collective_bytes_per_step"] == 960 * 540 * 8
