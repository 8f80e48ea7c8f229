"""The N > 1 path on CPU: slab partition, key order, and a world_size-2 gloo MIN all-reduce
that must reproduce the single-process WTA of the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from stereomatch_amd import sharding, synth


def test_slab_bounds_cover_every_level_once():
    for D in (2, 16, 100, 256, 1024):
        for world in (1, 2, 3, 8):
            if world > D:
                continue
            seen = []
            for r in range(world):
                d0, n = sharding.slab_bounds(D, world, r)
                assert n >= D // world
                seen += list(range(d0, d0 + n))
            assert seen == list(range(D))
    with pytest.raises(ValueError):
        sharding.slab_bounds(4, 8, 0)


def test_key_order_matches_float_order_and_tie_break():
    c = np.array([0.0, -0.0, 1.5, -2.25, 3e-39, -3e-39, 260100.0, 1e30, -1e30], np.float32)
    s = sharding.ordered_bits(c)
    for i in range(len(c)):
        for j in range(len(c)):
            assert (c[i] < c[j]) == (s[i] < s[j]), (c[i], c[j])
    k = sharding.encode_keys(np.float32([5.0, 5.0, 7.0]), [9, 3, 0])
    assert np.argmin(k) == 1                       # equal cost -> lowest d
    d, b = sharding.decode_keys(k)
    assert list(d) == [9, 3, 0] and list(b) == [5.0, 5.0, 7.0]
    d, b = sharding.decode_keys(sharding.encode_keys(c, np.arange(len(c))))
    assert np.array_equal(b[2:], c[2:]) and b[1] == 0.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cost_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cost = np.load(cost_path)                       # aggregated + padded DSI [H, W, D]
    D = cost.shape[2]
    d0, n = sharding.slab_bounds(D, world, rank)
    slab = cost[:, :, d0:d0 + n]
    local = np.argmin(slab, axis=2)                 # first minimum inside the slab
    best = np.take_along_axis(slab, local[..., None], axis=2)[..., 0]
    keys = torch.from_numpy(sharding.encode_keys(best, local + d0).T.copy())   # column-major
    sharding.all_reduce_min_keys(keys)
    if rank == 0:
        np.save(out_path, keys.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_slab_merge_equals_oracle_wta(oracle, tmp_path, world):
    w, h, D = 72, 20, 24
    left, right = synth.noise_pair(w, h, 1)
    p = oracle.make_params(w, h, 1, disp_max=D - 1, match_fn=1, match_max=2, aggr_window_size=5)
    cost = oracle.aggregate(oracle.raw_costs(left, right, p), p)     # many ties (match_max 2)
    want_d, want_b = oracle.wta(cost)
    cp, op = str(tmp_path / "cost.npy"), str(tmp_path / "keys.npy")
    np.save(cp, cost)
    mp.spawn(_worker, args=(world, _free_port(), cp, op), nprocs=world, join=True)
    d, b = sharding.decode_keys(np.load(op).T)
    assert np.array_equal(d, want_d)
    util.assert_bit_equal(b, want_b, "merged best cost")


def _pipeline_worker(rank, world, port, cost_path, out_path, n_frames, collective="allreduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cost = np.load(cost_path)                       # [F, H, W, D]: one DSI per frame
    D = cost.shape[3]
    d0, n = sharding.slab_bounds(D, world, rank)
    got, log = {}, []

    h, w = cost.shape[1:3]

    def produce(buf, f):
        slab = cost[f][:, :, d0:d0 + n]
        local = np.argmin(slab, axis=2)
        best = np.take_along_axis(slab, local[..., None], axis=2)[..., 0]
        buf.view(-1)[:w * h].copy_(torch.from_numpy(sharding.encode_keys(best, local + d0).T.copy()).view(-1))
        log.append(("p", f))

    def consume(buf, f):
        if collective == "reduce_scatter":          # merged keys on rank 0 only
            assert (buf is None) == (rank != 0)
        if buf is not None:
            got[f] = buf.view(-1)[:w * h].view(w, h).numpy().copy()
        log.append(("c", f))

    if collective == "reduce_scatter":              # flat buffers padded to a multiple of the ranks
        n_keys = sharding.padded_key_count(w * h, world)
        bufs = [torch.full((n_keys,), np.iinfo(np.int64).max, dtype=torch.int64) for _ in range(2)]
    else:
        bufs = [torch.empty((w, h), dtype=torch.int64) for _ in range(2)]
    pipe = sharding.KeyReducePipeline(bufs, produce, consume, collective=collective)
    for f in range(n_frames):
        pipe.submit(f)
    pipe.flush()
    # frame f is consumed after frame f + 1 was produced (the overlap), the last one at flush
    want = [("p", 0)]
    for f in range(1, n_frames):
        want += [("p", f), ("c", f - 1)]
    want.append(("c", n_frames - 1))
    assert log == want, log
    if rank == 0:
        np.save(out_path, np.stack([got[f] for f in range(n_frames)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("collective,world,w", [("allreduce", 2, 40), ("reduce_scatter", 2, 40), ("reduce_scatter", 3, 41)])
def test_gloo_key_reduce_pipeline(oracle, tmp_path, collective, world, w):
    """KeyReducePipeline (double-buffered, the reduction of frame i under the compute of frame
    i + 1) returns, for every frame, the oracle's WTA over all levels -- as one MIN all-reduce, or as a MIN
    reduce-scatter whose chunks are gathered on rank 0 (41 x 12 keys over 3 ranks: a padded buffer)."""
    h, D, F = 12, 16, 5
    costs, want = [], []
    for f in range(F):
        left, right = synth.noise_pair(w, h, 1, seed=synth.SEED_NOISE + f)
        p = oracle.make_params(w, h, 1, disp_max=D - 1, match_fn=1, match_max=3, aggr_window_size=3)
        c = oracle.aggregate(oracle.raw_costs(left, right, p), p)
        costs.append(c)
        want.append(oracle.wta(c))
    cp, op = str(tmp_path / "costs.npy"), str(tmp_path / "keys.npy")
    np.save(cp, np.stack(costs))
    mp.spawn(_pipeline_worker, args=(world, _free_port(), cp, op, F, collective), nprocs=world, join=True)
    keys = np.load(op)
    for f in range(F):
        d, b = sharding.decode_keys(keys[f].T)
        assert np.array_equal(d, want[f][0])
        util.assert_bit_equal(b, want[f][1], "frame %d best cost" % f)


def _staged_worker(rank, world, port, img_path, out_path, seq, kw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    imgs = np.load(img_path)
    left, right = imgs[0], imgs[1]
    h, w, b = left.shape
    whole = O.make_params(w, h, b, **seq, **kw)
    d0, n = sharding.slab_bounds(whole.disp_n, world, rank)
    # the rank's slab as a DSI of its own: first level at disp_min + d0 * step, n levels -- raw costs,
    # aggregation and PadCosts never look at another level (what matcher.ComputeSlabKeys relies on)
    step = seq.get("disp_step", 1.0)
    mine = dict(seq, disp_min=seq["disp_min"] + int(d0 * step), disp_max=seq["disp_min"] + int((d0 + n - 1) * step))
    p = O.make_params(w, h, b, **mine, **kw)
    assert p.disp_n == n
    cost = O.aggregate(O.raw_costs(left, right, p), p)
    local, best = O.wta(cost)
    keys = torch.from_numpy(sharding.encode_keys(best, local + d0).T.copy())
    sharding.all_reduce_min_keys(keys)
    if rank == 0:
        np.save(out_path, keys.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kw", [
    (2, dict(aggr_fn=3, aggr_iter=3, aggr_minfilter=3, match_fn=2, match_max=20)),
    (3, dict(aggr_fn=2, aggr_iter=2, match_interval=1, match_max=4)),
    (2, dict(aggr_fn=4, aggr_iter=4, diff_beta=0.25, match_fn=1, match_max=1000)),
])
def test_gloo_staged_slabs_equal_whole_volume(oracle, tmp_path, world, kw):
    """Every rank aggregates only its own levels (separate raw costs, iterated aggregation, shiftable
    window, PadCosts on a DSI that starts at its first level); the merged keys equal WTA on the whole
    volume -- the sharding rule of the staged pipeline, on the oracle."""
    w, h = 64, 18
    seq = dict(disp_min=-2, disp_max=21, disp_step=1.0, frame_ref=1, frame_match=0)
    left, right, _ = synth.scene_pair(w, h, 12, n_bands=4, seed=5)
    p = oracle.make_params(w, h, 4, **seq, **kw)
    want_d, want_b = oracle.wta(oracle.aggregate(oracle.raw_costs(left, right, p), p))
    ip, op = str(tmp_path / "imgs.npy"), str(tmp_path / "keys.npy")
    np.save(ip, np.stack([left, right]))
    mp.spawn(_staged_worker, args=(world, _free_port(), ip, op, seq, kw), nprocs=world, join=True)
    d, b = sharding.decode_keys(np.load(op).T)
    assert np.array_equal(d, want_d)
    util.assert_bit_equal(b, want_b, "merged best cost")
