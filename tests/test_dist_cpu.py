"""N > 1 path on CPU: world_size-2 gloo run of the shard + all-gather logic (dist.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from qtos_amd import dist as qd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, n = 7, 5  # ragged: 4 + 3
    start = np.arange(B * 24, dtype=float).reshape(B, 24)
    goal = np.arange(B * 3, dtype=float).reshape(B, 3)

    def fake_solve(s, g):  # stands in for the GPU solve: nodes encode which problem they belong to
        return s[:, :n] * 2.0 + g[:, :1], (s[:, 0] % 2).astype(np.int32)

    nodes, status = qd.plan_sharded(fake_solve, start, goal)
    np.save(os.path.join(tmp, "nodes%d.npy" % rank), nodes)
    np.save(os.path.join(tmp, "status%d.npy" % rank), status)
    # the same gather batch after batch through reused buffers (bench.py's N > 1 path)
    import torch
    b, e = qd.shard_bounds(B, world, rank)
    work = qd.gather_buffers(B, n, world, torch.float64, "cpu")
    ok = True
    for rep in range(3):
        loc_n, loc_s = fake_solve(start[b:e] + rep, goal[b:e])
        an, as_ = qd.gather_plans(torch.as_tensor(loc_n), torch.as_tensor(loc_s), B, work=work)
        wn, ws = fake_solve(start + rep, goal)
        ok = ok and np.array_equal(an.numpy(), wn) and np.array_equal(as_.numpy(), ws)
    np.save(os.path.join(tmp, "reuse%d.npy" % rank), np.array([ok]))
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    from qtos_amd.dist import shard_bounds
    for B in (1, 7, 256, 2048, 2049):
        for W in (1, 2, 3, 8):
            spans = [shard_bounds(B, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_world_size_2_gloo_allgather(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    B, n = 7, 5
    start = np.arange(B * 24, dtype=float).reshape(B, 24)
    goal = np.arange(B * 3, dtype=float).reshape(B, 3)
    want_nodes = start[:, :n] * 2.0 + goal[:, :1]
    want_status = (start[:, 0] % 2).astype(np.int32)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("nodes%d.npy" % r)), want_nodes)
        assert np.array_equal(np.load(tmp_path / ("status%d.npy" % r)), want_status)
        assert np.load(tmp_path / ("reuse%d.npy" % r)).all()


def _settle_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import time
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a "step" holds a collective, as bench.py's does for N > 1; rank 0's steps settle at once, rank 1's take three steps longer
    durations = [0.010] * 12 if rank == 0 else [0.030, 0.022, 0.016, 0.010, 0.010, 0.010, 0.010, 0.010, 0.010, 0.010, 0.010, 0.010]
    count = {"n": 0}

    def step():
        time.sleep(durations[count["n"]])
        count["n"] += 1
        t = torch.ones(1)
        dist.all_reduce(t)            # (ranks that ran different numbers of steps would pair this with the wrong collective)
        assert float(t.item()) == world

    def agree(done):
        f = torch.tensor([1 if done else 0], dtype=torch.int32)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item())
    n = bench.settle(step, dist.barrier, 10, 0.2, agree)
    np.save(os.path.join(tmp, "settle%d.npy" % rank), np.array([n, count["n"]]))
    dist.destroy_process_group()


def test_adaptive_warmup_runs_the_same_number_of_steps_on_every_rank(tmp_path):
    """bench.py's adaptive warm-up (settle) with two ranks over gloo: a step holds a collective, so the decision to stop is the
    ranks' AND after every step -- the rank whose steps are steady from the start keeps stepping until the other one has settled."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_settle_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "settle0.npy"), np.load(tmp_path / "settle1.npy")
    assert a[0] == b[0] == a[1] == b[1] and 4 <= a[0] <= 8, (a, b)
