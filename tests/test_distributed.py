"""world_size-2 gloo tests (CPU) of the N>1 path: sharding covers every frame exactly once and the
single all-gather reassembles per-rank keypoint tensors in global frame order."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from object_keypoints_amd import distributed as d
    r, _, w = d.init(backend="gloo")
    start, count = d.shard(total, r, w)
    # fixed-capacity payload [frames, K, cap, 4]; value encodes the global frame index, NaN marks unused slots
    pts = torch.full((count, 3, 4, 4), float("nan"), dtype=torch.float64)
    for i in range(count):
        pts[i, :, :2, :] = float(start + i)
    gathered = d.all_gather_keypoints(pts, total_frames=total)
    slowest = d.max_over_ranks(float(rank + 1), torch.device("cpu"))
    d.barrier()
    q.put((rank, start, count, gathered.numpy(), slowest))


import pytest


@pytest.mark.parametrize("total", [8, 7])          # 7 frames on 2 ranks: blocks of 4 and 3 (padded for the collective, trimmed after)
def test_allgather_world2_gloo(total):
    from object_keypoints_amd import distributed as d
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = []
    for rank, start, count, gathered, slowest in results:
        covered += list(range(start, start + count))
        assert gathered.shape == (total, 3, 4, 4)
        assert np.array_equal(gathered[:, 0, 0, 0], np.arange(total, dtype=np.float64))   # global frame order
        assert np.isnan(gathered[:, :, 2:, :]).all()
        assert slowest == 2.0                                                              # max over ranks
    assert covered == list(range(total))
    assert d.shard(10, 0, 4) == (0, 3) and d.shard(10, 3, 4) == (8, 2)
    assert sum(d.shard(511, r, 8)[1] for r in range(8)) == 511


def test_single_process_is_passthrough():
    from object_keypoints_amd import distributed as d
    x = torch.zeros(2, 3, 4, 4)
    assert d.all_gather_keypoints(x) is x
    assert d.max_over_ranks(3.5, torch.device("cpu")) == 3.5
