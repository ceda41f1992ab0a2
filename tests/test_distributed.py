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


def _worker_cups(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from object_keypoints_amd import distributed as d
    r, _, w = d.init(backend="gloo")
    per_rank, K, cap = 64, 4, 64                   # BASELINE configs[3]: 64 frames per rank, K = 4 maps (config/cups.json), capacity 64
    start, count = d.shard(per_rank * w, r, w)
    pts = torch.full((count, K, cap, 4), float("nan"), dtype=torch.float64)
    for i in range(count):
        for k in range(K):
            n = 4 if k == 0 else 4 - (i + k) % 2           # four objects per frame: four centres, up to four keypoints of each type
            pts[i, k, :n, :3] = float(start + i) + 0.01 * k
            pts[i, k, :n, 3] = 0.5
    gathered = d.all_gather_keypoints(pts, total_frames=per_rank * w)
    d.barrier()
    q.put((rank, start, count, tuple(gathered.shape), gathered[:, :, 0, 0].numpy(), int(torch.isnan(gathered[:, 0, 4:, :]).all()), gathered.element_size() * gathered.numel()))


def test_allgather_world2_gloo_cups_payload():
    """The all-gather payload of `bench.py --workload cups512` (BASELINE configs[3]) at world size 2: [64 N, 4, cap, 4] fp64, every
    rank ends with all 128 frames in global order, unused slots NaN."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cups, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, start, count, shape, first, tail_nan, nbytes in results:
        assert (start, count) == (64 * rank, 64)
        assert shape == (128, 4, 64, 4) and nbytes == 128 * 4 * 64 * 4 * 8
        assert np.allclose(first, np.arange(128)[:, None] + 0.01 * np.arange(4)[None])      # frame order and map order survive the gather
        assert tail_nan == 1


def _worker_ws1(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from object_keypoints_amd import distributed as d
    assert d.launched_by_torchrun()
    r, _, w = d.init(backend="gloo")
    pts = torch.arange(2 * 3 * 4 * 4, dtype=torch.float64).reshape(2, 3, 4, 4)
    out = d.all_gather_keypoints(pts, total_frames=2)                 # a real collective on a one-rank group
    d.barrier()
    q.put((w, dist.is_initialized(), out is not pts, bool(torch.equal(out, pts)), d.max_over_ranks(1.25, torch.device("cpu"))))
    dist.destroy_process_group()


def test_torchrun_world_size_1_goes_through_the_process_group():
    """Under torchrun a single rank still creates its process group and calls the collectives (on the GPU box: RCCL) - the
    one-GPU rehearsal of the 8-GPU path (bench.py --gpus 1 --spawn)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_ws1, args=(_free_port(), q))
    p.start()
    world, initialised, copied, equal, slowest = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert world == 1 and initialised and copied and equal and slowest == 1.25


def _worker_missing_total(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from object_keypoints_amd import distributed as d
    d.init(backend="gloo")
    try:
        d.all_gather_keypoints(torch.zeros(2 + rank, 3, 4, 4, dtype=torch.float64))      # uneven blocks, no total_frames
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, str(e)))
    d.barrier()


def test_uneven_blocks_without_total_frames_are_refused_before_the_collective():
    """Ranks with different block sizes must never reach all_gather_into_tensor (a hang on RCCL): with more than one rank
    total_frames is mandatory and every rank raises locally."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_missing_total, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("total_frames" in got[r] and "required" in got[r] for r in range(world))


def test_single_process_is_passthrough():
    from object_keypoints_amd import distributed as d
    x = torch.zeros(2, 3, 4, 4)
    assert d.all_gather_keypoints(x) is x
    assert d.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def _worker_world8(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from object_keypoints_amd import distributed as d
    r, _, w = d.init(backend="gloo")
    K, cap = 4, 16                                  # config/cups.json: K = 4 maps
    start, count = d.shard(total, r, w)
    pts = torch.full((count, K, cap, 4), float("nan"), dtype=torch.float64)
    for i in range(count):
        for k in range(K):
            pts[i, k, :1 + (start + i + k) % 3, :] = float(1000 * (start + i) + k)
    gathered = d.all_gather_keypoints(pts, total_frames=total)
    slowest = d.max_over_ranks(float(10 + rank), torch.device("cpu"))
    d.barrier()
    q.put((rank, start, count, tuple(gathered.shape), gathered[:, :, 0, 0].numpy(), slowest))


@pytest.mark.parametrize("total", [64, 61])        # 8 ranks x 8 frames, and an uneven total (blocks of 8, 8, 8, 8, 8, 7, 7, 7)
def test_allgather_world8_gloo_cups_payload(total):
    """The rehearsal of the driver's 8-GPU run (BASELINE configs[3]): world size 8, the cups payload [frames, 4, cap, 4] fp64,
    every rank ends with all frames in global order; blocks that differ by a frame are padded for the collective and trimmed after it."""
    from object_keypoints_amd import distributed as d
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_world8, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    covered = []
    want = np.array([[1000.0 * f + k for k in range(4)] for f in range(total)])
    for rank, start, count, shape, first, slowest in results:
        covered += list(range(start, start + count))
        assert (start, count) == d.shard(total, rank, world) and shape == (total, 4, 16, 4)
        assert np.array_equal(first, want) and slowest == 17.0
    assert covered == list(range(total))
    sizes = [d.shard(61, r, 8)[1] for r in range(8)]
    assert sizes == [8, 8, 8, 8, 8, 7, 7, 7] and max(sizes) - min(sizes) == 1
