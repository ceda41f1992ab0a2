"""N > 1 on the hardware a one-GPU box has: TWO processes of the HIP path at once on the one MI355X, frames sharded with
distributed.shard(128, r, 2), the fixed-capacity 3D-keypoint tensor of each rank gathered by ONE all_gather_keypoints call
(gloo carries the collective here - two ranks cannot share one device in an RCCL communicator; the collective itself on RCCL is
exercised by `bench.py --gpus 1 --spawn`).  The gathered [128, K, cap, 4] tensor must equal the single-process result row for row:
frames are independent and the network is batch-invariant bit for bit, so the shard boundary must not show.

The reference has no counterpart (perception/pipeline.py:183 asserts one frame per call); SURVEY.md 8(e)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

TOTAL, WORLD, CAP = 128, 2, 256


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frames(start, count, dev):
    """Frames start .. start + count of the global sequence: seeded per GLOBAL frame index, generated on the device."""
    out = torch.empty((count, 3, 511, 511), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    for i in range(count):
        gen.manual_seed(4242 + start + i)
        out[i] = torch.randn((3, 511, 511), generator=gen, device=dev, dtype=torch.float32)
    return out


def _pipeline(dev, k=3, config=(1, 3)):
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.models import KeypointNet
    from object_keypoints_amd.perception.pipeline import BatchedKeypointPipeline
    from object_keypoints_amd.perception.utils import camera_utils as cu
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = KeypointNet(features=128, heatmaps_out=k, compute_dtype=torch.bfloat16)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
    net.eval().to(dev)
    params = cu.load_calibration_params(os.path.join(repo, "config", "calibration.yaml"))
    camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
    camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
    return BatchedKeypointPipeline(net, {"keypoint_config": list(config)}, camera, capacity=CAP)


def _points(pipe, frames):
    """Network -> peak-NMS -> per-peak 3D points on the network's OWN maps (random weights: flat heat maps with ~100 peaks each, more
    centre peaks than okp_group_objects groups - the grouping stage is not part of the all-gathered payload and is left out here)."""
    from object_keypoints_amd import ops
    with torch.no_grad():
        heat, depth, _ = pipe.net.deployed(frames)
        count, _, xyc = ops.peak_nms(heat, cap=CAP)
        points = ops.lift_peaks(pipe.cam, count, xyc, depth, int(pipe.max_index[0]), int(pipe.max_index[1]))
    return points, count


def _rank(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from object_keypoints_amd import _lib, distributed as d
    r, _, w = d.init(backend="gloo")                     # (the gloo group; the compute below is the HIP path on cuda:0)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    start, count = d.shard(TOTAL, r, w)
    pipe = _pipeline(dev)
    points, cnt = _points(pipe, _frames(start, count, dev))
    torch.cuda.synchronize()
    d.barrier()                                          # both ranks have had their launches on the device at the same time
    gathered = d.all_gather_keypoints(points.cpu(), total_frames=TOTAL)
    q.put((rank, start, count, gathered.numpy() if rank == 0 else None, int(cnt.sum()), bool((cnt > CAP).any()),
           os.path.basename(_lib.LIB_PATH)))
    d.barrier()


def test_two_ranks_on_one_gpu_gather_what_one_process_computes():
    ctx = mp.get_context("spawn")                        # fresh interpreters: the children initialise HIP themselves
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    try:
        # meanwhile, the single-process result in this process: the same 128 frames as two passes of 64
        dev = torch.device("cuda", 0)
        pipe = _pipeline(dev)
        single = []
        for start in range(0, TOTAL, 64):
            single.append(_points(pipe, _frames(start, 64, dev))[0].cpu().numpy())
        single = np.concatenate(single)
        results, waited = [], 0
        while len(results) < WORLD:
            try:
                results.append(q.get(timeout=5))
            except Exception:                            # queue.Empty: a rank that died will never answer
                waited += 5
                assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
                assert waited < 900, "ranks did not answer"
        results.sort(key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()                                 # (this exact child, by handle)
    covered = []
    for rank, start, count, gathered, n_peaks, overflow, lib in results:
        covered += list(range(start, start + count))
        assert not overflow and n_peaks >= count and lib == "libokp_hip.so"
    assert covered == list(range(TOTAL))
    gathered = results[0][3]
    assert gathered.shape == (TOTAL, 3, CAP, 4) == single.shape
    assert np.array_equal(gathered, single, equal_nan=True)          # row for row, bit for bit (NaN = unused slot)
    assert np.isfinite(gathered[:, :, 0, :]).all()                   # every map of every frame holds at least one peak


# ---- the rehearsal of the driver's 8-GPU run: EIGHT ranks (one process each, all on this box's one GPU), config/cups.json (K = 4 maps),
# ---- an uneven total: 61 frames -> blocks of 8, 8, 8, 8, 8, 7, 7, 7, padded for the collective and trimmed after it ----------------------
TOTAL8, WORLD8 = 61, 8


def _rank8(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from object_keypoints_amd import distributed as d
    r, _, w = d.init(backend="gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    start, count = d.shard(TOTAL8, r, w)
    pipe = _pipeline(dev, k=4, config=(1, 1, 1))
    points, cnt = _points(pipe, _frames(start, count, dev))
    torch.cuda.synchronize()
    d.barrier()
    gathered = d.all_gather_keypoints(points.cpu(), total_frames=TOTAL8)
    slowest = d.max_over_ranks(float(rank), torch.device("cpu"))
    q.put((rank, start, count, gathered.numpy() if rank == 0 else tuple(gathered.shape), slowest, bool((cnt > CAP).any())))
    d.barrier()


def test_eight_ranks_cups_uneven_total_gather_what_one_process_computes():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank8, args=(r, WORLD8, port, q)) for r in range(WORLD8)]
    for p in procs:
        p.start()
    try:
        dev = torch.device("cuda", 0)
        pipe = _pipeline(dev, k=4, config=(1, 1, 1))
        single = _points(pipe, _frames(0, TOTAL8, dev))[0].cpu().numpy()
        results, waited = [], 0
        while len(results) < WORLD8:
            try:
                results.append(q.get(timeout=5))
            except Exception:
                waited += 5
                assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
                assert waited < 1200, "ranks did not answer"
        results.sort(key=lambda t: t[0])
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    covered = []
    for rank, start, count, gathered, slowest, overflow in results:
        covered += list(range(start, start + count))
        assert count == (8 if rank < 5 else 7) and slowest == 7.0 and not overflow
        if rank:
            assert gathered == (TOTAL8, 4, CAP, 4)
    assert covered == list(range(TOTAL8))
    gathered = results[0][3]
    assert gathered.shape == (TOTAL8, 4, CAP, 4) == single.shape
    assert np.array_equal(gathered, single, equal_nan=True)          # the seven shard boundaries do not show; padded rows are trimmed
