"""Parity at the benchmark's own shapes and precision (BASELINE configs[2]: 64 frames, bf16) - the launches the
frames/s figure is made of: whole-network frame independence and run-to-run bit-reproducibility with the hourglass
side streams on, the streaming fire kernel and the fused heads at N=64 against the oracle on a frame sample, and
hipGraph replays against the eager step."""
import os

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALIB = os.path.join(REPO, "config", "calibration.yaml")


def _need(dtype):
    from object_keypoints_amd import ops
    if dtype not in ops._DTYPES:
        pytest.skip(f"{dtype} is not built into this library")


def _net(dtype, k=3, seed=0):
    from object_keypoints_amd import synth
    _need(dtype)
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=k, compute_dtype=dtype)
    shapes = {kk: tuple(v.shape) for kk, v in net.state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=seed)
    net.load_state_dict({kk: torch.from_numpy(np.array(v)) for kk, v in vals.items()})
    return net.eval().cuda()


def _frames(n, seed=77):
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    return torch.randn((n, 3, 511, 511), generator=gen, device="cuda", dtype=torch.float32)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_batch64_frame_independence_and_side_stream_determinism(dtype):
    """(a) two eager runs of the 64-frame step with the side streams on are bit-equal; (b) a frame gives the same bits
    wherever it sits in the batch (the batch reversed); (c) frames 0, 17, 63 give the bits they give when run alone."""
    from object_keypoints_amd import ops
    net = _net(dtype)
    x = _frames(64)
    assert ops.SIDE_STREAMS
    with torch.no_grad():
        a = [t.clone() for t in net.deployed(x)]
        b = net.deployed(x)
        torch.cuda.synchronize()
        for u, v in zip(a, b):
            assert torch.equal(u, v)                                  # (a)
        r = net.deployed(torch.flip(x, dims=[0]).contiguous())
        for u, v in zip(a, r):
            assert torch.equal(u, torch.flip(v, dims=[0]))            # (b)
        for i in (0, 17, 63):
            s = net.deployed(x[i:i + 1])
            for u, v in zip(a, s):
                assert torch.equal(u[i:i + 1], v), f"frame {i} differs between batch 64 and batch 1"    # (c)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("h,stride", [(64, 1), (32, 1), (64, 2)])
def test_streaming_fire_kernel_at_bench_shapes(h, stride, dtype):
    """okp_fire2 <256,128> exactly as the bench launches it (N=64; 64x64 and 32x32 at stride 1, 64x64 -> 32x32 at stride 2;
    XCD-aware tile order) against the oracle's fire_module on frames 0, 31 and 63."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    _need(dtype)
    o = onet.load_synthetic(onet.fire_module(256, 256, stride=stride), seed=21)
    m = bb.fire_module(256, 256, stride=stride)
    m.load_state_dict(o.state_dict())
    m.eval()
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = torch.randn((64, h, h, 256), generator=gen, device="cuda").to(dtype)
    l0 = ops.COUNTERS["launches"]
    got = m(ops.Act(x))
    assert ops.COUNTERS["launches"] - l0 == 1
    sample = [0, 31, 63]
    with torch.no_grad():
        ref = o(x[sample].float().permute(0, 3, 1, 2).cpu())
    g = got.t[sample].float().permute(0, 3, 1, 2).cpu()
    scale = float(ref.abs().max())
    eps = 0.03 if dtype == torch.bfloat16 else 0.004
    assert float((g - ref).abs().max()) <= eps * scale + eps
    again = m(ops.Act(x))
    assert torch.equal(again.t, got.t)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_heads_at_bench_batch(dtype):
    """okp_heads at N=64 (the three workgroups of a tile on one XCD) against the three-launch path, all 64 frames."""
    from object_keypoints_amd import ops
    net = _net(dtype, seed=3)
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    cnv = ops.Act((torch.randn((64, 64, 64, 256), generator=gen, device="cuda") * 0.7).to(dtype))
    keep = ops.FUSE_HEADS
    try:
        ops.FUSE_HEADS = True
        fused = net._run_heads(1, cnv, sigmoid=True)
        ops.FUSE_HEADS = False
        ref = net._run_heads(1, cnv, sigmoid=True)
    finally:
        ops.FUSE_HEADS = keep
    for a, b in zip(fused, ref):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-2 * scale + 2e-3


@pytest.mark.parametrize("side", [False, True])
def test_graph_replay_equals_eager_at_batch64(side):
    """hipGraph replays of the 64-frame bf16 step are bit-equal to the eager step, five replays in a row, with the
    hourglass branches captured serially (default) and forked onto side streams."""
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import pipeline as op
    net = _net(torch.bfloat16)
    cam_o = op.eval_camera(CALIB)
    pipe = pp.BatchedKeypointPipeline(net, {"keypoint_config": [1, 3]}, cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size), capacity=128)
    x0, x1 = _frames(64, seed=1), _frames(64, seed=2)
    keep = pp.GRAPH_SIDE_STREAMS
    pp.GRAPH_SIDE_STREAMS = side
    try:
        with torch.no_grad():
            step = pipe.capture(x0)
            eager = pipe.forward_device(x1)
            for _ in range(5):
                out = step.replay(x1)
                torch.cuda.synchronize()
                for key in ("heat", "depth", "centers", "count", "xyc"):
                    assert torch.equal(out[key], eager[key]), key
    finally:
        pp.GRAPH_SIDE_STREAMS = keep
    # the captured step pins its plans and refuses to replay after the weights changed
    net.load_state_dict(net.state_dict())
    with pytest.raises(pp.OkpError):
        step.replay(x1)
