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
    if isinstance(dtype, str):
        return
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


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, "float32x3", "float32mix"])
def test_batch64_frame_independence_and_side_stream_determinism(dtype):
    """(a) two eager runs of the 64-frame step with the side streams on are bit-equal; (b) a frame gives the same bits
    wherever it sits in the batch (the batch reversed); (c) frames 0, 17, 63 give the bits they give when run alone (split-product
    configurations: the values, to summation noise)."""
    from object_keypoints_amd import ops
    net = _net(dtype)
    x = _frames(64)                     # (the fp32-storage configurations run it as two passes of 32 frames: the 2 GiB view limit)
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
        # (c) 16-bit configurations: every tile of a plan sums in the same order, so a frame's bits do not depend on the batch size.
        # Split-product configurations: the patch-resident kernel (16x16x32 MFMAs; the heuristic's choice at batch 64) and the gather
        # tiles (32x32x16; batch 1) add the same products in another order - the same frame agrees to summation noise, not bit for bit
        # (float32mix: that noise passes through fp16 rounding points, where 1e-6 upstream flips a rounding: its own measured frame-to-model
        #  scatter, tests/precision/bounds.py, is the scale)
        split = isinstance(dtype, str)
        tol = 5e-4 if dtype == "float32mix" else 2e-5
        for i in (0, 17, 63):
            s = net.deployed(x[i:i + 1])
            for u, v in zip(a, s):
                if split:
                    assert float((u[i:i + 1] - v).abs().max()) <= tol * (1.0 + float(v.abs().max())), f"frame {i} differs between batch 64 and batch 1"
                else:
                    assert torch.equal(u[i:i + 1], v), f"frame {i} differs between batch 64 and batch 1"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c,co,h,stride", [(256, 256, 64, 1), (256, 256, 32, 1), (256, 256, 64, 2),
                                            (384, 384, 16, 1), (384, 256, 16, 1), (384, 384, 16, 2), (384, 384, 8, 1), (512, 512, 8, 1), (512, 384, 8, 1)])
def test_streaming_fire_kernel_at_bench_shapes(c, co, h, stride, dtype):
    """okp_fire2 exactly as the bench launches it at N=64 (XCD-aware tile order) against the oracle's fire_module on frames 0, 31
    and 63: <256,128> at 64x64 and 32x32 (stride 1, skip values of the depth-wise branch from the LDS ring) and 64x64 -> 32x32
    (stride 2); the wide instances of the lower levels, whose squeeze weights stream through the per-wave LDS ring (384 -> 192 at
    16x16 / 8x8 and at stride 2, 512 -> 256 and 512 -> 192 at 8x8) or through three register sets (384 -> 128)."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    _need(dtype)
    o = onet.load_synthetic(onet.fire_module(c, co, stride=stride), seed=21)
    m = bb.fire_module(c, co, stride=stride)
    m.load_state_dict(o.state_dict())
    m.eval()
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = torch.randn((64, h, h, c), generator=gen, device="cuda").to(dtype)
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
@pytest.mark.parametrize("c,co,n,h,w", [(256, 256, 2, 16, 16),      # one round of the grid: 2 x 8 tiles (okp_fire2's launcher), second pixel block empty
                                        (256, 256, 64, 22, 32),     # 3 x 8 tiles, the last tile row holds two of its three rows
                                        (256, 384, 64, 32, 32),     # the 256 -> 192 instance (the 32 x 32 -> 16 x 16 module of the hourglass)
                                        (384, 512, 5, 16, 48),      # 384 -> 256, three tiles per row
                                        (384, 384, 7, 18, 16)])     # 384 -> 192 on a 9 x 8 output map
def test_stride2_fire_on_the_matrix_pipe(c, co, n, h, w, dtype):
    """The stride-2 instances of okp_fire2 with the depth-wise branch on the matrix pipe (output maps a multiple of 8 wide: tiles of 3 x 8 or
    2 x 8 output pixels, pixel block = 2 rows x 8 columns) against the oracle's fire_module: ragged last tile rows, the one-round tile choice,
    every (cin, mid) instance; one launch, deterministic."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    _need(dtype)
    o = onet.load_synthetic(onet.fire_module(c, co, stride=2), seed=33)
    m = bb.fire_module(c, co, stride=2)
    m.load_state_dict(o.state_dict())
    m.eval()
    gen = torch.Generator(device="cuda"); gen.manual_seed(6)
    x = torch.randn((n, h, w, c), generator=gen, device="cuda").to(dtype)
    l0 = ops.COUNTERS["launches"]
    got = m(ops.Act(x))
    assert ops.COUNTERS["launches"] - l0 == 1
    assert tuple(got.t.shape) == (n, (h + 1) // 2, (w + 1) // 2, co)
    sample = sorted({0, n // 2, n - 1})
    with torch.no_grad():
        ref = o(x[sample].float().permute(0, 3, 1, 2).cpu())
    g = got.t[sample].float().permute(0, 3, 1, 2).cpu()
    scale = float(ref.abs().max())
    eps = 0.03 if dtype == torch.bfloat16 else 0.004
    assert float((g - ref).abs().max()) <= eps * scale + eps
    assert torch.equal(m(ops.Act(x)).t, got.t)
    # a frame's values do not depend on the batch it is in (the tile shape may: 2 x 8 against 3 x 8 tiles)
    one = m(ops.Act(x[n - 1:n].contiguous()))
    assert torch.equal(one.t[0], got.t[n - 1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", ["innermost_level", "pair_384", "single_384"])
def test_resident_chains_at_bench_batch(form, dtype):
    """okp_fire_chain at N=64 as the bench launches it: the innermost hourglass level in one launch (stride-2 fire(384, 512) from 8x8,
    six fire(512, 512), fire(512, 384): entry / exit form) and a pair of fire(384, 384) at 8x8 - against the oracle's modules on frames
    0, 31, 63, bit-equal to the single-frame launch for those frames (one workgroup per frame) and run to run."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    _need(dtype)
    shapes = [(384, 512, 2)] + [(512, 512, 1)] * 6 + [(512, 384, 1)] if form == "innermost_level" else [(384, 384, 1)] * (2 if form == "pair_384" else 1)
    omods = [onet.load_synthetic(onet.fire_module(a, b, stride=st), seed=80 + i) for i, (a, b, st) in enumerate(shapes)]
    mods = []
    for o, (a, b, st) in zip(omods, shapes):
        m = bb.fire_module(a, b, stride=st)
        m.load_state_dict(o.state_dict())
        mods.append(m.eval())
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    x = torch.randn((64, 8, 8, 384), generator=gen, device="cuda").to(dtype)
    l0 = ops.COUNTERS["launches"]
    got = bb.run_fire_modules(mods, ops.Act(x))
    assert ops.COUNTERS["launches"] - l0 == 1
    sample = [0, 31, 63]
    with torch.no_grad():
        ref = x[sample].float().permute(0, 3, 1, 2).cpu()
        for o in omods:
            ref = o(ref)
    g = got.t[sample].float().permute(0, 3, 1, 2).cpu()
    assert g.shape == ref.shape
    scale = float(ref.abs().max())
    eps = (0.03 if dtype == torch.bfloat16 else 0.004) * max(1, len(shapes) // 2)
    assert float((g - ref).abs().max()) <= eps * scale + eps, float((g - ref).abs().max())
    for i in sample:
        one = bb.run_fire_modules(mods, ops.Act(x[i:i + 1].contiguous()))
        assert torch.equal(one.t, got.t[i:i + 1]), f"frame {i} differs between batch 64 and batch 1"
    again = bb.run_fire_modules(mods, ops.Act(x))
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


@pytest.mark.parametrize("side,dtype", [(False, torch.bfloat16), (True, torch.bfloat16), (True, "float32mix"), (True, "float32x3")])
def test_graph_replay_equals_eager_at_batch64(side, dtype):
    """hipGraph replays of the 64-frame step are bit-equal to the eager step, five replays in a row, with the hourglass branches
    captured serially and forked onto side streams (bf16), for the mixed configuration (fp16 sub-networks, casts, fp16 side
    outputs and the chunked high-resolution front inside the capture) and for the split-product configuration (its stem, patch-resident
    and one-launch fire kernels of round 5 under capture)."""
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import pipeline as op
    net = _net(dtype)
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


def test_config4_cups_multi_object_batch64():
    """BASELINE configs[3], one rank's share: 64 frames, config/cups.json (K = 4: centre + three single-instance keypoint
    types), multi-object scenes (4 objects per frame).  The bf16 network runs the 64 frames (shapes of the four-map heads,
    frame independence), and peaks -> 3D -> objects on the injected scenes equal the oracle's pipeline frame by frame for
    a sample; `points` is the fixed-capacity all-gather payload [64, 4, cap, 4] (NaN = unused slot)."""
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import pipeline as op
    cfg = {"keypoint_config": [1, 1, 1]}
    net = _net(torch.bfloat16, k=4)
    cam_o = op.eval_camera(CALIB)
    pipe = pp.BatchedKeypointPipeline(net, cfg, cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size), capacity=128)
    x = _frames(64, seed=11)
    with torch.no_grad():
        out = pipe.forward_device(x)
        assert tuple(out["heat"].shape) == (64, 4, 64, 64) and tuple(out["centers"].shape) == (64, 3, 2, 64, 64)
        assert tuple(out["points"].shape) == (64, 4, 128, 4) and out["points"].dtype == torch.float64
        rev = pipe.forward_device(torch.flip(x, dims=[0]).contiguous())
        assert torch.equal(out["heat"], torch.flip(rev["heat"], dims=[0])) and torch.equal(out["count"], torch.flip(rev["count"], dims=[0]))
    scenes = [synth.bump_scene([1, 1, 1], n_objects=4, seed=31, index=i) for i in range(64)]
    to = lambda key: torch.from_numpy(np.stack([s[key] for s in scenes])).cuda()
    post = pipe.postprocess_device(to("heat"), to("depth"), to("centers"))
    assert not bool(post["overflow"])
    pts = post["points"].cpu().numpy()
    cnt = post["count"].cpu().numpy()
    assert int(cnt[:, 0].min()) >= 3                       # the four centres (two may merge inside one 5x5 window)
    for n in range(64):
        for k in range(4):
            c = int(cnt[n, k])
            assert np.isnan(pts[n, k, c:]).all() and not np.isnan(pts[n, k, :c]).any()
    opipe = op.ObjectKeypointPipeline([64, 64], None, cfg)
    opipe.reset(cam_o)
    for n in (0, 21, 42, 63):
        want = opipe(scenes[n]["heat"][None], scenes[n]["depth"][None], scenes[n]["centers"][None])
        got = pipe.objects(post, n)
        assert len(got) == len(want) >= 3
        for g, w_ in zip(got, want):
            for a, b in zip(g["keypoints"], w_["keypoints"]):
                assert np.asarray(a).shape == np.asarray(b).shape
                if np.asarray(a).size:
                    np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
            for a, b in zip(g["p_C"], w_["p_C"]):
                assert (a is None) == (b is None)
                if a is not None:
                    np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)          # north_star: 3D points within 1e-4 m


def test_config5_stereo_stream_fp16_network_fp32_geometry():
    """BASELINE configs[4] in miniature: 8 camera streams = 4 stereo pairs, fp16 convolutions, fp32 / fp64 geometry.
    (a) the fp16 network on the 8 frames stays within the absolute fp16 bounds of the fp32 HIP path (which meets the 1e-3
    bar against the reference), frame by frame; (b) on heat maps rendered from known 3D points through the left and right
    cameras, peaks -> association -> triangulation recover the points: the geometry never sees fp16."""
    import json
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import geometry as og
    _need(torch.float16)
    x = _frames(8, seed=21)
    with torch.no_grad():
        h16, d16, c16 = _net(torch.float16).deployed(x)
        h32, d32, c32 = _net(torch.float32).deployed(x)
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests", "precision"))
    from bounds import BOUNDS                              # absolute fp16 bounds (tests/precision/bounds.py)
    for n in range(8):
        assert float((h16[n] - h32[n]).abs().max()) <= BOUNDS["f16"]["heat_max"]
        assert float((d16[n] - d32[n]).abs().max()) <= BOUNDS["f16"]["depth_max"]
    assert h16.dtype == torch.float32 and d16.dtype == torch.float32          # heads hand fp32 maps to the geometry stage
    p = og.load_calibration_params(CALIB)
    scale = 0.25
    oleft = og.FisheyeCamera(p["K"], p["D"], p["image_size"]).scale(scale)
    oright = og.FisheyeCamera(p["Kp"], p["Dp"], p["image_size"]).scale(scale)
    stereo = cu.StereoCamera(cu.FisheyeCamera(oleft.K, oleft.D, oleft.image_size), cu.FisheyeCamera(oright.K, oright.D, oright.image_size), p["T_RL"])
    assoc = pp.AssociationComponent(max_distance=3.0); assoc.reset(stereo)
    tri = pp.TriangulationComponent(); tri.reset(stereo)
    H, W = 180, 320
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    base = np.array([[0.05, 0.02, 1.0], [0.21, 0.10, 1.1], [-0.18, -0.12, 0.9], [0.10, -0.20, 1.2], [-0.25, 0.15, 1.05]])
    maps, truth = [], []
    for pair in range(4):                                  # four rigs looking at the point set from slightly different poses
        # (points well separated across epipolar lines: matching by epipolar distance alone is ambiguous otherwise)
        X = base * np.array([1.0 - 0.04 * pair, 1.0 + 0.03 * pair, 1.0]) + np.array([0.01 * pair, -0.008 * pair, 0.05 * pair])
        truth.append(X)
        for cam, T in ((oleft, np.eye(4)), (oright, p["T_RL"])):
            m = np.zeros((H, W), np.float32)
            for px, py in cam.project(X, T):
                m += np.exp(-((xs - px) ** 2 + (ys - py) ** 2) / 4.0).astype(np.float32)
            maps.append(np.clip(m, 0, 1))
    heat = torch.from_numpy(np.stack(maps)[:, None]).cuda()                 # [8 streams, 1 map, H, W]: ONE launch for all cameras
    count, _, xyc = [t.cpu().numpy() for t in ops.peak_nms(heat, cap=16)]
    for pair in range(4):
        cl, cr = int(count[2 * pair, 0]), int(count[2 * pair + 1, 0])
        left2d, right2d = xyc[2 * pair, 0, :cl, :2].astype(np.float64), xyc[2 * pair + 1, 0, :cr, :2].astype(np.float64)
        match = assoc(left2d, right2d)
        ok = match >= 0
        assert ok.sum() >= 4                               # two bumps of a view may merge; the rest must pair up
        X_hat = tri(left2d[ok], right2d[match[ok]])
        d = np.linalg.norm(X_hat[:, None] - truth[pair][None], axis=2)
        assert d.min(axis=1).max() < 0.05, d.min(axis=1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["conv3x3_64", "conv3x3_s2_128", "residual_s2_skip_128", "unpool_32_to_64"])
def test_patch_kernel_at_batch64_against_cpu_convolution(case, dtype):
    """The dominant kernel at the bench's batch of 64 against an INDEPENDENT reference: torch's fp32 CPU convolution of the same
    (16-bit-rounded) operands on frames 0, 31 and 63 - stride 1 with residual, stride 2 (four parity-class patches), the fused
    conv2 + strided 1x1 skip, and the transposed convolution's four sub-pixel classes with the merge add.  (The tile-13-equals-
    tile-6 tests compare two kernels of this build with each other; this one does not.)"""
    import torch.nn.functional as F
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception.backbone import conv_taps, unpool_merge
    _need(dtype)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(11)
    rnd = lambda *shape: torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(dtype)
    q = lambda a: torch.from_numpy(a).to(dtype).float()
    rw = lambda name, shape, fan: synth.normal_like(name, shape, 60) / np.float32(np.sqrt(fan))
    n, frames = 64, (0, 31, 63)
    nchw = lambda t, i: t[i:i + 1].float().permute(0, 3, 1, 2).cpu()
    if case == "unpool_32_to_64":
        m = unpool_merge(256).eval()
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(rw("up2", (256, 256, 4, 4), 4 * 256))); m.bias.copy_(torch.from_numpy(synth.normal_like("up2b", (256,), 61) * np.float32(0.1)))
        low, up1 = rnd(n, 32, 32, 256), rnd(n, 64, 64, 256)
        ops.LAUNCH_HOOK = hook = _TileSpy()
        try:
            got = m(ops.Act(low), ops.Act(up1)).t
        finally:
            ops.LAUNCH_HOOK = None
        assert hook.tiles == [13]
        wq, bq = m.weight.detach().to(dtype).float(), m.bias.detach().float()
        for i in frames:
            ref = nchw(up1, i) + F.conv_transpose2d(nchw(low, i), wq, bq, stride=2, padding=1)
            err = float((nchw(got, i) - ref).abs().max())
            assert err <= (0.03 if dtype == torch.bfloat16 else 0.004) * (1.0 + float(ref.abs().max())), (i, err)
        return
    if case == "conv3x3_64":
        h = w = 64
        wt, b = rw("w64", (256, 256, 3, 3), 256 * 9), synth.normal_like("b64", (256,), 62) * np.float32(0.1)
        plan = ops.ConvPlan(dtype, [256], [1], 256, conv_taps(wt), b, relu=True)
        srcs, res = [rnd(n, h, w, 256)], rnd(n, h, w, 256)
        ref_fn = lambda i: F.relu(F.conv2d(nchw(srcs[0], i), q(wt), torch.from_numpy(b), padding=1) + nchw(res, i))
    elif case == "conv3x3_s2_128":
        h = w = 128
        wt, b = rw("ws2", (256, 128, 3, 3), 128 * 9), synth.normal_like("bs2", (256,), 63) * np.float32(0.1)
        plan = ops.ConvPlan(dtype, [128], [2], 256, conv_taps(wt), b, relu=True)
        srcs, res = [rnd(n, 2 * h, 2 * w, 128)], None
        ref_fn = lambda i: F.relu(F.conv2d(nchw(srcs[0], i), q(wt), torch.from_numpy(b), stride=2, padding=1))
    else:
        h = w = 128
        w2, ws = rw("w2", (256, 256, 3, 3), 256 * 9), rw("wskip", (256, 128, 1, 1), 128)
        b = synth.normal_like("b2", (256,), 64) * np.float32(0.1)
        taps = conv_taps(w2) + [(1, 0, 0, np.ascontiguousarray(ws[:, :, 0, 0]))]
        plan = ops.ConvPlan(dtype, [256, 128], [1, 2], 256, taps, b, relu=True)
        srcs, res = [rnd(n, h, w, 256), rnd(n, 2 * h, 2 * w, 128)], None
        ref_fn = lambda i: F.relu(F.conv2d(nchw(srcs[0], i), q(w2), torch.from_numpy(b), padding=1) + F.conv2d(nchw(srcs[1], i), q(ws), stride=2))
    out = ops.Act.empty(n, h, w, 256, dtype, dev)
    plan([ops.Act(s) for s in srcs], out, h, w, res=ops.Act(res) if res is not None else None, tile=13)
    for i in frames:
        ref = ref_fn(i)
        err = float((nchw(out.t, i) - ref).abs().max())
        assert err <= (0.03 if dtype == torch.bfloat16 else 0.004) * (1.0 + float(ref.abs().max())), (i, err)      # output rounding of O(1-5) values


class _TileSpy:
    def __init__(self):
        self.tiles = []

    def before(self, plan, tile, macs):
        self.tiles.append(tile)

    def after(self, token):
        pass


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_heads_at_batch64_against_the_oracle(dtype):
    """okp_heads at N=64 against the ORACLE's prediction modules (torch CPU fp32 on the same 16-bit-rounded input) on frames
    0, 31 and 63: heat (post-sigmoid), depth and centre maps."""
    from object_keypoints_amd import ops
    from oracle import net as onet
    _need(dtype)
    net = _net(dtype, seed=3)
    oracle = onet.load_synthetic(onet.KeypointNet(features=128, heatmaps_out=3), seed=3)
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    cnv = (torch.randn((64, 64, 64, 256), generator=gen, device="cuda") * 0.7).to(dtype)
    heat, depth, centers = net._run_heads(1, ops.Act(cnv), sigmoid=True)
    tol = 0.02 if dtype == torch.bfloat16 else 0.003          # three 16-bit roundings (input, two hidden layers) of O(1) values
    for i in (0, 31, 63):
        x = cnv[i:i + 1].float().permute(0, 3, 1, 2).cpu()
        with torch.no_grad():
            rh = torch.sigmoid(oracle.heatmap_head.output_head2(x)); rd = oracle.depth_head.output_head2(x); rc = oracle.center_head.output_head2(x)
        assert float((heat[i:i + 1].cpu() - rh).abs().max()) <= tol
        assert float((depth[i:i + 1].cpu() - rd).abs().max()) <= tol * (1.0 + float(rd.abs().max()))
        assert float((centers[i:i + 1].cpu().reshape(rc.shape) - rc).abs().max()) <= tol * (1.0 + float(rc.abs().max()))


def test_stereo_stream_pipeline_tick_recovers_the_rendered_points():
    """StereoStreamPipeline (BASELINE configs[4]: a tick = the frames of all stereo pairs -> one network batch -> peaks -> association ->
    ONE triangulation launch): on heat maps rendered from known 3D points through the left and right camera of every pair the tick
    returns those points per pair and key-point type; the hipGraph form of the network pass gives the same bits; an odd frame count
    and an exceeded capacity are refused."""
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import geometry as og
    p = og.load_calibration_params(CALIB)
    offset = np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])
    mk = lambda K, D: cu.FisheyeCamera(K, D, p["image_size"]).scale(511 / 720).cut(offset).scale(64 / 511)
    left, right = mk(p["K"], p["D"]), mk(p["Kp"], p["Dp"])
    stereo = cu.StereoCamera(left, right, p["T_RL"])
    net = _net(torch.float16)
    pipe = pp.StereoStreamPipeline(net, stereo, {"keypoint_config": [1, 3]}, capacity=8, max_distance=1.5)
    n_pairs, K, size = 2, 3, 64
    ys, xs = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    base = {0: [[0.00, 0.00, 0.55]], 1: [[0.12, -0.10, 0.60]], 2: [[-0.16, 0.12, 0.50], [0.14, 0.04, 0.65], [-0.12, -0.16, 0.70]]}
    heat = np.zeros((2 * n_pairs, K, size, size), np.float32)
    truth = []
    for pr in range(n_pairs):
        t = {}
        for k, pts in base.items():
            X = np.array(pts) + np.array([0.02 * pr, -0.015 * pr, 0.04 * pr])
            t[k] = X
            for side, (cam, T) in enumerate(((left, np.eye(4)), (right, p["T_RL"]))):
                for px, py in cam.project(X, T):
                    heat[2 * pr + side, k] += np.exp(-((xs - px) ** 2 + (ys - py) ** 2) / 4.0)
        truth.append(t)
    heat_dev = torch.from_numpy(np.clip(heat, 0, 1)).cuda()
    frames = _frames(2 * n_pairs, seed=31)
    out = pipe.tick(frames, heat_override=heat_dev)
    assert len(out) == n_pairs
    for pr in range(n_pairs):
        for k in range(K):
            got, want = out[pr][k], truth[pr][k]
            assert got.shape == want.shape and got.dtype == np.float64
            d = np.linalg.norm(got[:, None] - want[None], axis=2).min(axis=1)
            assert d.max() < 0.06                      # 64 x 64 maps, 6.2 cm baseline: centimetres (bench.py: run_stream8)
    pipe.capture(frames)
    out_g = pipe.tick(frames, heat_override=heat_dev, use_graph=True)
    with pytest.raises(pp.OkpError):
        pipe.tick(frames)                              # the network's own maps (random weights: ~100 peaks each) exceed the capacity: loud
    for pr in range(n_pairs):
        for k in range(K):
            assert np.array_equal(out[pr][k], out_g[pr][k])
    with pytest.raises(pp.OkpError):
        pipe.tick(frames[:3])
    tiny = pp.StereoStreamPipeline(net, stereo, {"keypoint_config": [1, 3]}, capacity=2)
    with pytest.raises(pp.OkpError):
        tiny.tick(frames, heat_override=heat_dev)      # three bumps in one map, room for two


def test_stream_wait_stream_orders_work_across_streams():
    """okp_stream_wait_stream (the fork / join of the hourglass branches; events without a system-scope fence): a consumer stream sees
    what the producer stream wrote before the edge, 600 edges in a row (the ring of 256 events is reused), in both directions; with
    ops.LIGHT_EVENTS = False semantics (torch's wait_stream) as the reference behaviour."""
    from object_keypoints_amd import ops
    dev = torch.device("cuda", 0)
    a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    n = 32 << 20
    src = torch.zeros(n, dtype=torch.float32, device=dev)
    mid = torch.zeros(n, dtype=torch.float16, device=dev)
    dst = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    bad = 0
    for it in range(600):
        prod, cons = (a, b) if it % 2 == 0 else (b, a)
        with torch.cuda.stream(prod):
            src.fill_(float(it % 1000))                                   # a 128 MB fill: still running when the consumer is enqueued
            ops.cast(ops.Act(src.view(1, 1, -1, 64)), torch.float16).t.view(-1)[:1]      # (another launch behind it on the producer)
            mid.copy_(src)
        ops.stream_wait(cons, prod)
        with torch.cuda.stream(cons):
            dst.copy_(mid)
            probe = dst[::1 << 20].clone()
        ops.stream_wait(prod, cons)                                        # the next iteration's producer must not overwrite early
        if it % 50 == 49:
            torch.cuda.synchronize()
            bad += int((probe != float(it % 1000)).sum())
    torch.cuda.synchronize()
    assert bad == 0
    assert ops.LIGHT_EVENTS


def test_host_fed_stream_equals_the_resident_tick():
    """configs[4] as a STREAM (reference harness: scripts/eval_model.py:274-293 feeds the model from the host; raw frames are 1280 x 720 uint8,
    perception/datasets/video.py:83-100): raw camera frames start in pinned host memory, cross PCIe on a copy stream under the previous
    tick (HostFrameFeed), are resized / cropped / normalised on the device, and the tick's results - the network's maps AND the triangulated
    points - equal, bit for bit, those of the resident tick on the fp32 crops the ORACLE's pre-processing makes of the same frames; several
    ticks in flight (different frames per tick) come out in order."""
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import geometry as og
    from oracle import pipeline as op
    from oracle import preprocess as opre
    p = og.load_calibration_params(CALIB)
    offset = np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])
    mk = lambda K, D: cu.FisheyeCamera(K, D, p["image_size"]).scale(511 / 720).cut(offset).scale(64 / 511)
    stereo = cu.StereoCamera(mk(p["K"], p["D"]), mk(p["Kp"], p["Dp"]), p["T_RL"])
    net = _net(torch.float16)
    pipe = pp.StereoStreamPipeline(net, stereo, {"keypoint_config": [1, 3]}, capacity=4096, max_distance=1.5)
    rng = np.random.default_rng(12)
    ticks_u8 = [rng.integers(0, 256, (2, 720, 1280, 3), dtype=np.uint8) for _ in range(3)]
    hosts = [torch.from_numpy(t).pin_memory() for t in ticks_u8]
    # resident reference: the oracle's resize + centre crop + normalisation (fp32 NCHW 511 x 511), uploaded beforehand
    crops = [torch.from_numpy(op.normalize_frames(np.stack([opre.resize_center_crop(f, 511) for f in t]))).cuda() for t in ticks_u8]
    with torch.no_grad():
        want_maps = [[m.clone() for m in net.deployed(c)] for c in crops]
        want_ticks = [pipe.tick(c) for c in crops]
        # maps: host-fed through the feed, tick by tick
        feed = pp.HostFrameFeed(hosts[0].shape, hosts[0].dtype)
        slot = feed.submit(hosts[0])
        for i in range(3):
            frames = feed.acquire(slot)
            nxt = feed.submit(hosts[i + 1]) if i + 1 < 3 else None
            got = net.deployed(frames)
            feed.release(slot)
            for g, w_ in zip(got, want_maps[i]):
                assert torch.equal(g, w_), f"tick {i}: host-fed maps differ from the resident tick's"
            slot = nxt
        # the whole tick, streamed: results in order, equal to the resident ticks
        got_ticks = list(pipe.stream(iter(hosts)))
    assert len(got_ticks) == 3
    for got, want in zip(got_ticks, want_ticks):
        assert len(got) == len(want) == 1
        for k in want[0]:
            assert np.array_equal(got[0][k], want[0][k])
    # a tensor of another shape, or one that already lives on the device, is refused by the feed
    with pytest.raises(pp.OkpError):
        feed.submit(hosts[0][:1])
    with pytest.raises(pp.OkpError):
        feed.submit(hosts[0].cuda())
