"""The PyTorch-ROCm custom-op boundary (north_star: "surfaced through PyTorch-ROCm custom ops"; SURVEY 8(b) "What the native
replacement must export"): the whole path - plan creation from a state_dict, the split-product network with its ABI-5/6 launch
arguments (pair-format tensors, fp16 side outputs), peaks, lifting, grouping, triangulation - runs on torch.ops.okp.* with NOT ONE
call through the ctypes binding, bit-equal to the ctypes path; and a torch-only caller builds and runs a convolution block with
nothing but torch.ops.okp calls."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALIB = os.path.join(REPO, "config", "calibration.yaml")


def _state(k=3, seed=0):
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.models import KeypointNet
    shapes = {kk: tuple(v.shape) for kk, v in KeypointNet(features=128, heatmaps_out=k).state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=seed)
    return {kk: torch.from_numpy(np.array(v)) for kk, v in vals.items()}


def _net(dtype, sd, k=3):
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=k, compute_dtype=dtype)
    net.load_state_dict(sd)
    return net.eval().cuda()


def _frames(n, seed=77):
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    return torch.randn((n, 3, 511, 511), generator=gen, device="cuda", dtype=torch.float32)


def _scene_maps(n, cfg, seed=5, n_objects=1):
    from object_keypoints_amd import synth
    scenes = [synth.bump_scene(cfg, n_objects=n_objects, seed=seed, index=i) for i in range(n)]
    return tuple(torch.from_numpy(np.stack([s[key] for s in scenes])).cuda() for key in ("heat", "depth", "centers"))


@pytest.mark.parametrize("dtype", ["float32x3", "float32mix", torch.bfloat16, torch.float32])
def test_whole_path_runs_on_the_dispatcher_with_zero_ctypes_calls(dtype):
    """A batch-16 network pass (float32x3 takes the patch-resident kernel with pair-format tensors, the split-product stem writing pairs and
    the one-launch split-product heads; float32mix adds the fp16 side outputs, fp16 residuals and subsampled outputs) and the batched
    pipeline behind it: COUNTERS["ctypes_launches"] stays 0 from plan creation on, and every output equals the ctypes binding's bit for bit."""
    from object_keypoints_amd import _lib, ops
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import pipeline as op
    assert _lib.torch_ops() is not None, "libokp_torch.so is built by build() next to libokp_hip.so"
    cfg = {"keypoint_config": [1, 3]}
    sd = _state()
    cam_o = op.eval_camera(CALIB)
    cam = cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size)
    x = _frames(16)
    maps = _scene_maps(16, [1, 3])

    def run():
        net = _net(dtype, sd)                         # plans are created inside the first pass, through the binding in force
        pipe = pp.BatchedKeypointPipeline(net, cfg, cam)
        with torch.no_grad():
            out = pipe.forward_device(x)
            post = pipe.postprocess_device(*maps)
        torch.cuda.synchronize()
        keys = ("heat", "depth", "centers", "count", "xyc", "points")
        return [out[k].clone() for k in keys] + [post[k].clone() for k in ("count", "yx", "xyc", "points", "n_obj", "sel", "n_votes", "assign", "pred")]

    before = dict(ops.COUNTERS)
    a = run()
    assert ops.COUNTERS["ctypes_launches"] == before["ctypes_launches"], "a launch or a plan creation went through ctypes while torch.ops.okp is loaded"
    assert ops.COUNTERS["launches"] > before["launches"]
    if dtype == "float32x3":
        assert ops.COUNTERS["pair_outputs"] > before["pair_outputs"]          # the ABI-6 arguments went through torch.ops.okp.conv_forward
    keep = _lib._torch_ops
    try:
        _lib._torch_ops = None                        # the ctypes binding serves everything
        c0 = ops.COUNTERS["ctypes_launches"]
        b = run()
        assert ops.COUNTERS["ctypes_launches"] > c0
    finally:
        _lib._torch_ops = keep
    for u, v in zip(a, b):
        assert u.dtype == v.dtype and torch.equal(torch.nan_to_num(u.double(), nan=-7.0), torch.nan_to_num(v.double(), nan=-7.0))


def test_torch_only_caller_builds_and_runs_a_convolution_block():
    """INTEGRATION.md's recipe, executed: plan creation from a module's tensors (conv_bn_create: fold BatchNorm, tap list, pack, upload)
    and the launch, with torch.ops.okp calls only - no Python helper of this package - against the torch-CPU convolution."""
    from object_keypoints_amd import _lib, ops
    T = _lib.torch_ops()
    assert T is not None
    g = torch.Generator().manual_seed(3)
    conv = torch.nn.Conv2d(64, 128, 3, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(128).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / 24.0)
        bn.weight.copy_(torch.rand(128, generator=g) + 0.5); bn.bias.copy_(torch.randn(128, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(128, generator=g) * 0.1); bn.running_var.copy_(torch.rand(128, generator=g) + 0.5)
    x = torch.randn((2, 64, 32, 48), generator=g)
    with torch.no_grad():
        want = torch.relu(bn(conv(x)))
    before = ops.COUNTERS["ctypes_launches"]
    for code, tol in ((_lib.OKP_F32, 2e-5), (_lib.OKP_F32X3, 2e-5)):
        plan = T.conv_bn_create(code, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, None, 1, True)
        xa = x.permute(0, 2, 3, 1).contiguous().cuda()                     # NHWC on the device
        out = torch.empty((2, 32, 48, 128), device="cuda")
        T.conv_forward(plan, xa, 0, None, 0, out, 0, 32, 48, None, 0, 1, 0, 0, 0, 1, None, None, None, 0, None, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = out.permute(0, 3, 1, 2).cpu()
        assert float((got - want).abs().max()) <= tol * (1 + float(want.abs().max()))
        assert T.conv_macs(plan, 2, 32, 48) == 2 * 32 * 48 * 128 * 64 * 9
        T.conv_destroy(plan)
    assert ops.COUNTERS["ctypes_launches"] == before
    # host tensors are refused where the device is expected, device tensors where host weights are expected
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        plan = T.conv_bn_create(_lib.OKP_F32, conv.weight, None, None, None, None, 1e-5, None, 1, False)
        try:
            T.conv_forward(plan, x.permute(0, 2, 3, 1).contiguous(), 0, None, 0, out, 0, 32, 48, None, 0, 1, 0, 0, 0, 1, None, None, None, 0, None, 0, 0)
        finally:
            T.conv_destroy(plan)
    with pytest.raises(RuntimeError, match="HOST tensor"):
        T.conv_create(_lib.OKP_F32, [64], [1], 128, [0], [0], [0], [conv.weight[:, :, 1, 1].cuda()], None, 0, [])


def test_geometry_and_boundary_ops_on_the_dispatcher_equal_ctypes():
    """triangulate_dlt, unproject_depth, camera_undistort, nms_maxpool, preprocess_u8, pack_frames(_u8), cast, add_f16_f32, dwconv3x3 and the
    packed stem: each through torch.ops.okp and through ctypes, same bits."""
    from object_keypoints_amd import _lib, ops
    from oracle import pipeline as op
    assert _lib.torch_ops() is not None
    cam_o = op.eval_camera(CALIB)
    cam = ops.make_camera(cam_o.K, cam_o.D)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    xy = torch.rand((257, 2), generator=g, device="cuda") * 60 + 2
    xy2 = xy + torch.randn((257, 2), generator=g, device="cuda") * 0.3
    depth = torch.rand((4, 64, 64), generator=g, device="cuda") + 0.3
    ids = (torch.arange(257, device="cuda") % 4).int()
    T_RL = np.eye(4)[:3]; T_RL[0, 3] = -0.1
    F = np.array([[0, -1e-3, 0.02], [1e-3, 0, -0.3], [-0.02, 0.3, 0.0]])
    heat = torch.rand((3, 2, 64, 64), generator=g, device="cuda")
    u8 = torch.randint(0, 256, (2, 720, 1280, 3), generator=g, device="cuda", dtype=torch.uint8)
    u8s = torch.randint(0, 256, (2, 63, 65, 3), generator=g, device="cuda", dtype=torch.uint8)
    fr = torch.randn((2, 3, 63, 65), generator=g, device="cuda")
    a32 = ops.Act(torch.randn((2, 16, 16, 64), generator=g, device="cuda"))
    wd = torch.randn((9, 64), generator=g, device="cuda") * 0.2
    bd = torch.randn((64,), generator=g, device="cuda") * 0.1
    stem_w = np.random.default_rng(1).standard_normal((128, 3, 7, 7)).astype(np.float32) / 12
    stem_b = np.zeros(128, np.float32)

    def run():
        res = [ops.triangulate_dlt(cam, cam, T_RL, xy, xy2), ops.triangulate_dlt(cam, cam, T_RL, xy, xy2, F=F),
               ops.unproject_depth(cam, xy, ids, depth, 63, 63), ops.camera_undistort(cam, xy), ops.nms_maxpool(heat, 5),
               ops.preprocess_u8(u8, torch.bfloat16).t, ops.pack_frames_u8(u8s, torch.float16).t, ops.pack_frames(fr, torch.float32).t]
        h16 = ops.cast(a32, torch.float16)
        res += [h16.t, ops.cast(h16, torch.float32).t, ops.add_f16_f32(h16, a32).t]
        o = ops.Act.empty(2, 16, 16, 64, torch.float32, "cuda")
        ops.dwconv3x3(a32, wd, bd, o, 1, res=a32)
        res.append(o.t)
        stem = ops.StemPlan(stem_w, stem_b, torch.bfloat16)
        so = ops.Act.empty(2, 32, 33, 128, torch.bfloat16, "cuda")
        stem(ops.pack_frames(fr, torch.bfloat16), so)
        res.append(so.t)
        flag = ops.capacity_overflow(torch.tensor([[3, 1], [70, 2]], dtype=torch.int32, device="cuda"), 64, 16)
        res.append(flag.reshape(1))
        torch.cuda.synchronize()
        return [r.clone() for r in res]

    c0 = ops.COUNTERS["ctypes_launches"]
    a = run()
    assert ops.COUNTERS["ctypes_launches"] == c0
    keep = _lib._torch_ops
    try:
        _lib._torch_ops = None
        b = run()
        assert ops.COUNTERS["ctypes_launches"] > c0
    finally:
        _lib._torch_ops = keep
    assert int(a[-1]) == 1
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.dtype == v.dtype and u.shape == v.shape
        assert torch.equal(torch.nan_to_num(u.double(), nan=-7.0), torch.nan_to_num(v.double(), nan=-7.0)), f"result {i} differs between the bindings"
