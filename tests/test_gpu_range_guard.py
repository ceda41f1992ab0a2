"""The fp16-range guard of the split-product configurations (include/okp.h: okp_conv_set_range_flag; KeypointNet.on_overflow).

float32x3 multiplies fp32 operands as fp16 halves: a value beyond +-65504 has no halves (inf, -inf), its products are NaN and the next ReLU
turns the NaN into 0 - silently - where the reference's fp32 model (perception/pipeline.py:20-27, py_utils/utils.py:143-156) returns numbers.
Every split-product kernel raises a device flag when a value it hands on leaves that range; a pass raises OkpError on it, or re-runs in exact
float32.  Healthy passes are bit-identical with and without the guard."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALIB = os.path.join(REPO, "config", "calibration.yaml")


def _state(seed=0, k=3):
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.models import KeypointNet
    shapes = {kk: tuple(v.shape) for kk, v in KeypointNet(features=128, heatmaps_out=k).state_dict().items()}
    return {kk: torch.from_numpy(np.array(v)) for kk, v in synth.fill_state_dict(shapes, seed=seed).items()}


def _net(dtype, sd, k=3, on_overflow="raise"):
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=k, compute_dtype=dtype)
    net.load_state_dict(sd)
    net.on_overflow = on_overflow
    return net.eval().cuda()


def _frames(n, seed=77):
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    return torch.randn((n, 3, 511, 511), generator=gen, device="cuda", dtype=torch.float32)


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


def test_layers_raise_the_flag_and_are_bit_identical_with_and_without_it():
    """The gather tiles, the patch-resident kernel (fp32 and pair-format output), the depth-wise branch and the one-launch fire module: with a
    flag attached a healthy launch leaves it 0 and writes the same bits as the unguarded plan; results beyond 65504 raise it."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps, fire_module
    dev = torch.device("cuda")
    w = (_rand((256, 256, 3, 3), 1) / np.sqrt(256 * 9)).numpy()
    b = np.zeros(256, np.float32)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def conv(n, hw, guarded, out_pairs=False, poison=None, tile=0):
        with ops.f32_split(True, False, flag if guarded else None):
            plan = ops.ConvPlan(torch.float32, [256], [1], 256, conv_taps(w), b, relu=True)
        x = _rand((n, hw, hw, 256), 2).to(dev)
        if poison is not None:
            x.fill_(6.0e4)
        out = ops.Act.empty(n, hw, hw, 256, torch.float32, dev)
        plan([ops.Act(x)], out, hw, hw, out_pairs=out_pairs, tile=tile)
        torch.cuda.synchronize()
        return out.t.clone()

    # the gather tiles (64 x 64, 128 x 128, 256 x 256); the patch-resident kernel with fp32 and with pair-format output
    for n, hw, pairs, tile in ((1, 16, False, 1), (4, 32, False, 2), (4, 32, False, 3), (4, 32, False, 13), (4, 32, True, 13)):
        flag.zero_()
        a, g = conv(n, hw, False, pairs, tile=tile), conv(n, hw, True, pairs, tile=tile)
        assert int(flag) == 0 and torch.equal(a.view(torch.int32), g.view(torch.int32))
        # operands IN range, a result out of it: every operand 6e4, so a channel's result is 6e4 x (the sum of its 2 304 weights, ~N(0, 1)).
        # (An operand that is itself out of range or NaN is its producer's business: inside the network every tensor a split-product kernel
        #  reads was written - and checked - by one, include/okp.h; data from outside is checked exactly where it enters: the stem, okp_cast.)
        flag.zero_()
        conv(n, hw, True, pairs, poison="big", tile=tile)
        assert int(flag) == 1, (n, hw, pairs, tile)
    # the one-launch split-product fire module (okp_fire_x3) and the unfused path with the depth-wise branch of the gather kernel
    fm = fire_module(256, 256).eval()
    g = torch.Generator().manual_seed(3)
    sdf = {}
    for k, v in fm.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sdf[k] = v
        elif v.dim() == 4:
            sdf[k] = torch.randn(v.shape, generator=g) / float(np.sqrt(v.shape[1] * v.shape[2] * v.shape[3]))
        elif k.endswith(("running_var", "weight")):
            sdf[k] = torch.rand(v.shape, generator=g) + 0.5
        else:
            sdf[k] = torch.randn(v.shape, generator=g) * 0.1
    fm.load_state_dict(sdf)
    fm = fm.cuda()
    for fused in (True, False):
        keep = ops.FUSE_FIRE_X3
        ops.FUSE_FIRE_X3 = fused
        try:
            outs = []
            for guarded in (False, True):
                fm._drop_plans()
                x = _rand((4, 32, 32, 256), 5).to(dev)
                flag.zero_()
                with ops.f32_split(True, False, flag if guarded else None):
                    outs.append(fm(ops.Act(x)).t.clone())
                torch.cuda.synchronize()
                assert int(flag) == 0
            assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
            x = ((torch.rand((4, 32, 32, 256), generator=torch.Generator().manual_seed(6)) * 2 - 1) * 6.0e4).to(dev)      # every operand in range,
            flag.zero_()                                                                                                  # squeeze values and results not
            with ops.f32_split(True, False, flag):
                fm(ops.Act(x))
            assert int(flag) == 1, f"fire module, fused={fused}"
        finally:
            ops.FUSE_FIRE_X3 = keep


def test_network_with_one_large_batchnorm_gain():
    """One BatchNorm gain scaled so that an activation passes 65504: float32x3 raises (default) or returns the exact-float32 result
    (on_overflow='float32'), exact float32 returns finite numbers as the reference does; the unscaled network is untouched by the guard."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import pipeline as op
    sd = _state()
    x = _frames(16)
    good = _net("float32x3", sd)
    with torch.no_grad():
        a = [t.clone() for t in good.deployed(x)]
        assert not good.range_overflow()
        b = good.deployed(x)
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    big = dict(sd)
    big["backbone.pre.1.bn2.weight"] = sd["backbone.pre.1.bn2.weight"] * 3.0e5
    exact = _net(torch.float32, big)
    with torch.no_grad():
        want = [t.clone() for t in exact.deployed(x[:2])]
    assert all(bool(torch.isfinite(t).all()) for t in want)          # the reference's arithmetic returns numbers
    raising = _net("float32x3", big)
    with torch.no_grad(), pytest.raises(ops.OkpError, match="fp16 range"):
        raising.deployed(x[:2])
    assert raising.range_overflow()
    falling = _net("float32x3", big, on_overflow="float32")
    with torch.no_grad(), pytest.warns(RuntimeWarning, match="re-run with the exact float32 kernels"):
        got = falling.deployed(x[:2])
    assert all(torch.equal(u, v) for u, v in zip(got, want)) and falling.configuration() == "float32x3"
    # deferred: nothing raises in the pass, the batched pipeline's overflow word carries the bit and objects() refuses the batch
    cam_o = op.eval_camera(CALIB)
    cam = cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size)
    pipe = pp.BatchedKeypointPipeline(_net("float32x3", big), {"keypoint_config": [1, 3]}, cam, capacity=4096, max_objects=64)
    with torch.no_grad():
        out = pipe.forward_device(x[:2])
    assert int(out["overflow"]) & ops.RANGE_OVERFLOW
    with pytest.raises(ops.OkpError):
        pipe.objects(out, 0)
    ok = pp.BatchedKeypointPipeline(good, {"keypoint_config": [1, 3]}, cam, capacity=4096, max_objects=64)
    with torch.no_grad():
        assert not int(ok.forward_device(x[:2])["overflow"]) & ops.RANGE_OVERFLOW
    # the mixed configuration shares the guard (its fp16 branches saturate the same way)
    mixed = _net("float32mix", big)
    with torch.no_grad(), pytest.raises(ops.OkpError):
        mixed.deployed(x[:2])
    # a frame value beyond the range - or not a number - is caught by the stem itself (exact check where data enters)
    for bad_value in (1.0e5, float("nan"), -float("inf")):
        y = x[:2].clone()
        y[1, 2, 100, 100] = bad_value
        with torch.no_grad(), pytest.raises(ops.OkpError):
            good.deployed(y)
    # ... and so is what an fp16 sub-network hands back to the split-product stream (okp_cast to fp32, okp_add_f16_f32)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    h = ops.Act(torch.full((1, 4, 4, 8), float("inf"), dtype=torch.float16, device="cuda"))
    with ops.f32_split(True, True, flag):
        ops.cast(h, torch.float32)
    assert int(flag) == 1
    flag.zero_()
    with ops.f32_split(True, True, flag):
        ops.add_f16_f32(ops.Act(torch.full((1, 4, 4, 8), float("nan"), dtype=torch.float16, device="cuda")), ops.Act(torch.zeros((1, 4, 4, 8), device="cuda")))
        ops.cast(ops.Act(torch.ones((1, 4, 4, 8), dtype=torch.float16, device="cuda")), torch.float32)
    assert int(flag) == 1
    flag.zero_()
    with ops.f32_split(True, True, flag):
        ops.cast(ops.Act(torch.ones((1, 4, 4, 8), dtype=torch.float16, device="cuda")), torch.float32)
        ops.add_f16_f32(ops.Act(torch.ones((1, 4, 4, 8), dtype=torch.float16, device="cuda")), ops.Act(torch.zeros((1, 4, 4, 8), device="cuda")))
    assert int(flag) == 0
    with torch.no_grad():
        c = good.deployed(x)                                        # the flag is per pass: the next healthy pass is clean, same bits as before
    assert all(torch.equal(u, v) for u, v in zip(a, c))


def test_load_audits_float32x3_for_its_range():
    """load_keypoint_net audits a split-product configuration at load: float32x3 on weights that leave the fp16 range falls back to exact
    float32 with a warning; the audit records where its frames came from and the batch it ran at (the kernels a deployment runs)."""
    from object_keypoints_amd.perception import pipeline as pp
    sd = _state()
    net = pp.load_keypoint_net(sd, compute_dtype="float32x3")
    assert net.configuration() == "float32x3" and net.audit["checked"] and net.audit["frames_source"] == "synthetic" and net.audit["batch"] == pp.AUDIT_BATCH
    assert net.audit["report"]["range_ok"] and not net.audit["fell_back"]
    big = dict(sd)
    big["backbone.pre.1.bn2.weight"] = sd["backbone.pre.1.bn2.weight"] * 3.0e5
    with pytest.warns(RuntimeWarning, match="falling back to exact float32"):
        fb = pp.load_keypoint_net(big, compute_dtype="float32x3")
    assert fb.configuration() == "float32" and fb.audit["fell_back"] and fb.audit["fell_back_to"] == "float32"
    with pytest.warns(RuntimeWarning):
        fm = pp.load_keypoint_net(big, compute_dtype="float32mix")
    assert fm.configuration() == "float32" and fm.audit["frames_source"] == "synthetic"
    mix = pp.load_keypoint_net(sd, compute_dtype="float32mix", audit_frames=_frames(2).cpu())
    assert mix.audit["frames_source"] == "caller" and mix.audit["batch"] == 2
