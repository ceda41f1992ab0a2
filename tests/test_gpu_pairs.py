"""Pair-format tensors between split-product 3x3 convolutions (okp_conv_args.src_pairs / out_pairs, ABI 6; backbone.PAIR_TENSORS).

A pair-format tensor keeps the geometry of an fp32 tensor but holds [8 x fp16 hi | 8 x fp16 lo] per 8 channels - exactly what the
patch-resident split-product kernel (okp_igemm_patch_x3.hip) makes of 8 fp32 values in LDS.  The format changes WHERE the split happens
(once per element in the producer's epilogue instead of on every landed patch), never a result: every test here asks for bit equality
with the fp32-tensor path on the same operands."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed):
    from object_keypoints_amd import synth
    return torch.from_numpy(synth.normal_like(f"pairs{seed}", shape, seed))


def _act(x):
    from object_keypoints_amd import ops
    return ops.Act.from_nchw(x.cuda(), torch.float32)


def _pairs(a):
    from object_keypoints_amd import ops
    return ops.Act.float_to_pairs(a.t)


def test_pair_format_round_trip_matches_the_kernel_split():
    """Act.float_to_pairs (torch) is the split the kernels perform: hi = fp16(x), lo = fp16(x - hi); hi + lo is x to 2^-22."""
    from object_keypoints_amd import ops
    x = _act(_rand((2, 64, 16, 16), 1) * 3.0)
    p = _pairs(x)
    assert p.pairs and p.t.shape == x.t.shape and p.t.dtype == torch.float32
    back = p.pairs_to_float()
    assert float((back.t - x.t).abs().max()) <= 2.0 ** -21 * float(x.t.abs().max())
    hl = p.t.view(torch.float16).view(2, 16, 16, 8, 2, 8)
    assert torch.equal(hl[..., 0, :].reshape(2, 16, 16, 64), x.t.half())


@pytest.mark.parametrize("cin,cout,n,h,w,stride,res", [
    (32, 256, 1, 16, 16, 1, False),
    (64, 256, 2, 32, 48, 1, True),
    (96, 512, 3, 16, 32, 1, False),
    (32, 256, 2, 16, 32, 2, False),     # stride 2: four parity patches per chunk, none of them split
    (128, 256, 1, 32, 16, 2, True),
    (256, 256, 2, 16, 16, 1, True),
])
def test_pair_source_and_pair_output_are_bit_identical_to_fp32_tensors(cin, cout, n, h, w, stride, res):
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    x = _rand((n, cin, h * stride, w * stride), 2)
    wt = _rand((cout, cin, 3, 3), 3) / np.sqrt(cin * 9)
    b = _rand((cout,), 4) * 0.1
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, [cin], [stride], cout, conv_taps(wt.numpy()), b.numpy(), relu=True)
    xa = _act(x)
    ra = _act(_rand((n, cout, h, w), 5)) if res else None
    ref = ops.Act.empty(n, h, w, cout, torch.float32, xa.t.device)
    plan([xa], ref, h, w, res=ra, tile=13)
    # (a) pair-format source: the patches arrive split
    got = ops.Act.empty(n, h, w, cout, torch.float32, xa.t.device)
    plan([_pairs(xa)], got, h, w, res=ra, tile=13)
    assert not got.pairs and torch.equal(got.t, ref.t)
    # (b) pair-format output: the epilogue splits what it would have stored
    outp = ops.Act.empty(n, h, w, cout, torch.float32, xa.t.device)
    plan([xa], outp, h, w, res=ra, tile=13, out_pairs=True)
    assert outp.pairs and torch.equal(outp.t.view(torch.int32), _pairs(ref).t.view(torch.int32))
    # (c) both
    outp2 = ops.Act.empty(n, h, w, cout, torch.float32, xa.t.device)
    plan([_pairs(xa)], outp2, h, w, res=ra, tile=13, out_pairs=True)
    assert torch.equal(outp2.t.view(torch.int32), outp.t.view(torch.int32))


@pytest.mark.parametrize("mask", [1, 2, 3])
def test_pair_format_per_source_of_a_two_source_plan(mask):
    """conv2 + projected 1x1/s2 skip of `residual` (py_utils/utils.py:177-185): either source, or both, in pair format."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    n, c0, c1, cout, h, w = 2, 64, 96, 256, 32, 16
    t = _act(_rand((n, c0, h, w), 6)); x = _act(_rand((n, c1, 2 * h, 2 * w), 7))
    w2 = _rand((cout, c0, 3, 3), 8) / np.sqrt(c0 * 9); ws = _rand((cout, c1, 1, 1), 9) / np.sqrt(c1)
    taps = conv_taps(w2.numpy()) + [(1, 0, 0, np.ascontiguousarray(ws.numpy()[:, :, 0, 0]))]
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, [c0, c1], [1, 2], cout, taps, (_rand((cout,), 10) * 0.1).numpy(), relu=True)
    ref = ops.Act.empty(n, h, w, cout, torch.float32, t.t.device)
    plan([t, x], ref, h, w, tile=13)
    got = ops.Act.empty(n, h, w, cout, torch.float32, t.t.device)
    plan([_pairs(t) if mask & 1 else t, _pairs(x) if mask & 2 else x], got, h, w, tile=13, out_pairs=True)
    assert torch.equal(got.t.view(torch.int32), _pairs(ref).t.view(torch.int32))


def test_pair_output_of_the_transposed_convolution_classes():
    """unpool_merge (CornerNet_Squeeze.py:35-36): four sub-pixel classes with the up1 residual, written in pair format."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone
    torch.manual_seed(3)
    m = backbone.unpool_merge(256).cuda().eval()
    low = _act(_rand((4, 256, 16, 16), 11)); up1 = _act(_rand((4, 256, 32, 32), 12))
    with ops.f32_split():
        plan = m._build(torch.float32)
    ref = ops.Act.empty(4, 32, 32, 256, torch.float32, low.t.device)
    plan([low], ref, 16, 16, res=up1, out_step=2, n_classes=4, tile=13)
    got = ops.Act.empty(4, 32, 32, 256, torch.float32, low.t.device)
    plan([low], got, 16, 16, res=up1, out_step=2, n_classes=4, tile=13, out_pairs=True)
    assert torch.equal(got.t.view(torch.int32), _pairs(ref).t.view(torch.int32))


@pytest.mark.parametrize("n,h,w", [(2, 64, 64), (1, 37, 51), (3, 511, 511)])
def test_stem_pair_output_is_the_split_of_its_fp32_output(n, h, w):
    """okp_stem_forward_nchw_pairs: the split-product stem writing pair format = the pairs of what okp_stem_forward_nchw writes (odd sizes:
    partial tiles at the right / lower edge)."""
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception import backbone
    torch.manual_seed(5)
    m = backbone.convolution(7, 3, 128, stride=2).cuda().eval()
    frames = torch.from_numpy(synth.normal_like("stem_pairs", (n, 3, h, w), 15)).cuda()
    with ops.f32_split():
        plan = m._build(torch.float32, stride="direct")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ref = ops.Act.empty(n, ho, wo, 128, torch.float32, frames.device)
    plan.from_nchw(frames, ref)
    got = ops.Act(torch.zeros(n, ho, wo, 128, dtype=torch.float32, device=frames.device))
    plan.from_nchw(frames, got, out_pairs=True)
    assert got.pairs and torch.equal(got.t.view(torch.int32), _pairs(ref).t.view(torch.int32))


def test_pair_format_is_refused_everywhere_else():
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    from object_keypoints_amd._lib import OkpError
    cin = cout = 256
    wt = _rand((cout, cin, 3, 3), 13) / np.sqrt(cin * 9)
    x = _act(_rand((1, cin, 16, 16), 14))
    out = ops.Act.empty(1, 16, 16, cout, torch.float32, x.t.device)
    with ops.f32_split():
        split_plan = ops.ConvPlan(torch.float32, [cin], [1], cout, conv_taps(wt.numpy()), None, relu=True)
    exact_plan = ops.ConvPlan(torch.float32, [cin], [1], cout, conv_taps(wt.numpy()), None, relu=True)
    with pytest.raises(OkpError, match="patch-resident"):
        split_plan([_pairs(x)], out, 16, 16, tile=3)                  # a gather tile does not read pairs
    with pytest.raises(OkpError, match="patch-resident"):
        split_plan([x], out, 16, 16, out_pairs=True)                   # the heuristic's tile for one frame is not 13
    assert not out.pairs
    with pytest.raises(OkpError, match="split-product"):
        exact_plan([_pairs(x)], out, 16, 16)
    with pytest.raises(OkpError, match="residual stays float32"):
        split_plan([x], out, 16, 16, res=_pairs(out), tile=13)
    with pytest.raises(OkpError, match="pair format"):
        ops.cast(_pairs(x), torch.float16)
    with pytest.raises(OkpError, match="pair format"):
        _pairs(x).view()
    with pytest.raises(OkpError, match="8-channel groups"):
        _pairs(x).slice(4, 8)
    assert _pairs(x).slice(8, 16).pairs


def _x3_net(k=3, seed=3):
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=k, compute_dtype=ops.F32X3)
    shapes = {kk: tuple(v.shape) for kk, v in net.state_dict().items()}
    net.load_state_dict({kk: torch.from_numpy(np.array(v)) for kk, v in synth.fill_state_dict(shapes, seed=seed).items()})
    return net.eval().cuda()


@pytest.mark.parametrize("n,h,w,k", [(2, 64, 64, 3), (1, 13, 9, 4), (3, 8, 8, 3), (64, 64, 64, 3)])
def test_split_product_heads_in_one_launch(n, h, w, k):
    """okp_heads_forward on split-product plans (okp_heads_x3_kernel: pair-format x, three-term products, h1 / h2 in LDS) against the
    three-launch path of the same plans on the same values (256 -> 384, 384 -> 96, pointwise output), and - first, middle and last frame - against
    the oracle's prediction modules on the CPU."""
    from object_keypoints_amd import ops
    from oracle import net as onet
    net = _x3_net(k)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    cnv = ops.Act(torch.randn((n, h, w, 256), generator=gen, device="cuda") * 0.7)
    l0 = ops.COUNTERS["launches"]
    fused = net._run_heads(1, _pairs(cnv), sigmoid=True)
    assert ops.COUNTERS["launches"] - l0 == 1
    ref = net._run_heads(1, cnv, sigmoid=True)
    assert ops.COUNTERS["launches"] - l0 == 1 + 3                       # 256 -> 384, 384 -> 96, the pointwise output kernel
    for a, b in zip(fused, ref):
        assert a.shape == b.shape and a.dtype == torch.float32
        assert float((a - b).abs().max()) <= 2e-5 * (1.0 + float(b.abs().max()))
    oracle = onet.load_synthetic(onet.KeypointNet(features=128, heatmaps_out=k), seed=3)
    for i in sorted({0, n // 2 - 1 if n > 2 else 0, n - 1}):             # (64 frames: frames 0, 31 and 63)
        x = cnv.t[i:i + 1].permute(0, 3, 1, 2).cpu()
        with torch.no_grad():
            rh = torch.sigmoid(oracle.heatmap_head.output_head2(x)); rd = oracle.depth_head.output_head2(x); rc = oracle.center_head.output_head2(x)
        assert float((fused[0][i:i + 1].cpu() - rh).abs().max()) <= 2e-5
        assert float((fused[1][i:i + 1].cpu() - rd).abs().max()) <= 2e-5 * (1.0 + float(rd.abs().max()))
        assert float((fused[2][i:i + 1].cpu().reshape(rc.shape) - rc).abs().max()) <= 2e-5 * (1.0 + float(rc.abs().max()))


def test_split_product_heads_refuse_an_fp32_activation_and_16_bit_heads_a_pair_one():
    from object_keypoints_amd import ops
    from object_keypoints_amd._lib import OkpError
    net = _x3_net()
    cnv = ops.Act(torch.randn((1, 8, 8, 256), device="cuda"))
    with ops.f32_split():
        l1, l2, w3, b3 = net._build_heads(1, torch.float32, cnv.t.device)
    out = torch.empty((1, 3, 8, 8), device="cuda")
    with pytest.raises(OkpError, match="pair-format"):
        ops.heads_fused(l1, l2, cnv, [(0, ops.ACT_NONE, out, 0)], w3, b3)


@pytest.mark.parametrize("n", [16, 5, 64, 65])
def test_network_with_pair_tensors_is_bit_identical(n):
    """The float32x3 network with backbone.PAIR_TENSORS on / off: same bits out.  At 16 frames the 64 x 64 convolutions fill the chip with
    patch-kernel tiles and the pair format is in use (counted); at 5 frames the heuristic keeps some of them on gather tiles and the
    tensors around those stay fp32; 64 frames is the bench's batch (until round 6: the stem and pre[1] in two frame chunks)."""
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception import backbone
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=ops.F32X3)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
    net.eval().cuda()
    x = torch.from_numpy(synth.frames(n, seed=4)).cuda()
    outs = {}
    keep_heads = ops.FUSE_HEADS_X3
    ops.FUSE_HEADS_X3 = False           # (the one-launch heads sum in another order: compared in test_network_with_fused_split_heads)
    try:
        for on in (True, False):
            backbone.PAIR_TENSORS = on
            before = ops.COUNTERS.get("pair_outputs", 0)
            with torch.no_grad():
                outs[on] = [o.clone() for o in net.deployed(x)]
            used = ops.COUNTERS.get("pair_outputs", 0) - before
            if on and n >= 16:
                # the stem, conv1 and the result of pre[1], conv1 of pre[2], the two hourglasses' merged maps, conv1 of inters[0] - once each
                # at 64 and 65 frames too (round 6: the stem kernel and the patch-resident kernel address the 2.1 GB map frame by frame)
                assert used == 7
            if not on:
                assert used == 0
    finally:
        backbone.PAIR_TENSORS = True
        ops.FUSE_HEADS_X3 = keep_heads
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)


def test_network_with_fused_split_heads():
    """float32x3 network at 16 frames: the last `cnvs` convolution writes pairs and the heads are one launch; outputs within the
    split-product tolerance of the three-launch heads (same products, sums in another order)."""
    from object_keypoints_amd import ops, synth
    net = _x3_net(seed=0)
    x = torch.from_numpy(synth.frames(16, seed=4)).cuda()
    outs = {}
    keep = ops.FUSE_HEADS_X3
    try:
        for on in (True, False):
            ops.FUSE_HEADS_X3 = on
            l0 = ops.COUNTERS["launches"]
            with torch.no_grad():
                outs[on] = [o.clone() for o in net.deployed(x)]
            outs[on, "launches"] = ops.COUNTERS["launches"] - l0
    finally:
        ops.FUSE_HEADS_X3 = keep
    assert outs[False, "launches"] - outs[True, "launches"] == 2
    for a, b in zip(outs[True], outs[False]):
        assert float((a - b).abs().max()) <= 2e-5 * (1.0 + float(b.abs().max()))
