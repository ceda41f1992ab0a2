"""HIP implicit-GEMM convolution vs a plain PyTorch fp32 reference of the same op (CPU conv2d),
through the C ABI.  Covers all three tile shapes, both dtypes, partial tiles, stride 2,
two-source plans (fused projected skip), residual epilogue and sub-pixel output placement."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


HALF = [torch.bfloat16, torch.float16]             # the two 16-bit precisions: one kernel source, two instantiations
ALL = [torch.float32] + HALF


def _tol(dtype, ref):
    """fp32: accumulation-order noise; bf16 (8 mantissa bits) and fp16 (11): output rounding + accumulated input rounding."""
    scale = float(ref.abs().max())
    if dtype == torch.float32:
        return 2e-4 + 1e-5 * scale
    return (0.02 if dtype == torch.bfloat16 else 0.003) * (scale + 1.0)


def _q(v, dtype):
    """Round to the compute dtype and back: references are evaluated on the same rounded operands."""
    return v.to(dtype).float()


def _rand(shape, seed):
    from object_keypoints_amd import synth
    return torch.from_numpy(synth.normal_like(f"convtest{seed}", shape, seed))


@pytest.mark.parametrize("dtype", ALL)
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("k,stride,cin,cout,n,h,w", [
    (3, 1, 64, 256, 2, 20, 24),     # many K slices, several co tiles
    (3, 2, 32, 64, 3, 17, 15),      # stride 2, odd sizes
    (1, 1, 192, 96, 2, 9, 7),       # 1x1, cout not a multiple of 64
    (1, 2, 16, 8, 1, 8, 8),         # tiny channel counts (partial K slice)
    (3, 1, 8, 16, 1, 5, 5),         # one partial slice per tap
])
def test_conv_matches_torch(dtype, tile, k, stride, cin, cout, n, h, w):
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps, conv_out_size
    dev = _dev()
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, k, k), 2) * (1.0 / np.sqrt(cin * k * k))
    b = _rand((cout,), 3) * 0.1
    pad = (k - 1) // 2
    x, wt = _q(x, dtype), _q(wt, dtype)      # compare against the same rounded operands
    ref = F.relu(F.conv2d(x, wt, b, stride=stride, padding=pad))
    plan = ops.ConvPlan(dtype, [cin], [stride], cout, conv_taps(wt.numpy()), b.numpy(), relu=True)
    xa = ops.Act.from_nchw(x.to(dev), dtype)
    ho, wo = conv_out_size(h, k, stride, pad), conv_out_size(w, k, stride, pad)
    out = ops.Act.empty(n, ho, wo, cout, dtype, dev)
    plan([xa], out, ho, wo, tile=tile)
    got = out.to_nchw().cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    assert err <= _tol(dtype, ref), f"max err {err}"


@pytest.mark.parametrize("dtype", ALL)
def test_two_source_residual_block(dtype):
    """relu(conv3x3(t) + conv1x1_s2(x) + bias): the fused conv2 + projected skip of `residual`."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    dev = _dev()
    n, c0, c1, cout, h, w = 2, 32, 16, 64, 10, 12
    t = _rand((n, c0, h, w), 4); x = _rand((n, c1, 2 * h, 2 * w), 5)
    w2 = _rand((cout, c0, 3, 3), 6) / np.sqrt(c0 * 9); ws = _rand((cout, c1, 1, 1), 7) / np.sqrt(c1)
    b = _rand((cout,), 8) * 0.1
    t, x, w2, ws = [_q(v, dtype) for v in (t, x, w2, ws)]
    ref = F.relu(F.conv2d(t, w2, b, padding=1) + F.conv2d(x, ws, stride=2))
    taps = conv_taps(w2.numpy()) + [(1, 0, 0, np.ascontiguousarray(ws.numpy()[:, :, 0, 0]))]
    plan = ops.ConvPlan(dtype, [c0, c1], [1, 2], cout, taps, b.numpy(), relu=True)
    out = ops.Act.empty(n, h, w, cout, dtype, dev)
    plan([ops.Act.from_nchw(t.to(dev), dtype), ops.Act.from_nchw(x.to(dev), dtype)], out, h, w)
    got = out.to_nchw().cpu()
    assert float((got - ref).abs().max()) <= _tol(dtype, ref)


@pytest.mark.parametrize("dtype", ALL)
def test_residual_epilogue_and_channel_window(dtype):
    """1x1 conv writing channels [8,24) of a 32-channel tensor with a residual read from a window."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    dev = _dev()
    n, cin, cout, h, w = 2, 24, 16, 6, 7
    x = _rand((n, cin, h, w), 9); r = _rand((n, 32, h, w), 10)
    wt = _rand((cout, cin, 1, 1), 11) / np.sqrt(cin)
    x, r, wt = [_q(v, dtype) for v in (x, r, wt)]
    ref = F.relu(F.conv2d(x, wt) + r[:, 8:24])
    plan = ops.ConvPlan(dtype, [cin], [1], cout, conv_taps(wt.numpy()), None, relu=True)
    big = ops.Act(torch.full((n, h, w, 32), -7.0, dtype=dtype, device=dev))
    ra = ops.Act.from_nchw(r.to(dev), dtype)
    plan([ops.Act.from_nchw(x.to(dev), dtype)], big.slice(8, 16), h, w, res=ra.slice(8, 16))
    full = big.t.float().cpu()
    got = full[..., 8:24].permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= _tol(dtype, ref)
    assert bool((full[..., :8] == -7.0).all()) and bool((full[..., 24:] == -7.0).all())   # neighbours untouched


@pytest.mark.parametrize("dtype", ALL)
def test_conv_transpose_as_subpixel_convs(dtype):
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import unpool_merge
    dev = _dev()
    n, c, h, w = 2, 16, 5, 7
    m = unpool_merge(c).eval()
    with torch.no_grad():
        m.weight.copy_(_rand((c, c, 4, 4), 12) / np.sqrt(4 * c)); m.bias.copy_(_rand((c,), 13) * 0.1)
    low = _rand((n, c, h, w), 14); up1 = _rand((n, c, 2 * h, 2 * w), 15)
    if dtype != torch.float32:
        low, up1 = _q(low, dtype), _q(up1, dtype)
        wq = _q(m.weight.detach(), dtype)
    else:
        wq = m.weight.detach()
    ref = up1 + F.conv_transpose2d(low, wq, m.bias.detach(), stride=2, padding=1)
    got = m(ops.Act.from_nchw(low.to(dev), dtype), ops.Act.from_nchw(up1.to(dev), dtype)).to_nchw().cpu()
    assert float((got - ref).abs().max()) <= _tol(dtype, ref)


def test_bad_arguments_are_reported():
    from object_keypoints_amd import ops
    with pytest.raises(ops.OkpError):
        ops.ConvPlan(torch.float32, [6], [1], 8, [(0, 0, 0, np.zeros((8, 6), np.float32))])   # cin not 16-byte multiple
    with pytest.raises(ops.OkpError):
        ops.Act(torch.zeros(1, 2, 2, 8))                                                        # CPU tensor: no fallback


@pytest.mark.parametrize("cin,cout,h,w,stride,n", [
    (256, 256, 16, 16, 1, 3),      # skip, strips of full rows
    (256, 256, 64, 64, 1, 1),      # skip, 2-D tiles with halo
    (384, 512, 8, 8, 2, 3),        # stride 2 to 4x4, <384,256>
    (512, 512, 4, 4, 1, 7),        # 4x4 maps (the network leaves these to two launches / the chain kernel)
    (384, 384, 13, 9, 1, 2),       # odd sizes
    (384, 256, 16, 16, 1, 2),      # no skip (cin != cout), <384,128>
    (256, 256, 33, 21, 1, 2),      # streaming kernel (okp_fire2): odd sizes, partial tiles on both axes
    (256, 256, 7, 5, 1, 3),        # streaming kernel: one small tile per frame
    (256, 256, 32, 32, 1, 4),      # streaming kernel: the 32x32 hourglass level
    (512, 384, 8, 8, 1, 2),        # streaming kernel <512,192>: no skip, streamed squeeze weights, 6 waves
    (384, 384, 16, 16, 1, 5),      # streaming kernel <384,192>: the 16x16 level
    (512, 512, 8, 8, 1, 3),        # streaming kernel <512,256>: 8 waves
    (256, 256, 64, 64, 2, 1),      # streaming kernel, stride 2: 64x64 -> 32x32
    (256, 256, 21, 35, 2, 2),      # stride 2, odd sizes (partial tiles)
    (384, 384, 16, 16, 2, 3),      # stride 2 <384,192>
])
@pytest.mark.parametrize("dtype", HALF)
def test_fused_fire_module_matches_oracle(cin, cout, h, w, stride, n, dtype):
    """One-launch 16-bit fire module vs the oracle's fire_module (fp32) on the same rounded input, and vs the
    three-launch HIP path, which it must reproduce up to the rounding of the squeeze tensor."""
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    dev = _dev()
    o = onet.load_synthetic(onet.fire_module(cin, cout, stride=stride), seed=21)
    m = bb.fire_module(cin, cout, stride=stride)
    m.load_state_dict(o.state_dict())
    m.eval()
    x = _q(_rand((n, cin, h, w), 31), dtype)
    with torch.no_grad():
        ref = o(x)
    xa = ops.Act.from_nchw(x.to(dev), dtype)
    keep_hw, ops.FUSE_FIRE_MIN_HW = ops.FUSE_FIRE_MIN_HW, 0
    try:
        l0 = ops.COUNTERS["launches"]
        got = m(xa).to_nchw().cpu()
        assert ops.COUNTERS["launches"] - l0 == 1
    finally:
        ops.FUSE_FIRE_MIN_HW = keep_hw
    keep, ops.FUSE_FIRE = ops.FUSE_FIRE, False
    try:
        l0 = ops.COUNTERS["launches"]
        unfused = m(xa).to_nchw().cpu()
        assert ops.COUNTERS["launches"] - l0 == 2
    finally:
        ops.FUSE_FIRE = keep
    scale = float(ref.abs().max())
    assert got.shape == ref.shape
    err = (got - ref).abs()
    eps = 1.0 if dtype == torch.bfloat16 else 0.15
    assert float(err.max()) <= eps * (0.03 * scale + 0.02), f"max err {float(err.max())} scale {scale}"
    assert float((got - unfused).abs().max()) <= eps * (0.02 * scale + 0.02)


@pytest.mark.parametrize("dtype", HALF)
def test_streaming_fire_without_skip(dtype):
    """okp_fire2 with the skip connection switched off (the reference module always has it at 256 -> 256; the C ABI
    takes it as an argument)."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    dev = _dev()
    o = onet.load_synthetic(onet.fire_module(256, 256), seed=22)
    m = bb.fire_module(256, 256)
    m.load_state_dict(o.state_dict())
    m.eval()
    o.skip = False
    m.skip = False
    x = _q(_rand((2, 256, 19, 23), 32), dtype)
    with torch.no_grad():
        ref = o(x)
    l0 = ops.COUNTERS["launches"]
    got = m(ops.Act.from_nchw(x.to(dev), dtype)).to_nchw().cpu()
    assert ops.COUNTERS["launches"] - l0 == 1
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= (1.0 if dtype == torch.bfloat16 else 0.15) * (0.03 * scale + 0.02)


@pytest.mark.parametrize("dtype", ALL)
@pytest.mark.parametrize("c,h,w,stride,n", [(128, 16, 16, 1, 2), (192, 9, 7, 1, 1), (64, 12, 12, 2, 2), (256, 8, 20, 1, 1)])
def test_standalone_depthwise_kernel(dtype, c, h, w, stride, n):
    """okp_dwconv3x3_forward (sliding-window kernel) vs torch depth-wise conv + bias + residual + ReLU."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_out_size
    dev = _dev()
    x = _rand((n, c, h, w), 41); wt = _rand((c, 1, 3, 3), 42) / 3.0; b = _rand((c,), 43) * 0.1
    ho, wo = conv_out_size(h, 3, stride, 1), conv_out_size(w, 3, stride, 1)
    r = _rand((n, c, ho, wo), 44)
    x, r = _q(x, dtype), _q(r, dtype)
    ref = F.relu(F.conv2d(x, wt, b, stride=stride, padding=1, groups=c) + r)
    wd = torch.from_numpy(np.ascontiguousarray(np.transpose(wt.numpy()[:, 0], (1, 2, 0)).reshape(9, c))).to(dev)
    out = ops.Act.empty(n, ho, wo, c, dtype, dev)
    ops.dwconv3x3(ops.Act.from_nchw(x.to(dev), dtype), wd, b.to(dev), out, stride, res=ops.Act.from_nchw(r.to(dev), dtype), relu=True)
    got = out.to_nchw().cpu()
    assert float((got - ref).abs().max()) <= _tol(dtype, ref)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("n,h,w", [(2, 64, 64), (1, 37, 45), (3, 511, 511), (2, 15, 130)])
def test_stem_kernel_matches_torch(n, h, w, dtype):
    """okp_stem_forward (dedicated bf16 7x7/s2 kernel) vs conv2d + bias + relu on the same bf16-rounded operands,
    including partial tiles (output sizes that are not multiples of 8 x 32)."""
    from object_keypoints_amd import ops
    dev = _dev()
    x = _q(_rand((n, 3, h, w), 11), dtype)
    wt = _q(_rand((128, 3, 7, 7), 12) * (1.0 / np.sqrt(147.0)), dtype)
    b = _rand((128,), 13) * 0.1
    ref = F.relu(F.conv2d(x, wt, b, stride=2, padding=3))
    plan = ops.StemPlan(wt.numpy(), b.numpy(), dtype)
    packed = ops.pack_frames(x.to(dev), dtype)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = ops.Act.empty(n, ho, wo, 128, dtype, dev)
    out.t.fill_(float("nan"))
    plan(packed, out)
    got = out.to_nchw().float().cpu()
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err <= _tol(dtype, ref), f"max err {err}"
    # the variant that reads the fp32 NCHW frames itself (no packing pass) produces the same bits
    out2 = ops.Act.empty(n, ho, wo, 128, dtype, dev)
    out2.t.fill_(float("nan"))
    plan.from_nchw(x.to(dev), out2)
    assert torch.equal(out2.t, out.t)


def test_stem_kernel_rejects_bad_views():
    from object_keypoints_amd import ops
    dev = _dev()
    plan = ops.StemPlan(np.zeros((128, 3, 7, 7), np.float32), np.zeros(128, np.float32))
    packed = ops.pack_frames(torch.zeros(1, 3, 33, 33, device=dev), torch.bfloat16)
    with pytest.raises(ops.OkpError):
        plan(packed, ops.Act.empty(1, 16, 17, 128, torch.bfloat16, dev))       # wrong output size
    with pytest.raises(ops.OkpError):
        plan(packed, ops.Act.empty(1, 17, 17, 64, torch.bfloat16, dev))        # too few channels
    with pytest.raises(ops.OkpError):
        ops.StemPlan(np.zeros((64, 3, 7, 7), np.float32), np.zeros(64, np.float32))


@pytest.mark.parametrize("c,h,w,n,count", [(512, 4, 4, 5, 6), (512, 3, 4, 2, 2), (512, 2, 2, 3, 8), (512, 1, 3, 1, 3),
                                             (384, 8, 8, 3, 2), (384, 7, 5, 2, 3), (384, 4, 4, 2, 2), (384, 8, 3, 1, 4), (384, 8, 8, 3, 1), (512, 4, 4, 2, 1)])
@pytest.mark.parametrize("dtype", HALF)
def test_fire_chain_matches_module_by_module(c, h, w, n, count, dtype):
    """okp_fire_chain_forward (activations resident in LDS across `count` fire(512, 512) modules) against the oracle's
    modules applied one by one (fp32) and against the product's own module-by-module path."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    dev = _dev()
    omods = [onet.load_synthetic(onet.fire_module(c, c), seed=30 + i) for i in range(count)]
    mods = []
    for o in omods:
        m = bb.fire_module(c, c)
        m.load_state_dict(o.state_dict())
        mods.append(m.eval())
    x = _q(_rand((n, c, h, w), 77), dtype)
    ref = x
    with torch.no_grad():
        for o in omods:
            ref = o(ref)
    xa = ops.Act.from_nchw(x.to(dev), dtype)
    keep = ops.FUSE_FIRE_CHAIN
    try:
        ops.FUSE_FIRE_CHAIN = True
        l0 = ops.COUNTERS["launches"]
        got = bb.run_fire_modules(mods, xa).to_nchw().float().cpu()
        assert ops.COUNTERS["launches"] - l0 == 1
        ops.FUSE_FIRE_CHAIN = False
        single = bb.run_fire_modules(mods, xa).to_nchw().float().cpu()
    finally:
        ops.FUSE_FIRE_CHAIN = keep
    scale = float(ref.abs().max())
    assert got.shape == ref.shape
    eps = 1.0 if dtype == torch.bfloat16 else 0.15
    assert float((got - ref).abs().max()) <= eps * (0.03 * scale * max(1, count // 2) + 0.02), float((got - ref).abs().max())
    assert float((got - single).abs().max()) <= eps * (0.02 * scale * max(1, count // 2) + 0.02)


@pytest.mark.parametrize("h,w,n,count", [(8, 8, 5, 6), (8, 8, 1, 1), (7, 5, 3, 2), (2, 3, 2, 6), (1, 1, 2, 1)])
@pytest.mark.parametrize("dtype", HALF)
def test_framed_fire_chain_matches_module_by_module(h, w, n, count, dtype):
    """The innermost hourglass level as ONE launch (entry / exit form of okp_fire_chain_forward): stride-2 fire(384, 512) on an
    h x w map, `count` fire(512, 512), fire(512, 384) - against the oracle's modules applied one by one (fp32) and against the
    product's own module-by-module path (hg_module n = 1: low1, low2, low3; CornerNet_Squeeze.py:10-51, modules.py hg_module)."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    dev = _dev()
    shapes = [(384, 512, 2)] + [(512, 512, 1)] * count + [(512, 384, 1)]
    omods = [onet.load_synthetic(onet.fire_module(a, b, stride=st), seed=60 + i) for i, (a, b, st) in enumerate(shapes)]
    mods = []
    for o, (a, b, st) in zip(omods, shapes):
        m = bb.fire_module(a, b, stride=st)
        m.load_state_dict(o.state_dict())
        mods.append(m.eval())
    x = _q(_rand((n, 384, h, w), 78), dtype)
    ref = x
    with torch.no_grad():
        for o in omods:
            ref = o(ref)
    xa = ops.Act.from_nchw(x.to(dev), dtype)
    keep = ops.FUSE_FIRE_CHAIN_FRAMED
    try:
        ops.FUSE_FIRE_CHAIN_FRAMED = True
        l0 = ops.COUNTERS["launches"]
        m0 = ops.COUNTERS["macs"]
        got = bb.run_fire_modules(mods, xa).to_nchw().float().cpu()
        assert ops.COUNTERS["launches"] - l0 == 1
        macs_fused = ops.COUNTERS["macs"] - m0
        again = bb.run_fire_modules(mods, xa).to_nchw().float().cpu()
        ops.FUSE_FIRE_CHAIN_FRAMED = False
        m0 = ops.COUNTERS["macs"]
        single = bb.run_fire_modules(mods, xa).to_nchw().float().cpu()
        assert ops.COUNTERS["macs"] - m0 == macs_fused            # the FLOP accounting does not depend on the fusion
    finally:
        ops.FUSE_FIRE_CHAIN_FRAMED = keep
    assert got.shape == ref.shape == (n, 384, (h - 1) // 2 + 1, (w - 1) // 2 + 1)
    assert torch.equal(got, again)
    scale = float(ref.abs().max())
    eps = 1.0 if dtype == torch.bfloat16 else 0.15
    depth = max(1, (count + 2) // 2)
    assert float((got - ref).abs().max()) <= eps * (0.03 * scale * depth + 0.02), float((got - ref).abs().max())
    assert float((got - single).abs().max()) <= eps * (0.02 * scale * depth + 0.02)


@pytest.mark.parametrize("case", ["conv3x3", "conv3x3_res_window", "conv3x3_src_window", "residual_s2_skip", "two_chunks", "merge_1x1", "one_tile", "conv3x3_s2", "conv3x3_s2_odd_input"])
@pytest.mark.parametrize("dtype", HALF)
def test_patch_resident_kernel_matches_gather_kernel(case, dtype):
    """Tile 13 (okp_igemm_patch: input patch + halo resident in LDS) against torch AND bit-for-bit against tile 6 (same
    K order, same MFMA shape => identical sums): zero padding on all four edges, residual read through a channel window,
    the stride-2 single-tap second source, several channel chunks, stride-2 3x3 (parity-class patches)."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    dev = _dev()
    rb = lambda v: _q(v, dtype)
    res = None
    if case in ("conv3x3", "conv3x3_res_window", "conv3x3_src_window", "two_chunks", "one_tile"):
        n, h, w = (1, 16, 16) if case == "one_tile" else (3, 32, 48)
        cin = 128 if case == "two_chunks" else 256
        x = rb(_rand((n, cin, h, w), 21)); wt = rb(_rand((256, cin, 3, 3), 22) / np.sqrt(cin * 9)); b = _rand((256,), 23) * 0.1
        ref = F.conv2d(x, wt, b, padding=1)
        plan = ops.ConvPlan(dtype, [cin], [1], 256, conv_taps(wt.numpy()), b.numpy(), relu=True)
        srcs = [ops.Act.from_nchw(x.to(dev), dtype)]
        if case == "conv3x3_src_window":
            # the source is a 256-channel window of a 320-channel tensor (pixel stride 640 bytes, base 64 bytes in): the per-tile offset
            # table of the patch pixels (okp_igemm_patch.hip) is built from the view's stride and base
            wide = torch.cat([rb(_rand((n, 32, h, w), 44)), x, rb(_rand((n, 32, h, w), 45))], dim=1)
            srcs = [ops.Act.from_nchw(wide.to(dev), dtype).slice(32, 256)]
        if case == "conv3x3_res_window":
            r = rb(_rand((n, 320, h, w), 24))
            ref = ref + r[:, 32:288]
            res = ops.Act.from_nchw(r.to(dev), dtype).slice(32, 256)
        ref = F.relu(ref)
    elif case in ("conv3x3_s2", "conv3x3_s2_odd_input"):
        # stride 2: four patch geometries (the parity classes of the taps), 17x17 ... 16x16 pixels with a pixel step of 2
        n, h, w = 2, 32, 48
        hi, wi = (2 * h, 2 * w) if case == "conv3x3_s2" else (2 * h - 1, 2 * w - 1)
        cin = 256 if case == "conv3x3_s2" else 128
        x = rb(_rand((n, cin, hi, wi), 36)); wt = rb(_rand((256, cin, 3, 3), 37) / np.sqrt(cin * 9)); b = _rand((256,), 38) * 0.1
        ref = F.relu(F.conv2d(x, wt, b, stride=2, padding=1))
        assert ref.shape[2:] == (h, w)
        plan = ops.ConvPlan(dtype, [cin], [2], 256, conv_taps(wt.numpy()), b.numpy(), relu=True)
        srcs = [ops.Act.from_nchw(x.to(dev), dtype)]
    elif case == "residual_s2_skip":
        n, h, w = 2, 32, 32
        t = rb(_rand((n, 256, h, w), 25)); x = rb(_rand((n, 128, 2 * h, 2 * w), 26))
        w2 = rb(_rand((256, 256, 3, 3), 27) / np.sqrt(256 * 9)); ws = rb(_rand((256, 128, 1, 1), 28) / np.sqrt(128)); b = _rand((256,), 29) * 0.1
        ref = F.relu(F.conv2d(t, w2, b, padding=1) + F.conv2d(x, ws, stride=2))
        taps = conv_taps(w2.numpy()) + [(1, 0, 0, np.ascontiguousarray(ws.numpy()[:, :, 0, 0]))]
        plan = ops.ConvPlan(dtype, [256, 128], [1, 2], 256, taps, b.numpy(), relu=True)
        srcs = [ops.Act.from_nchw(t.to(dev), dtype), ops.Act.from_nchw(x.to(dev), dtype)]
    else:
        n, h, w = 2, 16, 32
        a = rb(_rand((n, 256, h, w), 30)); c = rb(_rand((n, 256, h, w), 31))
        wa = rb(_rand((256, 256, 1, 1), 32) / 16); wc = rb(_rand((256, 256, 1, 1), 33) / 16); b = _rand((256,), 34) * 0.1
        ref = F.relu(F.conv2d(a, wa, b) + F.conv2d(c, wc))
        taps = [(0, 0, 0, np.ascontiguousarray(wa.numpy()[:, :, 0, 0])), (1, 0, 0, np.ascontiguousarray(wc.numpy()[:, :, 0, 0]))]
        plan = ops.ConvPlan(dtype, [256, 256], [1, 1], 256, taps, b.numpy(), relu=True)
        srcs = [ops.Act.from_nchw(a.to(dev), dtype), ops.Act.from_nchw(c.to(dev), dtype)]
    outs = {}
    tiles = (6, 13, 3, 14)          # (14: the patch-resident kernel on 32x32x16 MFMAs, the MFMA-shape experiment: sums in the order of gather tile 3)
    for tile in tiles:
        big = ops.Act(torch.full((n, h, w, 288), -7.0, dtype=dtype, device=dev))
        plan(srcs, big.slice(16, 256), h, w, res=res, tile=tile)
        outs[tile] = big.t.float().cpu()
        assert bool((outs[tile][..., :16] == -7.0).all()) and bool((outs[tile][..., 272:] == -7.0).all())   # neighbours untouched
    for tile in tiles[1:]:
        got = outs[tile][..., 16:272].permute(0, 3, 1, 2)
        assert float((got - ref).abs().max()) <= _tol(dtype, ref), tile
    assert torch.equal(outs[13], outs[6])
    assert torch.equal(outs[14], outs[3])

@pytest.mark.gpu
def test_patch_resident_kernel_refuses_other_shapes():
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    dev = _dev()
    wt = _rand((256, 256, 3, 3), 35) * 0.01
    plan = ops.ConvPlan(torch.bfloat16, [256], [1], 256, conv_taps(wt.numpy()), None, relu=False)
    x = ops.Act(torch.zeros((1, 20, 16, 256), dtype=torch.bfloat16, device=dev))
    out = ops.Act.empty(1, 20, 16, 256, torch.bfloat16, dev)
    with pytest.raises(RuntimeError, match="tile 13"):
        plan([x], out, 20, 16, tile=13)                      # height not a multiple of 16


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("h,w", [(32, 16), (16, 16)])
def test_patch_resident_kernel_transposed_conv_classes(h, w, dtype):
    """The 4x4/s2 transposed convolution + hourglass merge as four sub-pixel classes of the patch-resident kernel (each class
    = 2x2 taps of the shared 18x18 patch, written at its output parity with `up1` added): against torch and bit-for-bit
    against the gather tile."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    dev = _dev()
    n, c = 2, 256
    m = bb.unpool_merge(c).eval()
    with torch.no_grad():
        m.weight.copy_(_rand((c, c, 4, 4), 40) / np.sqrt(4 * c)); m.bias.copy_(_rand((c,), 41) * 0.1)
    low = _q(_rand((n, c, h, w), 42), dtype); up1 = _q(_rand((n, c, 2 * h, 2 * w), 43), dtype)
    wq = _q(m.weight.detach(), dtype)
    ref = up1 + F.conv_transpose2d(low, wq, m.bias.detach(), stride=2, padding=1)
    la, ua = ops.Act.from_nchw(low.to(dev), dtype), ops.Act.from_nchw(up1.to(dev), dtype)
    outs = {}
    keep = bb.UNPOOL_TILE
    try:
        for tile in (6, 13):
            bb.UNPOOL_TILE = tile
            outs[tile] = m(la, ua).t.float().cpu()
    finally:
        bb.UNPOOL_TILE = keep
    for tile in (13,):
        got = outs[tile].permute(0, 3, 1, 2)
        assert float((got - ref).abs().max()) <= _tol(dtype, ref), tile
    assert torch.equal(outs[13], outs[6])


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", ["conv3x3_64", "conv3x3_s2_128", "residual_s2_skip_128"])
def test_patch_resident_kernel_full_size_matches_gather_kernel(case, dtype):
    """The bench's own shapes (64 frames; 1 024 to 4 096 tiles through the XCD-aware order, inputs of up to 1.07 GB):
    tile 13 bit-for-bit against tile 6 on device-generated data - the size-independent property available here, since
    both kernels add the same products in the same order."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    dev = _dev()
    g = torch.Generator(device=dev); g.manual_seed(7)
    rnd = lambda *shape: ops.Act(torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(dtype))
    n = 64
    if case == "conv3x3_64":
        h = w = 64
        wt = (_rand((256, 256, 3, 3), 50) / np.sqrt(256 * 9)).numpy()
        plan = ops.ConvPlan(dtype, [256], [1], 256, conv_taps(wt), _rand((256,), 51).numpy() * 0.1, relu=True)
        srcs, res = [rnd(n, h, w, 256)], rnd(n, h, w, 256)
    elif case == "conv3x3_s2_128":
        h = w = 128
        wt = (_rand((256, 128, 3, 3), 52) / np.sqrt(128 * 9)).numpy()
        plan = ops.ConvPlan(dtype, [128], [2], 256, conv_taps(wt), _rand((256,), 53).numpy() * 0.1, relu=True)
        srcs, res = [rnd(n, 2 * h, 2 * w, 128)], None
    else:
        h = w = 128
        w2 = (_rand((256, 256, 3, 3), 54) / np.sqrt(256 * 9)).numpy(); ws = (_rand((256, 128, 1, 1), 55) / np.sqrt(128)).numpy()
        taps = conv_taps(w2) + [(1, 0, 0, np.ascontiguousarray(ws[:, :, 0, 0]))]
        plan = ops.ConvPlan(dtype, [256, 128], [1, 2], 256, taps, _rand((256,), 56).numpy() * 0.1, relu=True)
        srcs, res = [rnd(n, h, w, 256), rnd(n, 2 * h, 2 * w, 128)], None
    out6 = ops.Act.empty(n, h, w, 256, dtype, dev)
    out13 = ops.Act.empty(n, h, w, 256, dtype, dev)
    plan(srcs, out6, h, w, res=res, tile=6)
    plan(srcs, out13, h, w, res=res, tile=13)
    torch.cuda.synchronize()
    assert torch.equal(out13.t, out6.t)
    assert bool(torch.isfinite(out13.t.float()).all()) and float(out13.t.float().abs().max()) > 0.5


@pytest.mark.parametrize("dtype", ALL)
def test_torch_ops_and_ctypes_bindings_launch_the_same_kernels(dtype):
    """The two host bindings of the C ABI (torch.ops.okp.* and ctypes) give bit-identical results: same plan, same args."""
    from object_keypoints_amd import _lib, ops
    from object_keypoints_amd.perception.backbone import conv_taps
    dev = _dev()
    assert _lib.torch_ops() is not None
    x = _q(_rand((2, 64, 20, 24), 61), dtype); r = _q(_rand((2, 128, 20, 24), 62), dtype)
    wt = _rand((128, 64, 3, 3), 63) / np.sqrt(64 * 9)
    plan = ops.ConvPlan(dtype, [64], [1], 128, conv_taps(wt.numpy()), np.zeros(128, np.float32), relu=True)
    xa, ra = ops.Act.from_nchw(x.to(dev), dtype), ops.Act.from_nchw(r.to(dev), dtype)
    outs = []
    keep = _lib._torch_ops
    try:
        for binding in (keep, None):
            _lib._torch_ops = binding
            out = ops.Act.empty(2, 20, 24, 128, dtype, dev)
            plan([xa], out, 20, 24, res=ra)
            outs.append(out.t.clone())
    finally:
        _lib._torch_ops = keep
    assert torch.equal(outs[0], outs[1])
    with pytest.raises(ops.OkpError):
        plan([xa], ops.Act.empty(2, 20, 24, 128, dtype, dev), 21, 24)          # output grid does not fit: reported by both bindings
