"""The patch-resident split-product kernel (okp_igemm_patch_x3.hip: tile 13 of OKP_F32X3 plans) - fp32 tensors, 3x3 / 4x4-transposed
convolutions with 256 output channels, the input patch split once into fp16 hi | lo in LDS.  Checked against INDEPENDENT references
(torch's fp64 / fp32 CPU convolutions of the same fp32 operands) with the split-product tolerance of tests/test_gpu_f32x3.py, against
the gather tile of the same plan (tile 3: same products, other summation order), at small shapes that cover every geometry the
kernel has (stride 1, stride 2 = four parity patches, fused strided 1x1 skip = one-step groups, sub-pixel classes, image borders,
several frames / channel tiles) and at the bench's batch of 64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed):
    from object_keypoints_amd import synth
    return torch.from_numpy(synth.normal_like(f"px3{seed}", shape, seed))


def _tol(ref):
    return 2e-5 * (1.0 + float(ref.abs().max()))       # far inside one fp16 rounding of an operand (2^-11): the low halves are in use


class _TileSpy:
    def __init__(self):
        self.tiles = []

    def before(self, plan, tile, macs):
        self.tiles.append(tile)

    def after(self, token):
        pass


def _act(x):
    from object_keypoints_amd import ops
    return ops.Act.from_nchw(x.cuda(), torch.float32)


@pytest.mark.parametrize("cin,cout,n,h,w,stride,res", [
    (32, 256, 1, 16, 16, 1, False),     # one chunk, one tile: every patch pixel outside the block is padding
    (64, 256, 2, 32, 48, 1, True),      # two chunks (the second one split during the first one's last step), interior halos, residual
    (96, 512, 3, 16, 32, 1, False),     # two channel tiles, three chunks
    (32, 256, 2, 16, 32, 2, False),     # stride 2: four parity patches per chunk (groups of 4, 2, 2 and 1 steps)
    (128, 256, 1, 32, 16, 2, True),     # stride 2, four chunks
    (256, 256, 2, 16, 16, 1, True),     # the network's 3x3 shape, small map
])
def test_patch_x3_conv_matches_cpu_fp64(cin, cout, n, h, w, stride, res):
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    x = _rand((n, cin, h * stride, w * stride), 1)
    wt = _rand((cout, cin, 3, 3), 2) / np.sqrt(cin * 9)
    b = _rand((cout,), 3) * 0.1
    r = _rand((n, cout, h, w), 4) if res else None
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=1)
    ref = F.relu(ref + r.double() if res else ref).float()
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, [cin], [stride], cout, conv_taps(wt.numpy()), b.numpy(), relu=True)
    xa, ra = _act(x), (_act(r) if res else None)
    out = ops.Act.empty(n, h, w, cout, torch.float32, xa.t.device)
    plan([xa], out, h, w, res=ra, tile=13)
    got = out.to_nchw().cpu()
    assert float((got - ref).abs().max()) <= _tol(ref)
    out3 = ops.Act.empty(n, h, w, cout, torch.float32, xa.t.device)
    plan([xa], out3, h, w, res=ra, tile=3)                       # the gather tile: same products, other order of the sum
    assert float((out3.t - out.t).abs().max()) <= 1e-5 * (1.0 + float(ref.abs().max()))
    again = ops.Act.empty(n, h, w, cout, torch.float32, xa.t.device)
    plan([xa], again, h, w, res=ra, tile=13)
    assert torch.equal(again.t, out.t)                           # run-to-run bit-reproducible


def test_patch_x3_two_sources_with_a_strided_skip():
    """conv2 + projected 1x1/s2 skip + ReLU of `residual` as one launch: the second source's patches are one-step groups (split at the
    start of their own step, behind a second barrier), the first chunk of the 3x3 source follows such a group."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    n, c0, c1, cout, h, w = 2, 64, 96, 256, 32, 16
    t = _rand((n, c0, h, w), 5); x = _rand((n, c1, 2 * h, 2 * w), 6)
    w2 = _rand((cout, c0, 3, 3), 7) / np.sqrt(c0 * 9); ws = _rand((cout, c1, 1, 1), 8) / np.sqrt(c1)
    b = _rand((cout,), 9) * 0.1
    ref = F.relu(F.conv2d(t.double(), w2.double(), b.double(), padding=1) + F.conv2d(x.double(), ws.double(), stride=2)).float()
    taps = conv_taps(w2.numpy()) + [(1, 0, 0, np.ascontiguousarray(ws.numpy()[:, :, 0, 0]))]
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, [c0, c1], [1, 2], cout, taps, b.numpy(), relu=True)
    ta, xa = _act(t), _act(x)
    out = ops.Act.empty(n, h, w, cout, torch.float32, ta.t.device)
    ops.LAUNCH_HOOK = spy = _TileSpy()
    try:
        plan([ta, xa], out, h, w)                                  # the heuristic's own choice
    finally:
        ops.LAUNCH_HOOK = None
    assert float((out.to_nchw().cpu() - ref).abs().max()) <= _tol(ref)
    out13 = ops.Act.empty(n, h, w, cout, torch.float32, ta.t.device)
    plan([ta, xa], out13, h, w, tile=13)
    assert float((out13.to_nchw().cpu() - ref).abs().max()) <= _tol(ref)
    assert spy.tiles in ([13], [3], [2], [4], [1])                 # (small problem: the heuristic may keep a gather tile)


def test_patch_x3_transposed_convolution_classes():
    """The 4x4/s2 transposed convolution + merge add: four sub-pixel classes, each a K range of its own whose first patch comes from
    the tile's prologue."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import unpool_merge
    n, c, h, w = 2, 256, 16, 32
    m = unpool_merge(c).eval()
    with torch.no_grad():
        m.weight.copy_(_rand((c, c, 4, 4), 12) / np.sqrt(4 * c)); m.bias.copy_(_rand((c,), 13) * 0.1)
    low = _rand((n, c, h, w), 14); up1 = _rand((n, c, 2 * h, 2 * w), 15)
    ref = (up1.double() + F.conv_transpose2d(low.double(), m.weight.detach().double(), m.bias.detach().double(), stride=2, padding=1)).float()
    from object_keypoints_amd.perception import backbone
    prev = backbone.UNPOOL_TILE
    backbone.UNPOOL_TILE = 13
    try:
        with ops.f32_split():
            got = m(_act(low), _act(up1)).to_nchw().cpu()
    finally:
        backbone.UNPOOL_TILE = prev
    assert float((got - ref).abs().max()) <= _tol(ref)


def test_patch_x3_small_operands_keep_their_low_halves():
    """Activations around 1e-2 (low halves in the fp16 subnormal range) and weights around 1e-4 (scaled per channel at plan creation):
    the relative accuracy of the in-LDS split is that of the in-register split."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    n, cin, cout, h, w = 1, 64, 256, 16, 16
    x = _rand((n, cin, h, w), 21) * 1e-2
    wt = _rand((cout, cin, 3, 3), 22) * (1e-4 / np.sqrt(cin * 9))
    ref = F.conv2d(x.double(), wt.double(), padding=1).float()
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, [cin], [1], cout, conv_taps(wt.numpy()), None, relu=False)
    out = ops.Act.empty(n, h, w, cout, torch.float32, torch.device("cuda:0"))
    plan([_act(x)], out, h, w, tile=13)
    err = float((out.to_nchw().cpu() - ref).abs().max())
    assert err <= 3e-6 * float(ref.abs().max()) + 2e-11, err


def test_patch_x3_is_what_the_heuristic_launches_at_the_network_shapes():
    """3x3 256 -> 256 at 64 x 64, N = 16 (256 tiles of 256 pixels: one per CU): tile 13 without being asked; the fp16 side output of the
    mixed configuration keeps the gather tile."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    dev = torch.device("cuda:0")
    wt = _rand((256, 256, 3, 3), 31) / np.sqrt(256 * 9)
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, [256], [1], 256, conv_taps(wt.numpy()), None, relu=True)
    x = ops.Act(torch.randn(16, 64, 64, 256, device=dev))
    out = ops.Act.empty(16, 64, 64, 256, torch.float32, dev)
    ops.LAUNCH_HOOK = spy = _TileSpy()
    try:
        plan([x], out, 64, 64)
        plan([x], out, 64, 64, out16=ops.Act.empty(16, 64, 64, 256, torch.float16, dev))
    finally:
        ops.LAUNCH_HOOK = None
    assert spy.tiles == [13, 3]
    with pytest.raises(ops.OkpError):
        plan([x], out, 64, 64, out16=ops.Act.empty(16, 64, 64, 256, torch.float16, dev), tile=13)


@pytest.mark.parametrize("case", ["conv3x3_64", "conv3x3_s2_128", "residual_s2_skip_64", "unpool_32_to_64"])
def test_patch_x3_at_batch64_against_cpu_convolution(case):
    """The kernel at the bench's batch of 64 (XCD-aware persistent grid, 16 384 tiles) against torch's fp32 CPU convolution on frames
    0, 31 and 63."""
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception import backbone
    from object_keypoints_amd.perception.backbone import conv_taps, unpool_merge
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(11)
    rnd = lambda *shape: torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
    rw = lambda name, shape, fan: synth.normal_like(name, shape, 60) / np.float32(np.sqrt(fan))
    n, frames = 64, (0, 31, 63)
    nchw = lambda t, i: t[i:i + 1].permute(0, 3, 1, 2).cpu().double()
    tol = lambda ref: 2e-5 * (1.0 + float(ref.abs().max()))
    if case == "unpool_32_to_64":
        m = unpool_merge(256).eval()
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(rw("up2", (256, 256, 4, 4), 4 * 256))); m.bias.copy_(torch.from_numpy(synth.normal_like("up2b", (256,), 61) * np.float32(0.1)))
        low, up1 = rnd(n, 32, 32, 256), rnd(n, 64, 64, 256)
        ops.LAUNCH_HOOK = hook = _TileSpy()
        try:
            with ops.f32_split():
                got = m(ops.Act(low), ops.Act(up1)).t
        finally:
            ops.LAUNCH_HOOK = None
        assert hook.tiles == [13]
        for i in frames:
            ref = nchw(up1, i) + F.conv_transpose2d(nchw(low, i), m.weight.detach().double(), m.bias.detach().double(), stride=2, padding=1)
            assert float((nchw(got, i) - ref).abs().max()) <= tol(ref), i
        return
    d = lambda a: torch.from_numpy(a).double()
    if case == "conv3x3_64":
        h = w = 64
        wt, b = rw("w64", (256, 256, 3, 3), 256 * 9), synth.normal_like("b64", (256,), 62) * np.float32(0.1)
        with ops.f32_split():
            plan = ops.ConvPlan(torch.float32, [256], [1], 256, conv_taps(wt), b, relu=True)
        srcs, res = [rnd(n, h, w, 256)], rnd(n, h, w, 256)
        ref_fn = lambda i: F.relu(F.conv2d(nchw(srcs[0], i), d(wt), d(b), padding=1) + nchw(res, i))
    elif case == "conv3x3_s2_128":
        h = w = 64
        wt, b = rw("ws2", (256, 128, 3, 3), 128 * 9), synth.normal_like("bs2", (256,), 63) * np.float32(0.1)
        with ops.f32_split():
            plan = ops.ConvPlan(torch.float32, [128], [2], 256, conv_taps(wt), b, relu=True)
        srcs, res = [rnd(n, 2 * h, 2 * w, 128)], None
        ref_fn = lambda i: F.relu(F.conv2d(nchw(srcs[0], i), d(wt), d(b), stride=2, padding=1))
    else:
        h = w = 64
        w2, ws = rw("w2", (256, 256, 3, 3), 256 * 9), rw("wskip", (256, 256, 1, 1), 256)
        b = synth.normal_like("b2", (256,), 64) * np.float32(0.1)
        taps = conv_taps(w2) + [(1, 0, 0, np.ascontiguousarray(ws[:, :, 0, 0]))]
        with ops.f32_split():
            plan = ops.ConvPlan(torch.float32, [256, 256], [1, 2], 256, taps, b, relu=True)
        srcs, res = [rnd(n, h, w, 256), rnd(n, 2 * h, 2 * w, 256)], None
        ref_fn = lambda i: F.relu(F.conv2d(nchw(srcs[0], i), d(w2), d(b), padding=1) + F.conv2d(nchw(srcs[1], i), d(ws), stride=2))
    out = ops.Act.empty(n, h, w, 256, torch.float32, dev)
    ops.LAUNCH_HOOK = hook = _TileSpy()
    try:
        plan([ops.Act(s) for s in srcs], out, h, w, res=ops.Act(res) if res is not None else None)
    finally:
        ops.LAUNCH_HOOK = None
    assert hook.tiles == [13]
    for i in frames:
        ref = ref_fn(i)
        assert float((nchw(out.t, i) - ref).abs().max()) <= tol(ref), i


@pytest.mark.parametrize("n,h,w", [(1, 16, 16), (2, 20, 37), (3, 64, 64), (64, 32, 32), (64, 64, 64)])
def test_split_fire_module_in_one_launch(n, h, w):
    """okp_fire_x3_kernel (fire_module 256 -> 128 -> 256 with skip in split-product mode: x ring split once in LDS, squeeze tile resident)
    against the oracle's fire_module (torch CPU fp32) and against the two-launch path it replaces, partial tiles and the bench's batch
    included (frames 0 / middle / last), bit-reproducible."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    o = onet.load_synthetic(onet.fire_module(256, 256), seed=21)
    m = bb.fire_module(256, 256)
    m.load_state_dict(o.state_dict())
    m.eval()
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = torch.randn((n, h, w, 256), generator=gen, device="cuda")
    with ops.f32_split():
        l0 = ops.COUNTERS["launches"]
        got = m(ops.Act(x))
        assert ops.COUNTERS["launches"] - l0 == 1
        again = m(ops.Act(x))
        ops.FUSE_FIRE_X3 = False
        try:
            two = m(ops.Act(x))
            assert ops.COUNTERS["launches"] - l0 == 4
        finally:
            ops.FUSE_FIRE_X3 = True
    assert torch.equal(again.t, got.t)
    sample = sorted({0, n // 2, n - 1})
    with torch.no_grad():
        ref = o(x[sample].permute(0, 3, 1, 2).cpu().double() if False else x[sample].permute(0, 3, 1, 2).cpu())
    g = got.t[sample].permute(0, 3, 1, 2).cpu()
    scale = 1.0 + float(ref.abs().max())
    assert float((g - ref).abs().max()) <= 2e-5 * scale
    assert float((got.t - two.t).abs().max()) <= 1e-5 * scale


def test_patch_x3_random_shapes_against_the_gather_tile():
    """Seeded sweep over the shapes the heuristic can hand to the split-product patch kernel: input channels 32..160 (one to five chunks,
    ragged against the 64-channel pairs), one or two channel tiles, 1..5 frames, maps of 16..80 pixels per side in steps of 16 (so that
    tiles sit on every image border), stride 1 / 2, with and without residual, with and without a strided 1x1 second source.  The
    gather tile of the same plan is the checker (the same products in another order), a CPU fp64 convolution the anchor for a sample."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    rng = np.random.default_rng(20251004)
    dev = torch.device("cuda:0")
    for case in range(24):
        cin = int(rng.choice([32, 64, 96, 128, 160]))
        cout = int(rng.choice([256, 256, 512]))
        n = int(rng.integers(1, 6))
        h, w = (int(v) for v in rng.choice([16, 32, 48, 64, 80], size=2))
        stride = int(rng.choice([1, 1, 2]))
        res = bool(rng.integers(0, 2))
        skip = int(rng.choice([0, 0, 32, 64])) if stride == 1 else 0
        g = torch.Generator(device=dev); g.manual_seed(1000 + case)
        x = torch.randn((n, h * stride, w * stride, cin), generator=g, device=dev)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        taps, cins, strides, srcs = conv_taps(wt), [cin], [stride], [ops.Act(x)]
        if skip:
            ws = (rng.standard_normal((cout, skip)) / np.sqrt(skip)).astype(np.float32)
            taps = taps + [(1, 0, 0, ws)]; cins.append(skip); strides.append(2)
            srcs.append(ops.Act(torch.randn((n, 2 * h, 2 * w, skip), generator=g, device=dev)))
        r = ops.Act(torch.randn((n, h, w, cout), generator=g, device=dev)) if res else None
        with ops.f32_split():
            plan = ops.ConvPlan(torch.float32, cins, strides, cout, taps, b, relu=True)
        o13 = ops.Act.empty(n, h, w, cout, torch.float32, dev)
        o3 = ops.Act.empty(n, h, w, cout, torch.float32, dev)
        plan(srcs, o13, h, w, res=r, tile=13)
        plan(srcs, o3, h, w, res=r, tile=3)
        scale = 1.0 + float(o3.t.abs().max())
        assert float((o13.t - o3.t).abs().max()) <= 1e-5 * scale, (case, cin, cout, n, h, w, stride, res, skip)
        if case % 6 == 0:
            ref = F.conv2d(x[:1].permute(0, 3, 1, 2).cpu().double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), stride=stride, padding=1)
            if skip:
                ref = ref + F.conv2d(srcs[1].t[:1].permute(0, 3, 1, 2).cpu().double(), torch.from_numpy(ws).double()[:, :, None, None], stride=2)
            if res:
                ref = ref + r.t[:1].permute(0, 3, 1, 2).cpu().double()
            assert float((o13.t[:1].permute(0, 3, 1, 2).cpu().double() - F.relu(ref)).abs().max()) <= 2e-5 * scale, case


def test_fire_x3_random_map_sizes_against_the_two_launch_path():
    """Seeded sweep of okp_fire_x3 over map sizes 16..70 (partial tiles in both directions, maps narrower than a tile row) and 1..4 frames
    against the squeeze + fused-tail launches of the same module."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    o = onet.load_synthetic(onet.fire_module(256, 256), seed=33)
    m = bb.fire_module(256, 256)
    m.load_state_dict(o.state_dict())
    m.eval()
    rng = np.random.default_rng(7)
    dev = torch.device("cuda:0")
    for case in range(12):
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(16, 71)), int(rng.integers(16, 71))
        g = torch.Generator(device=dev); g.manual_seed(case)
        x = ops.Act(torch.randn((n, h, w, 256), generator=g, device=dev))
        with ops.f32_split():
            got = m(x)
            ops.FUSE_FIRE_X3 = False
            try:
                two = m(x)
            finally:
                ops.FUSE_FIRE_X3 = True
        assert float((got.t - two.t).abs().max()) <= 1e-5 * (1.0 + float(two.t.abs().max())), (case, n, h, w)


@pytest.mark.parametrize("n,h,w", [(1, 32, 32), (2, 38, 50), (3, 33, 47), (64, 64, 64)])
def test_split_stride2_fire_module_in_one_launch(n, h, w):
    """okp_fire_x3_kernel<2> (fire_module 256 -> 128 -> 256 at stride 2, no skip: the 64 x 64 -> 32 x 32 module of the hourglass) against the
    oracle's fire_module and against the squeeze + fused-tail launches it replaces: odd input sizes (the last input row / column is a
    padding neighbour only), partial tiles, the bench's batch; bit-reproducible."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    o = onet.load_synthetic(onet.fire_module(256, 256, stride=2), seed=41)
    m = bb.fire_module(256, 256, stride=2)
    m.load_state_dict(o.state_dict())
    m.eval()
    gen = torch.Generator(device="cuda"); gen.manual_seed(8)
    x = torch.randn((n, h, w, 256), generator=gen, device="cuda")
    with ops.f32_split():
        l0 = ops.COUNTERS["launches"]
        got = m(ops.Act(x))
        assert ops.COUNTERS["launches"] - l0 == 1
        again = m(ops.Act(x))
        ops.FUSE_FIRE_X3_S2 = False
        try:
            two = m(ops.Act(x))
            assert ops.COUNTERS["launches"] - l0 == 4
        finally:
            ops.FUSE_FIRE_X3_S2 = True
    assert tuple(got.t.shape) == (n, (h + 1) // 2, (w + 1) // 2, 256)
    assert torch.equal(again.t, got.t)
    sample = sorted({0, n // 2, n - 1})
    with torch.no_grad():
        ref = o(x[sample].permute(0, 3, 1, 2).cpu())
    g = got.t[sample].permute(0, 3, 1, 2).cpu()
    scale = 1.0 + float(ref.abs().max())
    assert float((g - ref).abs().max()) <= 2e-5 * scale
    assert float((got.t - two.t).abs().max()) <= 1e-5 * scale


def test_stride2_fire_x3_random_map_sizes_against_the_two_launch_path():
    """Seeded sweep of the stride-2 okp_fire_x3 over input maps of 32..90 pixels per side and 1..4 frames against the two launches of the module."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    o = onet.load_synthetic(onet.fire_module(256, 256, stride=2), seed=43)
    m = bb.fire_module(256, 256, stride=2)
    m.load_state_dict(o.state_dict())
    m.eval()
    rng = np.random.default_rng(9)
    dev = torch.device("cuda:0")
    for case in range(10):
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(32, 91)), int(rng.integers(32, 91))
        g = torch.Generator(device=dev); g.manual_seed(100 + case)
        x = ops.Act(torch.randn((n, h, w, 256), generator=g, device=dev))
        with ops.f32_split():
            l0 = ops.COUNTERS["launches"]
            got = m(x)
            assert ops.COUNTERS["launches"] - l0 == 1, (case, n, h, w)
            ops.FUSE_FIRE_X3_S2 = False
            try:
                two = m(x)
            finally:
                ops.FUSE_FIRE_X3_S2 = True
        assert float((got.t - two.t).abs().max()) <= 1e-5 * (1.0 + float(two.t.abs().max())), (case, n, h, w)
