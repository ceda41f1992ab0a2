"""Board power and shader clock per KERNEL: each of the step's big kernels alone, launched back to back for ~2.5 s next to rocm-smi
(bench.power_probe).  usage: probe_kernel_power.py   (prints one line per kernel: us per launch, W, MHz)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception import backbone as bb
from object_keypoints_amd.perception.backbone import conv_taps
dev = torch.device("cuda")
rng = np.random.default_rng(0)
n = 64


def conv_case(hw, cin, stride, scale, dtype=torch.bfloat16, tile=0):
    wt = (rng.standard_normal((256, cin, 3, 3)) / np.sqrt(cin * 9) * scale).astype(np.float32)
    plan = ops.ConvPlan(dtype, [cin], [stride], 256, conv_taps(wt), np.zeros(256, np.float32), relu=True)
    x = ops.Act((torch.randn(n, hw * stride, hw * stride, cin, device=dev) * scale).to(torch.bfloat16 if dtype == torch.bfloat16 else torch.float32 if dtype == ops.F32X3 else dtype))
    out = ops.Act.empty(n if dtype != ops.F32X3 else n, hw, hw, 256, x.t.dtype, dev)
    return lambda: plan([x], out, hw, hw, tile=tile)


def fire_case(hw):
    m = bb.fire_module(256, 256).eval()
    x = ops.Act(torch.randn(n, hw, hw, 256, device=dev).bfloat16())
    return lambda: m(x)


def stem_case():
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=torch.bfloat16)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
    net.eval().cuda()
    frames = torch.from_numpy(synth.frames(n, seed=1)).cuda()
    pre0 = net.backbone.pre[0]
    return lambda: pre0(frames, torch.bfloat16) if False else net.backbone.stem(frames, torch.bfloat16)


# usage: probe_kernel_power.py [shapes]   (shapes: 16x16x32 against 32x32x16 instantiation of the patch kernel, tile 13 / 14, J per launch)
if len(sys.argv) > 1 and sys.argv[1] == "shapes":
    cases = []
    for hw, cin, st in ((64, 256, 1), (128, 256, 1), (64, 256, 2)):
        for tile, nm in ((13, "16x16x32"), (14, "32x32x16"), (13, "16x16x32"), (14, "32x32x16")):
            cases.append((f"patch kernel {nm}, 3x3 {cin}->256 at {hw}x{hw} stride {st}, random operands", conv_case(hw, cin, st, 1.0, tile=tile)))
    cases.append(("patch kernel 16x16x32, 3x3 256->256 at 64x64, all-zero operands", conv_case(64, 256, 1, 0.0, tile=13)))
    cases.append(("patch kernel 32x32x16, 3x3 256->256 at 64x64, all-zero operands", conv_case(64, 256, 1, 0.0, tile=14)))
else:
  cases = [("patch kernel, 3x3 256->256 at 64x64, random operands", conv_case(64, 256, 1, 1.0)),
           ("patch kernel, 3x3 256->256 at 64x64, all-zero operands", conv_case(64, 256, 1, 0.0)),
           ("patch kernel, 3x3 256->256 at 128x128, random operands", conv_case(128, 256, 1, 1.0)),
           ("patch kernel, stride-2 3x3 128->256 at 128x128", conv_case(128, 128, 2, 1.0)),
           ("okp_fire2 256->128->256 at 64x64", fire_case(64)),
           ("okp_fire2 256->128->256 at 32x32", fire_case(32))]
with torch.no_grad():
    for name, fn in cases:
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r = bench.power_probe(fn, seconds=2.5)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{name:78s} {us:8.1f} us per launch   {r['board_W_mean']:7.0f} W   {r['sclk_MHz_mean']:6.0f} MHz   {r['board_W_mean'] * us * 1e-6:6.3f} J per launch   ({r['samples']} samples)" if r else f"{name}: no rocm-smi")
