"""The stride-2 fire module 256 -> 128 -> 256 of the split-product configuration: one launch (okp_fire_x3_kernel<2>) against squeeze + fused tail,
times and the difference of the results.  usage (GPU box): python scripts/probe_fire_x3_s2.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
torch.manual_seed(0)
for (n, h, w) in ((64, 64, 64), (3, 38, 50), (2, 33, 47)):
    m = bb.fire_module(256, 256, stride=2).eval()
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point: v.copy_(torch.randn_like(v) * (0.05 if "weight" in k and v.dim() == 4 else 0.3) + (1.0 if "running_var" in k or ("bn" in k and "weight" in k) else 0.0))
    m.load_state_dict(m.state_dict()); 
    for mod in m.modules():
        if hasattr(mod, "running_var"): mod.running_var.abs_().add_(0.5)
    x = ops.Act(torch.randn(n, h, w, 256, device="cuda"))
    outs = {}
    for flag in (True, False):
        ops.FUSE_FIRE_X3_S2 = flag
        with ops.f32_split():
            l0 = ops.COUNTERS["launches"]; y = m(x); outs[flag] = y.t.clone(); nl = ops.COUNTERS["launches"] - l0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3): m(x)
            e0.record()
            for _ in range(10): m(x)
            e1.record(); torch.cuda.synchronize()
        print(f"n={n} {h}x{w} fused={flag}: launches {nl}, {e0.elapsed_time(e1)/10*1e3:.1f} us")
    d = (outs[True] - outs[False]).abs().max().item(); s = outs[False].abs().max().item()
    print(f"   max |fused - two-launch| = {d:.3e} (max |y| {s:.3f}), finite {bool(torch.isfinite(outs[True]).all())}")
