"""Time one fire module (256 -> 256, stride 1) at a hourglass level: unfused (2 launches) vs one-launch kernels.
usage: probe_fire.py [hw=64] [n=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
kw = dict(hw=64, n=64, cin=256)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = int(v)
m = bb.fire_module(kw["cin"], kw["cin"]).eval()
x = ops.Act(torch.randn(kw["n"], kw["hw"], kw["hw"], kw["cin"], device="cuda").bfloat16())
for fuse in (False, True, True, False):
    ops.FUSE_FIRE = fuse; ops.FUSE_FIRE_MIN_HW = 0
    for _ in range(3): y = m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = m(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = 2 * x.t.numel() * 2 / 1e9
    print(f"fire {kw} fused={fuse}: {us:.1f} us  ({gb / us * 1e6:.0f} GB/s of x+out)")
