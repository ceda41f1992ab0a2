"""Time the fire module 256 -> 128 -> 256 (skip) of the split-product configuration: one launch (okp_fire_x3) against squeeze + fused tail.
usage: probe_fire_x3.py [hw=64] [n=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
kw = dict(hw=64, n=64)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = int(v)
m = bb.fire_module(256, 256).eval()
x = ops.Act(torch.randn(kw["n"], kw["hw"], kw["hw"], 256, device="cuda"))
mb = 2 * x.t.numel() * 4 / 1e6
for flag in (True, False, True, False):
    ops.FUSE_FIRE_X3 = flag
    with ops.f32_split():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m(x)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{'one launch (okp_fire_x3)' if flag else 'squeeze + fused tail     '}: {us:7.1f} us   {mb / us:5.2f} TB/s of x + out ({mb:.0f} MB)")
