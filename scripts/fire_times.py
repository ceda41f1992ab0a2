"""Isolated time of every one-launch fire module (okp_fire2) the network issues, by shape.
usage: fire_times.py [n=64]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import models

n = 64
for a in sys.argv[1:]:
    k, v = a.split("="); n = int(v) if k == "n" else n
net = models.KeypointNet(features=128, heatmaps_out=3, compute_dtype=torch.bfloat16).eval().cuda()
calls = []
orig = ops.fire_fused
def spy(squeeze, expand, wd, bd, x, out, stride, skip):
    calls.append((squeeze, expand, wd, bd, x, out, stride, skip))
    return orig(squeeze, expand, wd, bd, x, out, stride, skip)
ops.fire_fused = spy
x = torch.randn(n, 3, 511, 511, device="cuda")
with torch.no_grad():
    net(x)
ops.fire_fused = orig
torch.cuda.synchronize()
groups = collections.OrderedDict()
for c in calls:
    x_, out = c[4], c[5]
    key = (x_.c, out.c, c[6], tuple(x_.t.shape[1:3]), bool(c[7]))
    groups.setdefault(key, []).append(c)
tot = 0.0
print(f"{'cin':>5} {'cout':>5} s {'HxW':>9} skip   n   us/launch  total_us   min-GB   GB/s")
for key, cs in groups.items():
    c = cs[0]
    for _ in range(3): orig(*c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): orig(*c)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    xb = c[4].t.numel() // c[4].t.shape[-1] * c[4].c * 2
    ob = c[5].t.numel() // c[5].t.shape[-1] * c[5].c * 2
    gb = (xb + ob) / 1e9
    tot += us * len(cs)
    print(f"{key[0]:5d} {key[1]:5d} {key[2]} {key[3][0]:4d}x{key[3][1]:<4d} {int(key[4]):4d} {len(cs):3d} {us:10.1f} {us*len(cs):9.1f} {gb:8.3f} {gb/us*1e6:6.0f}")
print("sum", tot, "us")
