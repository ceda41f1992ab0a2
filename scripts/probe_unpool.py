"""Time the transposed convolution + merge (unpool_merge) at a hourglass level.  usage: probe_unpool.py [hw=32] [n=64] (hw = input size)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
kw = dict(hw=32, n=64, c=256, tile=13)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = int(v)
m = bb.unpool_merge(kw["c"]).eval()
low = ops.Act(torch.randn(kw["n"], kw["hw"], kw["hw"], kw["c"], device="cuda").bfloat16())
up1 = ops.Act(torch.randn(kw["n"], 2 * kw["hw"], 2 * kw["hw"], kw["c"], device="cuda").bfloat16())
for tile in (kw["tile"], kw["tile"]):
    bb.UNPOOL_TILE = tile
    for _ in range(3): y = m(low, up1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = m(low, up1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * kw["n"] * (2 * kw["hw"]) ** 2 * kw["c"] * kw["c"] * 4
    print(f"unpool {kw} tile {tile}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s")
