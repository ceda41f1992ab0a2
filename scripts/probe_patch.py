"""Time one 3x3 convolution 256 -> 256 (bf16) with the gather tile (6) and the patch-resident kernel (13).
usage: probe_patch.py [hw=64] [n=64] [res=0] [cin=256] [stride=1] [zero=0] [tiles=13,13]   (hw = output size; zero=1: all-zero operands)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception.backbone import conv_taps
kw = dict(hw=64, n=64, res=0, cin=256, stride=1, zero=0, tiles="13,13")
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = v if k == "tiles" else int(v)
n, hw, cin = kw["n"], kw["hw"], kw["cin"]
rng = np.random.default_rng(0)
wt = (rng.standard_normal((256, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
st = kw["stride"]
if kw["zero"]: wt[:] = 0
plan = ops.ConvPlan(torch.bfloat16, [cin], [st], 256, conv_taps(wt), np.zeros(256, np.float32), relu=True)
x = ops.Act(torch.randn(n, hw * st, hw * st, cin, device="cuda").bfloat16() * (0 if kw["zero"] else 1))
r = ops.Act(torch.randn(n, hw, hw, 256, device="cuda").bfloat16()) if kw["res"] else None
out = ops.Act.empty(n, hw, hw, 256, torch.bfloat16, x.t.device)
for _ in range(60): plan([x], out, hw, hw, res=r, tile=int(kw['tiles'].split(',')[0]))       # warm clocks
for tile in [int(v) for v in kw['tiles'].split(',')]:
    for _ in range(3): plan([x], out, hw, hw, res=r, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan([x], out, hw, hw, res=r, tile=tile)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * n * hw * hw * 256 * cin * 9
    print(f"tile {tile:2d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
