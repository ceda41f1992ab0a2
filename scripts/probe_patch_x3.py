"""Time one 3x3 convolution -> 256 channels of a split-product (OKP_F32X3) plan on the gather tile (3) and the patch-resident kernel (13).
usage: probe_patch_x3.py [hw=64] [n=64] [res=0] [cin=256] [stride=1] [skip=0] [zero=0] [tiles=3,13]
(hw = output size; skip = channels of a fused strided 1x1 second source; zero=1: all-zero operands, the rate without data-dependent power)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception.backbone import conv_taps
kw = dict(hw=64, n=64, res=0, cin=256, stride=1, skip=0, zero=0, tiles="3,13")
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = v if k == "tiles" else int(v)
n, hw, cin, st, skip = kw["n"], kw["hw"], kw["cin"], kw["stride"], kw["skip"]
rng = np.random.default_rng(0)
wt = (rng.standard_normal((256, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
if kw["zero"]: wt[:] = 0
taps, cins, strides = conv_taps(wt), [cin], [st]
z = 0.0 if kw["zero"] else 1.0
srcs = [ops.Act(torch.randn(n, hw * st, hw * st, cin, device="cuda") * z)]
if skip:
    ws = (rng.standard_normal((256, skip)) / np.sqrt(skip)).astype(np.float32) * np.float32(z)
    taps = taps + [(1, 0, 0, ws)]; cins.append(skip); strides.append(2)
    srcs.append(ops.Act(torch.randn(n, hw * 2, hw * 2, skip, device="cuda") * z))
with ops.f32_split():
    plan = ops.ConvPlan(torch.float32, cins, strides, 256, taps, np.zeros(256, np.float32), relu=True)
r = ops.Act(torch.randn(n, hw, hw, 256, device="cuda")) if kw["res"] else None
out = ops.Act.empty(n, hw, hw, 256, torch.float32, srcs[0].t.device)
tiles = [int(v) for v in kw["tiles"].split(",")]
for _ in range(30): plan(srcs, out, hw, hw, res=r, tile=tiles[0])       # warm clocks
ref = None
for tile in tiles:
    for _ in range(3): plan(srcs, out, hw, hw, res=r, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan(srcs, out, hw, hw, res=r, tile=tile)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * n * hw * hw * 256 * (cin * 9 + skip)
    d = "" if ref is None else f"   max |diff| to the first tile {float((out.t - ref).abs().max()):.2e}"
    if ref is None: ref = out.t.clone()
    print(f"tile {tile:2d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s algorithmic  {3 * fl / us / 1e6:7.1f} issued{d}")
