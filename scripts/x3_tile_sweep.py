"""Split-product (OKP_F32X3) 1x1 layers of the network at N = 64 on every gather tile: which tile the HBM-bound launches want.
usage: x3_tile_sweep.py [n=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
n = int(sys.argv[1].split("=")[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
dev = torch.device("cuda")
def run(name, cins, cout, hw, res=False, strides=None):
    taps = [(i, 0, 0, (rng.standard_normal((cout, c)) / np.sqrt(sum(cins))).astype(np.float32)) for i, c in enumerate(cins)]
    strides = strides or [1] * len(cins)
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, cins, strides, cout, taps, np.zeros(cout, np.float32), relu=True)
    srcs = [ops.Act(torch.randn(n, hw * s, hw * s, c, device=dev)) for c, s in zip(cins, strides)]
    out = ops.Act.empty(n, hw, hw, cout, torch.float32, dev)
    r = ops.Act(torch.randn(n, hw, hw, cout, device=dev)) if res else None
    mb = (sum(s.t.numel() for s in srcs) + out.t.numel() * (2 if res else 1)) * 4 / 1e6
    line = f"{name:34s} {mb:7.0f} MB:"
    for tile in (0, 1, 2, 3, 4):
        for _ in range(3): plan(srcs, out, hw, hw, res=r, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan(srcs, out, hw, hw, res=r, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        line += f"  t{tile} {us:6.1f} us ({mb / us:5.2f} TB/s)"
    print(line)
run("merge 256+256 -> 256 @64", [256, 256], 256, 64)
run("heads l1 256 -> 384 @64", [256], 384, 64)
run("heads l2 384 -> 96 @64", [384], 96, 64)
run("squeeze 256 -> 128 @64", [256], 128, 64)
run("squeeze 256 -> 128 @32", [256], 128, 32)
run("skip 256 -> 256 @64 +res", [256], 256, 64, res=True)
run("squeeze 384 -> 192 @16", [384], 192, 16)
