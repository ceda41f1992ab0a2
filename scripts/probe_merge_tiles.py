"""The 1x1 layers of the 16-bit step that run on the gather tiles, on every tile that applies: the inter-stack merge (two sources, K = 512,
py_utils/modules.py:89-91) and a one-source 1x1 256 -> 256 at 64 x 64, N = 64.  HIP events around 20 launches each.
usage: probe_merge_tiles.py [bf16|f16] [n=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
w = lambda co, ci: (torch.randn((co, ci), generator=g) / np.sqrt(ci)).numpy()
cases = {
    "merge 2 x (256 -> 256), K=512": (ops.ConvPlan(dtype, [256, 256], [1, 1], 256, [(0, 0, 0, w(256, 256)), (1, 0, 0, w(256, 256))], np.zeros(256, np.float32), relu=True), 2),
    "1x1 256 -> 256": (ops.ConvPlan(dtype, [256], [1], 256, [(0, 0, 0, w(256, 256))], np.zeros(256, np.float32), relu=True), 1),
    "1x1 256 -> 384 (heads layer 1 shape)": (ops.ConvPlan(dtype, [256], [1], 384, [(0, 0, 0, w(384, 256))], np.zeros(384, np.float32), relu=True), 1),
}
for label, (plan, ns) in cases.items():
    srcs = [ops.Act(torch.randn((n, 64, 64, 256), device=dev).to(dtype)) for _ in range(ns)]
    out = ops.Act.empty(n, 64, 64, plan.cout, dtype, dev)
    mb = (sum(s.t.numel() for s in srcs) + out.t.numel()) * 2 / 1e6
    for tile in (0, 6, 3, 2, 8, 1):
        try:
            for _ in range(3):
                plan(srcs, out, 64, 64, tile=tile)
        except ops.OkpError as e:
            print(f"{label:40s} tile {tile}: {e}")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan(srcs, out, 64, 64, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{label:40s} tile {tile:2d}: {us:7.1f} us  {mb / us:5.2f} TB/s of {mb:.0f} MB  {2 * n * 4096 * plan.cout * plan.alg_k / us / 1e6:6.0f} TFLOP/s", flush=True)
