"""Split-product (OKP_F32X3) implicit-GEMM tiles on the HBM-bound 1x1 layers of the float32mix step at N=64: time per launch by tile code.
usage (GPU box): python scripts/x3_tile_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
N = 64
CASES = [  # name, cins, strides, cout, (h, w) of the output, residual?, out16?
    ("inters_/cnvs_ two-source 1x1", [256, 256], [1, 1], 256, 64, False, True),
    ("heads layer 1 256->384", [256], [1], 384, 64, False, False),
    ("fire squeeze 256->128", [256], [1], 128, 64, False, False),
    ("pre[2] skip 256->256 (compact source) + fp16 res", [256], [1], 256, 64, True, True),
    ("pre[1] skip 128->256 (compact source) + fp16 res", [128], [1], 256, 128, True, True),
]
for name, cins, strides, cout, hw, res, o16 in CASES:
    taps = [(s, 0, 0, (rng.standard_normal((cout, c)) / np.sqrt(c * len(cins))).astype(np.float32)) for s, c in enumerate(cins)]
    with ops.f32_split():
        plan = ops.ConvPlan(torch.float32, cins, strides, cout, taps, np.zeros(cout, np.float32), relu=True)
    srcs = [ops.Act(torch.randn((N, hw * st, hw * st, c), device=dev)) for c, st in zip(cins, strides)]
    out = ops.Act.empty(N, hw, hw, cout, torch.float32, dev)
    r16 = ops.Act(torch.randn((N, hw, hw, cout), device=dev).half()) if res else None
    s16 = ops.Act.empty(N, hw, hw, cout, torch.float16, dev) if o16 else None
    row = []
    for tile in (2, 3, 4):
        for _ in range(3): plan(srcs, out, hw, hw, res=r16, out16=s16, tile=tile)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): plan(srcs, out, hw, hw, res=r16, out16=s16, tile=tile)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 20 * 1e6)
    print(f"{name:40s} tile2 {row[0]:7.1f} us  tile3 {row[1]:7.1f} us  tile4 {row[2]:7.1f} us")

# the same question for the 16-bit tiles (bf16): short-K 1x1 launches of the hourglass / inter-stack merge
for name, cins, cout, hw in [("level-1 low1[0] squeeze 256->192 @32x32", [256], 192, 32), ("inters_/cnvs_ two-source 1x1 @64x64", [256, 256], 256, 64),
                             ("level-2 squeeze 384->192 @16x16", [384], 192, 16)]:
    taps = [(s, 0, 0, (rng.standard_normal((cout, c)) / np.sqrt(c * len(cins))).astype(np.float32)) for s, c in enumerate(cins)]
    plan = ops.ConvPlan(torch.bfloat16, cins, [1] * len(cins), cout, taps, np.zeros(cout, np.float32), relu=True)
    srcs = [ops.Act(torch.randn((N, hw, hw, c), device=dev).bfloat16()) for c in cins]
    out = ops.Act.empty(N, hw, hw, cout, torch.bfloat16, dev)
    row = {}
    for tile in (8, 2, 4, 6):
        for _ in range(3): plan(srcs, out, hw, hw, tile=tile)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): plan(srcs, out, hw, hw, tile=tile)
        torch.cuda.synchronize()
        row[tile] = (time.perf_counter() - t0) / 50 * 1e6
    print(f"bf16 {name:42s} " + "  ".join(f"tile{t} {v:6.1f} us" for t, v in row.items()) + f"   heuristic: tile {plan.select_tile(srcs, out, hw, hw) if hasattr(plan, 'select_tile') else '?'}")
