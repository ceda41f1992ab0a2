import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
m = bb.convolution(7, 3, 128, stride=2).eval()
for name, dt, n in (("bf16", torch.bfloat16, 64), ("x3", torch.float32, 32)):
    x = torch.randn(n, 3, 511, 511, device="cuda")
    ctx = ops.f32_split() if name == "x3" else torch.no_grad()
    with ctx:
        for _ in range(5): out = m.forward_frames(x, dt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): out = m.forward_frames(x, dt)
        e1.record(); torch.cuda.synchronize()
    print(f"stem {name} n={n}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", end="  ")
print()
