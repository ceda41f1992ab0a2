import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from object_keypoints_amd import ops
n, hw, cout = 64, 4, 256
rng = np.random.default_rng(0)
big = torch.empty(768 * 1024 * 1024 // 4, device="cuda")
for cin in (128, 512):
    w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    with ops.f32_split():
        plans = [ops.ConvPlan(torch.float32, [cin], [1], cout, [(0, 0, 0, w)], np.zeros(cout, np.float32), relu=True) for _ in range(1)]
    plan = plans[0]
    x = ops.Act(torch.randn(n, hw, hw, cin, device="cuda"))
    out = ops.Act.empty(n, hw, hw, cout, torch.float32, x.t.device)
    for _ in range(5): plan([x], out, hw, hw)
    torch.cuda.synchronize()
    for mode in ("hot", "cold"):
        for _ in range(30):
            if mode == "cold":
                big.zero_()                      # 768 MB of stores: L2 and the Infinity Cache now hold something else
            plan([x], out, hw, hw)
        torch.cuda.synchronize()
print("done")
