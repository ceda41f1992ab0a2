"""okp_fire2 reproducibility at the bench shapes: R launches on the same input, compared with the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
res = []
for (c, hw, stride, n) in [(256, 64, 1, 64), (256, 32, 1, 64), (256, 64, 2, 64), (384, 16, 1, 64), (512, 8, 1, 64), (256, 64, 1, 8)]:
    m = bb.fire_module(c, c, stride=stride).eval()
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = ops.Act(torch.randn((n, hw, hw, c), generator=gen, device="cuda").bfloat16())
    first = m(x).t.clone(); torch.cuda.synchronize()
    bad = 0
    for _ in range(R):
        y = m(x).t; torch.cuda.synchronize()
        bad += 0 if torch.equal(y, first) else 1
    res.append(f"{c}@{hw}s{stride}n{n}:{bad}/{R}")
print(os.environ.get("OKP_LIB", "default").split("/")[-1], " ".join(res), flush=True)
