"""Back-to-back launches on one stream: where do the 6-19 us gaps between kernels come from?  Run under rocprofv3 --kernel-trace
(scripts/probe_gaps.sh); sequences are separated by a 1 ms sleep so they can be told apart in the trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
from object_keypoints_amd.perception.backbone import conv_taps
n, hw = 64, 64
rng = np.random.default_rng(0)
wt = (rng.standard_normal((256, 256, 3, 3)) / np.sqrt(256 * 9)).astype(np.float32)
plan = ops.ConvPlan(torch.bfloat16, [256], [1], 256, conv_taps(wt), np.zeros(256, np.float32), relu=True)
x = ops.Act(torch.randn(n, hw, hw, 256, device="cuda").bfloat16())
o1 = ops.Act.empty(n, hw, hw, 256, torch.bfloat16, x.t.device)
o2 = ops.Act.empty(n, hw, hw, 256, torch.bfloat16, x.t.device)
fire = bb.fire_module(256, 256).eval()
x32 = ops.Act(torch.randn(n, 32, 32, 256, device="cuda").bfloat16())
def P(a, b): plan([a], b, hw, hw, tile=13)
def F(a): return fire(a)
seqs = {
    "PPPP": lambda: (P(x, o1), P(o1, o2), P(o2, o1), P(o1, o2)),
    "PPPP_indep": lambda: (P(x, o1), P(x, o1), P(x, o1), P(x, o1)),
    "FFFF64": lambda: F(F(F(F(x)))),
    "FFFF32": lambda: F(F(F(F(x32)))),
    "PFPF": lambda: (P(x, o1), F(o1), P(x, o2), F(o2)),
    "P6P6": lambda: (plan([x], o1, hw, hw, tile=6), plan([o1], o2, hw, hw, tile=6), plan([o2], o1, hw, hw, tile=6)),
}
for name, f in seqs.items():
    for _ in range(3):
        f(); torch.cuda.synchronize(); time.sleep(0.002)
    print(name, flush=True)
