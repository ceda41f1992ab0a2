"""Time one fire module launch (okp_fire2).  usage: probe_fire2_time.py [c=256] [hw=64] [n=64] [stride=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
kw = dict(c=256, co=0, hw=64, n=64, stride=1)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = int(v)
m = bb.fire_module(kw["c"], kw["co"] or kw["c"], stride=kw["stride"]).eval()
x = ops.Act(torch.randn((kw["n"], kw["hw"], kw["hw"], kw["c"]), device="cuda").bfloat16())
res = []
for rep in range(3):
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): m(x)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"fire {kw}: " + " ".join(f"{v:.1f}" for v in res) + " us")
