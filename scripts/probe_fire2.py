"""Time the 16-bit one-launch fire module (okp_fire2) alone.  usage: probe_fire2.py [hw=64] [n=64] [c=256] [co=256] [stride=1] [dtype=bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
kw = dict(hw=64, n=64, c=256, co=256, stride=1, dtype="bf16")
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = v if k == "dtype" else int(v)
dt = torch.bfloat16 if kw["dtype"] == "bf16" else torch.float16
m = bb.fire_module(kw["c"], kw["co"], stride=kw["stride"]).eval()
x = ops.Act(torch.randn(kw["n"], kw["hw"], kw["hw"], kw["c"], device="cuda").to(dt))
for _ in range(5): out = m(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): m(x)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
mb = (x.t.numel() + out.t.numel()) * 2 / 1e6
print(f"fire({kw['c']}, {kw['co']}) stride {kw['stride']} at {kw['hw']}x{kw['hw']}, N={kw['n']}: {us:7.1f} us   {mb / us:5.2f} TB/s of x + out ({mb:.0f} MB)")
