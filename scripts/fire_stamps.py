"""One launch of the 64x64 / 32x32 fire module 256 -> 128 -> 256 in a -DOKP_PATCH_STAMPS build: how long the two workgroups of a CU run.
usage: OKP_LIB=.../libokp_hip_S.so python scripts/fire_stamps.py [hw=64] [n=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.pop("OKP_PATCH_STAMPS_PRINT", None)
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
kw = dict(hw=64, n=64)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = int(v)
m = bb.fire_module(256, 256).eval()
x = ops.Act(torch.randn(kw["n"], kw["hw"], kw["hw"], 256, device="cuda").bfloat16())
for _ in range(3): y = m(x)
torch.cuda.synchronize()
os.environ["OKP_PATCH_STAMPS_PRINT"] = "1"
y = m(x)
torch.cuda.synchronize()
