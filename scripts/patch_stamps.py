"""One launch of a patch-resident convolution in a build with -DOKP_PATCH_STAMPS (OKP_EXTRA_CFLAGS): the launcher prints the shader-clock
stamps of every K-step (wait for the LDS-DMA | barrier | until the first MFMA group was issued | rest of the step body).
usage: OKP_LIB=.../libokp_hip_S.so OKP_PATCH_STAMPS_PRINT=1 python scripts/patch_stamps.py [hw=64] [n=64] [cin=256] [stride=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pr = os.environ.pop("OKP_PATCH_STAMPS_PRINT", None)
from object_keypoints_amd import ops
from object_keypoints_amd.perception.backbone import conv_taps
kw = dict(hw=64, n=64, res=0, cin=256, stride=1)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = int(v)
n, hw, cin, st = kw["n"], kw["hw"], kw["cin"], kw["stride"]
rng = np.random.default_rng(0)
wt = (rng.standard_normal((256, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
plan = ops.ConvPlan(torch.bfloat16, [cin], [st], 256, conv_taps(wt), np.zeros(256, np.float32), relu=True)
x = ops.Act(torch.randn(n, hw * st, hw * st, cin, device="cuda").bfloat16())
out = ops.Act.empty(n, hw, hw, 256, torch.bfloat16, x.t.device)
for _ in range(5): plan([x], out, hw, hw, tile=13)
torch.cuda.synchronize()
os.environ["OKP_PATCH_STAMPS_PRINT"] = "1"
plan([x], out, hw, hw, tile=13)
torch.cuda.synchronize()
