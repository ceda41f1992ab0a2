"""Micro-benchmark of one implicit-GEMM convolution through the C ABI (for rocprofv3 / tuning).
usage: bench_conv.py [k=3] [cin=256] [cout=256] [hw=64] [n=64] [stride=1] [dtype=bf16] [tile=0] [iters=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception.backbone import conv_taps, conv_out_size
kw = dict(k=3, cin=256, cout=256, hw=64, n=64, stride=1, dtype="bf16", tile=0, iters=20)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = v if k == "dtype" else int(v)
dtype = torch.bfloat16 if kw["dtype"] == "bf16" else torch.float32
k, cin, cout, hw, n, stride = kw["k"], kw["cin"], kw["cout"], kw["hw"], kw["n"], kw["stride"]
w = (np.random.default_rng(0).standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
plan = ops.ConvPlan(dtype, [cin], [stride], cout, conv_taps(w), np.zeros(cout, np.float32), relu=True)
x = ops.Act(torch.randn(n, hw, hw, cin, device="cuda").to(dtype))
ho = conv_out_size(hw, k, stride, (k - 1) // 2)
out = ops.Act.empty(n, ho, ho, cout, dtype, x.t.device)
for _ in range(3): plan([x], out, ho, ho, tile=kw["tile"])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(kw["iters"]): plan([x], out, ho, ho, tile=kw["tile"])
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / kw["iters"]
flops = 2.0 * n * ho * ho * cout * cin * k * k
print(f"{kw}: {ms*1e3:.1f} us/launch  {flops/ms/1e9:.1f} TFLOP/s")
