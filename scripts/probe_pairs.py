"""Pair-format tensors on the split-product patch kernel (okp_conv_args.src_pairs / out_pairs): one 3x3 -> 256 channels with fp32 / pair
sources and fp32 / pair output, alternating in one process.
usage: probe_pairs.py [hw=64] [n=64] [res=0] [cin=256] [stride=1] [skip=0] [rounds=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception.backbone import conv_taps
kw = dict(hw=64, n=64, res=0, cin=256, stride=1, skip=0, rounds=4)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = int(v)
n, hw, cin, st, skip = kw["n"], kw["hw"], kw["cin"], kw["stride"], kw["skip"]
rng = np.random.default_rng(0)
wt = (rng.standard_normal((256, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
taps, cins, strides = conv_taps(wt), [cin], [st]
srcs = [ops.Act(torch.randn(n, hw * st, hw * st, cin, device="cuda"))]
if skip:
    ws = (rng.standard_normal((256, skip)) / np.sqrt(skip)).astype(np.float32)
    taps = taps + [(1, 0, 0, ws)]; cins.append(skip); strides.append(2)
    srcs.append(ops.Act(torch.randn(n, hw * 2, hw * 2, skip, device="cuda")))
with ops.f32_split():
    plan = ops.ConvPlan(torch.float32, cins, strides, 256, taps, np.zeros(256, np.float32), relu=True)
psrcs = [ops.Act.float_to_pairs(s.t) for s in srcs]
r = ops.Act(torch.randn(n, hw, hw, 256, device="cuda")) if kw["res"] else None
out = ops.Act.empty(n, hw, hw, 256, torch.float32, srcs[0].t.device)
variants = {"fp32 in, fp32 out": (srcs, False), "pairs in, fp32 out": (psrcs, False), "fp32 in, pairs out": (srcs, True), "pairs in, pairs out": (psrcs, True)}
for _ in range(30): plan(srcs, out, hw, hw, res=r, tile=13)       # warm clocks
res = {k: [] for k in variants}
for rd in range(kw["rounds"]):
    for k in (list(variants) if rd % 2 == 0 else list(variants)[::-1]):
        s, op = variants[k]
        for _ in range(3): plan(s, out, hw, hw, res=r, tile=13, out_pairs=op)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan(s, out, hw, hw, res=r, tile=13, out_pairs=op)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
print(" ".join(f"{k}={v}" for k, v in kw.items()))
for k, v in res.items():
    print(f"  {k:20s} median {sorted(v)[len(v) // 2]:7.1f} us   all {' '.join('%.1f' % t for t in v)}")
