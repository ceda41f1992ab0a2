"""Stress: the deployed outputs of a 64-frame batch must be bit-identical over many repetitions (side streams on, eager), for every
precision configuration.  usage: stress_determinism.py [reps=30] [dtypes=bf16,f16,f32mix,f32x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from object_keypoints_amd import ops
reps, names = 30, ["bf16", "f16", "f32mix", "f32x3"]
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "reps": reps = int(v)
    if k == "dtypes": names = v.split(",")
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(11)
x = torch.randn((64, 3, 511, 511), device=dev, generator=gen)
bad = 0
with torch.no_grad():
    for name in names:
        compute = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32x3": ops.F32X3, "f32mix": ops.F32MIX}[name]
        net = bench.build_net(compute).to(dev)
        first = [t.clone() for t in net.deployed(x)]
        diff = 0
        for r in range(reps):
            out = net.deployed(x)
            torch.cuda.synchronize()
            for a, b in zip(first, out):
                if not torch.equal(a, b):
                    diff += 1
                    print(f"{name}: repetition {r} differs in {int((a != b).sum())} elements, max {float((a - b).abs().max()):.3g}", flush=True)
        print(f"{name}: {reps} repetitions of 64 frames, {diff} mismatching outputs; finite: {all(bool(torch.isfinite(t).all()) for t in first)}", flush=True)
        bad += diff
        # tensor lifetimes across the side streams (hg_module.forward keeps branch outputs alive by program order, not by
        # Tensor.record_stream): passes of changing batch size make the caching allocator hand the same blocks to other tensors;
        # every pass must equal the pass of the same frames with the branches on the main stream
        diff = 0
        for r in range(max(4, reps // 4)):
            for b in (64, 24, 48, 8, 56):
                xb = x[:b]
                got = [t.clone() for t in net.deployed(xb)]
                ops.SIDE_STREAMS = False
                want = net.deployed(xb)
                ops.SIDE_STREAMS = True
                torch.cuda.synchronize()
                diff += sum(0 if torch.equal(a, c) else 1 for a, c in zip(got, want))
        print(f"{name}: changing batch sizes, side streams against one stream: {diff} mismatching outputs", flush=True)
        bad += diff
        del net
sys.exit(1 if bad else 0)
