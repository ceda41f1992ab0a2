"""Experiment: one 64-frame batch as ONE pass vs. two 32-frame passes on two HIP streams (the idle chip time of one half's hourglass
low levels could be filled by the other half's trunk convolutions).  usage: two_stream_split.py [dtype=bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from object_keypoints_amd import ops
precision = "bf16"
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "dtype": precision = v
dev = torch.device("cuda", 0)
compute = {"bf16": torch.bfloat16, "f16": torch.float16, "f32mix": ops.F32MIX}[precision]
net = bench.build_net(compute).to(dev)
frames = torch.randn((64, 3, 511, 511), device=dev)
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
with torch.no_grad():
    one = timed(lambda: net.deployed(frames))
    for parts in (2, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        chunks = frames.chunk(parts)
        def split():
            main = torch.cuda.current_stream()
            for s, c in zip(streams, chunks):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    net.deployed(c)
            for s in streams: main.wait_stream(s)
        t = timed(split)
        def serial():
            for c in chunks: net.deployed(c)
        ts = timed(serial)
        print(f"{precision}: one pass of 64: {one:.3f} ms; {parts} passes of {64 // parts} on {parts} streams: {t:.3f} ms; the same passes one after the other: {ts:.3f} ms")
