"""Host time of one eager batch-1 network pass (launch rate only: the GPU is synchronised outside the timed region... it is not -
at batch 1 the eager pass is host-bound, so wall time per pass = host time per pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from object_keypoints_amd import ops
dev = torch.device("cuda", 0)
net = bench.build_net(torch.bfloat16).to(dev)
x = torch.randn((1, 3, 511, 511), device=dev)
def run(n=300):
    with torch.no_grad():
        for _ in range(10): net.deployed(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): net.deployed(x)
        t1 = time.perf_counter()            # host done enqueueing
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
for side in (True, False, True, False):
    ops.SIDE_STREAMS = side
    h, w = run()
    print(f"side streams {side}: host {h:.3f} ms per pass, wall {w:.3f} ms per pass, {ops.COUNTERS['launches']} launches so far")
