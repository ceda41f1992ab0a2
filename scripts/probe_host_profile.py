"""Host-side cost of one eager batch-1 network pass: cProfile of 200 passes (the GPU is never the bottleneck at batch 1 eager)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
net = bench.build_net(torch.bfloat16).to(dev)
x = torch.randn((1, 3, 511, 511), device=dev)
with torch.no_grad():
    for _ in range(5): net.deployed(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200): net.deployed(x)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
