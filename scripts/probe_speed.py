import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet

def build(dtype):
    net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=0)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in vals.items()})
    return net.eval()

for dtype, batches in ((torch.bfloat16, (1, 8, 64)), (torch.float32, (1, 16))):
    net = build(dtype)
    for n in batches:
        x = torch.randn(n, 3, 511, 511, device="cuda")
        for _ in range(2): net.deployed(x)
        torch.cuda.synchronize()
        ops.COUNTERS["launches"] = 0
        t = time.time(); iters = 5
        for _ in range(iters): net.deployed(x)
        torch.cuda.synchronize()
        dt = (time.time() - t) / iters
        print(f"{dtype} batch {n}: {dt*1e3:.2f} ms/fwd  {n/dt:.1f} frames/s  {74.565e9*n/dt/1e12:.1f} TFLOP/s  launches/fwd {ops.COUNTERS['launches']//iters}", flush=True)
