"""Read-before-write hunt: poison the caching allocator's free blocks with NaN bit patterns, run the forward pass, and
compare with the un-poisoned result.  usage: probe_poison.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=torch.bfloat16)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
net.eval().cuda()
x = torch.randn(n, 3, 511, 511, device="cuda")
base = [t.clone() for t in net.deployed(x)]
torch.cuda.synchronize()
def poison():
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    blocks = []
    for gb in (8, 4, 2, 1, 1, 1, 0.5, 0.5, 0.25, 0.25):
        try:
            b = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda"); b.fill_(0xFF); blocks.append(b)
        except Exception:
            pass
    torch.cuda.synchronize()
    del blocks
for rep in range(2):
    poison()
    out = net.deployed(x)
    torch.cuda.synchronize()
    print(f"poisoned run {rep}:", [bool(torch.equal(a, b)) for a, b in zip(base, out)], "nan:", [bool(torch.isnan(t).any()) for t in out])
