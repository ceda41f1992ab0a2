"""What the deep hourglass levels cost on the critical path: the network pass with hg_module levels n <= LEVEL replaced by their up1 branch
alone (wrong results, right timing of everything else).  usage: ablate_deep_levels.py [dtype=f32x3] [n=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception import backbone
from object_keypoints_amd.perception.models import KeypointNet
kw = dict(dtype="f32x3", n=64)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = v if k == "dtype" else int(v)
dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32x3": ops.F32X3, "f32mix": ops.F32MIX}[kw["dtype"]]
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=dtype)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
net.eval().cuda()
x = torch.from_numpy(synth.frames(kw["n"], seed=1)).cuda()
orig = backbone.hg_module.forward
CUT = [0]
def patched(self, x, out_pairs=False):
    if self.n <= CUT[0]:
        return self.up1(x)
    return orig(self, x, out_pairs=out_pairs)
backbone.hg_module.forward = patched
def run():
    with torch.no_grad():
        for _ in range(5): net.deployed(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): net.deployed(x)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
    return sorted(ts)[len(ts) // 2]
for cut, what in ((0, "complete network"), (1, "without the 8x8 level's low path (4x4 modules)"), (2, "without the 16x16 level's low path (8x8 and below)"),
                  (3, "without the 32x32 level's low path (16x16 and below)"), (0, "complete network")):
    CUT[0] = cut
    print(f"{kw['dtype']} n={kw['n']}  {what:60s} {run():8.3f} ms per pass")
