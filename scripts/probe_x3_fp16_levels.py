"""float32x3 with the innermost hourglass levels in fp16 (the mixed configuration's MIX_FP16_LEVELS alone, branches three-term): error against
float32x3 and against the oracle's heat maps is what decides; timing beside it."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=ops.F32X3)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
net.eval().cuda()
x = torch.from_numpy(synth.frames(64, seed=1)).cuda()
def run(cfg, levels):
    net.set_compute_dtype(cfg)
    ops.MIX_FP16_LEVELS = levels
    with torch.no_grad():
        for _ in range(3): out = net.deployed(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = net.deployed(x)
        e1.record(); torch.cuda.synchronize()
    return [o.clone() for o in out], e0.elapsed_time(e1) / 10
ref, t0 = run(ops.F32X3, 2)
print(f"float32x3: {t0:.3f} ms")
ops.MIX_BRANCH_SINGLE = False; ops.MIX_STEM_FP16 = False; ops.MIX_COMPACT = False
for lv in (1, 2, 3):
    out, t = run(ops.F32MIX, lv)
    d = [float((a - b).abs().max()) for a, b in zip(out, ref)]
    print(f"three-term branches, fp16 levels n <= {lv}: {t:.3f} ms; max |diff| to float32x3: heat {d[0]:.2e} depth {d[1]:.2e} centers {d[2]:.2e}")
