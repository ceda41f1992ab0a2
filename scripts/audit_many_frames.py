"""KeypointNet.precision_audit over many synthetic frames: the maximum heat-map difference of a configuration against float32x3 on the same
weights, per frame and overall (the committed bounds come from 1-4 frames).  usage: audit_many_frames.py [frames=64] [dtypes=f32mix,f16,bf16] [seeds=0,1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
frames, names, seeds = 64, ["f32mix", "f16", "bf16"], [0, 1]
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "frames": frames = int(v)
    if k == "dtypes": names = v.split(",")
    if k == "seeds": seeds = [int(s) for s in v.split(",")]
for seed in seeds:
    for name in names:
        compute = {"bf16": torch.bfloat16, "f16": torch.float16, "f32mix": ops.F32MIX}[name]
        net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=compute)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=seed).items()})
        net.eval().cuda()
        worst = []
        for start in range(0, frames, 16):
            x = torch.from_numpy(synth.frames(min(16, frames - start), seed=7, start=start)).cuda()
            with torch.no_grad():
                mine = net.deployed(x)[0].float().clone()
                keep = (net.compute_dtype, net.mfma_split, net.mixed)
                net.compute_dtype, net.mfma_split, net.mixed = ops.parse_compute_dtype(ops.F32X3)
                ref = net.deployed(x)[0].float()
                net.compute_dtype, net.mfma_split, net.mixed = keep
            worst += (mine - ref).abs().flatten(1).max(dim=1).values.cpu().tolist()
        w = np.array(worst)
        print(f"weights seed {seed} {name:7s}: heat error max over {frames} frames {w.max():.2e}; per-frame max: median {np.median(w):.2e}, p90 {np.quantile(w, 0.9):.2e}", flush=True)
