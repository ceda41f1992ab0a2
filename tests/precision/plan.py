"""TEST INFRASTRUCTURE (CPU): which float32mix plan do given weights allow?

The mixed configuration trades precision for speed in two places whose cost depends on the weights: single-term products inside the trunk's
residual branches (ops.MIX_BRANCH_SINGLE) and fp16 inner hourglass levels (OKP_MIX_FP16_LEVELS).  This tool runs the rounding-point model
(emulate.py) of every combination on a few frames and prints the predicted heat / depth error next to the speed each plan measured on
MI355X, then names the fastest plan inside the bound.

    python tests/precision/plan.py [--state-dict model.pt] [--heatmaps 3] [--bound 5.5e-4] [--frames 2]

Without --state-dict the synthetic test network is used (the default plan of object_keypoints_amd must come out: tests/test_precision_emulation.py).
"""
import argparse
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import emulate                                           # noqa: E402
from object_keypoints_amd import synth                   # noqa: E402

# frames/s measured on one MI355X box (profiles/r03o_f32mix_fp16_levels.txt; relative order is what matters here)
MEASURED = {(0, True): 5493, (1, True): 5637, (2, True): 5711, (3, True): 5945, (4, True): 6906, (2, False): 3832}


def evaluate(emu, x, frames_ref, levels, branch):
    out = emu.forward(x, emulate.mixed_policy(levels, branch, branch))
    return float((out[0] - frames_ref[0]).abs().max()), float((out[1] - frames_ref[1]).abs().max())


def choose(emu, x, bound):
    ref = emu.forward(x, emulate.Policy(None))
    rows = []
    for (levels, branch), fps in sorted(MEASURED.items(), key=lambda kv: -kv[1]):
        heat, depth = evaluate(emu, x, ref, levels, branch)
        rows.append({"fp16_levels": levels, "branch_single_term": branch, "frames_per_s": fps, "heat_max": heat, "depth_max": depth, "ok": heat <= bound})
    best = next((r for r in rows if r["ok"]), None)
    return rows, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--state-dict")
    ap.add_argument("--heatmaps", type=int, default=3)
    ap.add_argument("--bound", type=float, default=5.5e-4)
    ap.add_argument("--frames", type=int, default=2)
    a = ap.parse_args()
    net, emu = emulate.build(a.heatmaps, 0)
    if a.state_dict:
        sd = torch.load(a.state_dict, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        net.load_state_dict({k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()})
        emu = emulate.EmuNet(net.eval())
    x = torch.from_numpy(synth.frames(a.frames, seed=1))
    rows, best = choose(emu, x, a.bound)
    for r in rows:
        print(f"fp16 levels {r['fp16_levels']} (OKP_MIX_FP16_LEVELS), single-term branches {int(r['branch_single_term'])} (ops.MIX_BRANCH_SINGLE): {r['frames_per_s']:5d} frames/s, "
              f"heat max {r['heat_max']:.2e}, depth max {r['depth_max']:.2e} {'<= bound' if r['ok'] else ''}")
    print("fastest plan inside the bound:", best and {k: best[k] for k in ("fp16_levels", "branch_single_term")})


if __name__ == "__main__":
    main()
