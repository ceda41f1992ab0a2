"""TEST INFRASTRUCTURE (CPU): what the shipped float32mix plan costs on weights it was NOT derived on.

The plan (fp32 skip stream with three-term products, single-term fp16 residual branches, fp16 hourglass levels n <= 2) was chosen
from the attribution table of ONE synthetic weight family - object_keypoints_amd.synth.fill_state_dict(branch_gain=0.3) - in
which the BatchNorm that closes every residual / fire branch has a gain of 0.3 and so attenuates whatever error the branch carries.
No trained weights exist in the reference tree.  This script prices the same plan with the rounding-point model
(tests/precision/emulate.py) on other families:

    derived-on       fill_state_dict(seed 0, branch_gain 0.3, head_gain 0.45)            (the test / bench network)
    other-seed       ... seed 3
    head-gain-x2     head_gain = 0.9: logits twice as large (steeper sigmoids)
    torch-default    BatchNorm gamma = 1, beta = 0 (torch's initialisation), running statistics CALIBRATED on two frames (every
                     BatchNorm then normalises its input to unit variance, as in a trained network): branches enter the stream with
                     gain 1, un-attenuated
    branch-gain-1    the synthetic gamma / beta ranges with branch_gain = 1.0, running statistics calibrated the same way
    (the un-calibrated forms of the last two - random running statistics that do not normalise anything - let the activations grow
     to 1e11 through the two hourglasses: not a network anyone would run, in any precision)

and writes tests/golden/precision_families.json: per family the heat / depth error of float32mix, all-fp16 and float32x3 against the
exact-fp32 run of the same weights (max over the frames, mean), the largest activation magnitude (fp16 operands overflow at 65504) and
whether the configuration is inside the 1e-3 heat bar.  tests/test_precision_emulation.py asserts the table's conclusions;
perception.pipeline.load_keypoint_net(audit_frames=...) is the product-side answer (on-device audit, fall-back to float32x3).

usage: python tests/precision/families.py [n_frames=2]        (10-20 s per family and frame pair on 8 cores)
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import emulate          # noqa: E402

FAMILIES = {
    "derived-on": dict(seed=0, branch_gain=0.3, head_gain=0.45),
    "other-seed": dict(seed=3, branch_gain=0.3, head_gain=0.45),
    "head-gain-x2": dict(seed=0, branch_gain=0.3, head_gain=0.9),
    "torch-default": dict(seed=0, branch_gain=1.0, head_gain=0.45, bn_identity=True, calibrate=True),
    "branch-gain-1": dict(seed=0, branch_gain=1.0, head_gain=0.45, calibrate=True),
}
CALIBRATION_FRAMES = dict(n=2, seed=11)


def family_state_dict(shapes, seed=0, branch_gain=0.3, head_gain=0.45, bn_identity=False, calibrate=False):
    """{key: np.ndarray} of one weight family (object_keypoints_amd.synth rules; bn_identity: gamma 1 / beta 0 on every BatchNorm)."""
    from object_keypoints_amd import synth
    vals = synth.fill_state_dict(shapes, seed=seed, branch_gain=branch_gain, head_gain=head_gain)
    if bn_identity:
        for k in list(vals):
            prefix, _, leaf = k.rpartition(".")
            if prefix + ".running_mean" in vals and leaf in ("weight", "bias"):
                vals[k] = np.full_like(vals[k], 1.0 if leaf == "weight" else 0.0)
    return vals


def calibrate_batchnorm(net):
    """Set every BatchNorm's running statistics to the statistics of its input on the calibration frames (one forward with the
    BatchNorm layers in training mode at momentum 1, everything else in eval mode): what training leaves behind."""
    from object_keypoints_amd import synth
    net.eval()
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    keep = [m.momentum for m in bns]
    for m in bns:
        m.train()
        m.momentum = 1.0
    with torch.no_grad():
        net(torch.from_numpy(synth.frames(CALIBRATION_FRAMES["n"], seed=CALIBRATION_FRAMES["seed"])))
    for m, mo in zip(bns, keep):
        m.momentum = mo
    return net.eval()


def build_family(name, heatmaps_out=3):
    """-> (oracle KeypointNet with the family's weights, its rounding-point emulator)."""
    from oracle import net as onet
    net = onet.KeypointNet(features=128, heatmaps_out=heatmaps_out)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    vals = family_state_dict(shapes, **FAMILIES[name])
    net.load_state_dict({k: torch.from_numpy(v.copy()) if v.ndim else torch.tensor(int(v)) for k, v in vals.items()})
    if FAMILIES[name].get("calibrate"):
        calibrate_batchnorm(net)
    return net.eval(), emulate.EmuNet(net.eval())


class _Magnitude(emulate.Policy):
    """Exact-fp32 policy that also records the largest stored activation (what an fp16 operand would have to hold)."""

    def __init__(self):
        super().__init__(None)
        self.amax = 0.0

    def act(self, name, t):
        self.amax = max(self.amax, float(t.abs().max()))
        return super().act(name, t)


def price(name, n_frames=2):
    from object_keypoints_amd import ops, synth
    _, emu = build_family(name)
    x = torch.from_numpy(synth.frames(n_frames, seed=1))
    mag = _Magnitude()
    ref = [t.numpy().astype(np.float64) for t in emu.forward(x, mag)]
    row = {"family": FAMILIES[name], "frames": n_frames, "activation_abs_max": mag.amax,
           "logit_note": "heat = sigmoid(logits): saturated maps hide logit error, steep ones amplify it"}
    for tag, pol in (("float32mix", emulate.mixed_policy(ops.MIX_FP16_LEVELS, ops.MIX_BRANCH_SINGLE, ops.MIX_STEM_FP16)),
                     ("float16", emulate.Policy(torch.float16)), ("float32x3", emulate.Policy(None, x3=True))):
        got = [t.numpy().astype(np.float64) for t in emu.forward(x, pol)]
        e = {k: np.abs(g.reshape(r.shape) - r) for k, g, r in zip(("heat", "depth", "centers"), got, ref)}
        per_frame = e["heat"].reshape(n_frames, -1).max(axis=1)
        row[tag] = {"heat_max": float(e["heat"].max()), "heat_mean": float(e["heat"].mean()), "heat_max_per_frame": [float(v) for v in per_frame],
                    "depth_max": float(e["depth"].max()), "depth_mean": float(e["depth"].mean()),
                    "finite": bool(all(np.isfinite(g).all() for g in got)), "meets_heat_1e-3": bool(e["heat"].max() <= 1e-3)}
    return row


if __name__ == "__main__":
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.set_num_threads(os.cpu_count() or 1)
    out_path = os.path.join(REPO, "tests", "golden", "precision_families.json")
    table = {}
    for name in FAMILIES:
        t0 = time.time()
        table[name] = price(name, n_frames)
        r = table[name]
        print(f"{name:14s} |act| max {r['activation_abs_max']:9.1f}  f32mix heat {r['float32mix']['heat_max']:.2e} (mean {r['float32mix']['heat_mean']:.1e})  "
              f"fp16 {r['float16']['heat_max']:.2e}  f32x3 {r['float32x3']['heat_max']:.2e}   [{time.time() - t0:.0f} s]", flush=True)
        with open(out_path, "w") as f:
            json.dump({"generator": "tests/precision/families.py", "bar": 1e-3, "families": table}, f, indent=1)
