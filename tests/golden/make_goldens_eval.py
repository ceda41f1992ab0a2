#!/usr/bin/env python3
"""Golden vector for the evaluation bookkeeping: runs the REFERENCE's `Results` class
(scripts/eval_model.py:129-232, read-only at /root/reference) on a synthetic multi-frame scene and stores the inputs it was
given and the row of its table.  The camera handed to it is the oracle's equidistant camera (the reference's own
FisheyeCamera.project needs cv2, which is not importable here); `Results` only calls `project` and `in_frame` on it.

    python tests/golden/make_goldens_eval.py      ->  tests/golden/evaluation.json
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from _ref_import import import_reference, REFERENCE_ROOT  # noqa: E402


def scene(seed=5, frames=6):
    """Ground-truth objects (centre + 3 keypoints each, world frame), camera poses, and detections = truth in the camera frame
    plus known perturbations; a few keypoints not lifted (None), one beyond the 2 m range, one object out of view."""
    rng = np.random.default_rng(seed)
    truth = np.array([[[0.00, 0.00, 1.00], [0.06, 0.00, 1.00], [0.00, 0.06, 1.02], [-0.05, -0.04, 0.98]],
                      [[0.25, -0.10, 1.20], [0.31, -0.10, 1.22], [0.25, -0.04, 1.18], [0.20, -0.15, 1.21]],
                      [[-0.30, 0.15, 0.90], [-0.24, 0.15, 0.92], [-0.30, 0.21, 0.90], [-0.34, 0.10, 0.88]]])
    out = []
    for f in range(frames):
        ang = 0.05 * f
        T_WC = np.eye(4)
        T_WC[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        T_WC[:3, 3] = [0.02 * f, -0.01 * f, -0.05 * f]
        R, t = T_WC[:3, :3], T_WC[:3, 3]
        truth_C = (truth - t) @ R                 # inverse transform of row vectors
        objects = []
        for o in range(truth.shape[0]):
            if (f + o) % 5 == 4:
                continue                                            # detection missed
            pts = truth_C[o] + rng.normal(0, 0.012, truth_C[o].shape)
            groups = [pts[0:1], pts[1:2], pts[2:4].copy()]
            if (f + o) % 3 == 0:
                groups[1] = None                                    # a keypoint type without detection
            if f == 2 and o == 1:
                groups[2] = [groups[2][0], None]                    # one of two keypoints not lifted
            if f == 3 and o == 0:
                groups[2][1] = groups[2][1] + np.array([0.0, 0.0, 1.5])      # beyond the 2 m range: counted as missing
            objects.append({"p_C": groups})
        out.append({"T_WC": T_WC, "objects": objects})
    return truth, out


def to_json(v):
    if v is None:
        return None
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [to_json(x) for x in v]
    if isinstance(v, dict):
        return {k: to_json(x) for k, x in v.items()}
    return v


def main():
    ref = import_reference()
    for name in ("hud", "rospy"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.Rect = lambda *a, **k: None
            sys.modules[name] = m
    a = sys.modules["albumentations"]
    for n in ("Compose", "SmallestMaxSize", "CenterCrop", "KeypointParams", "RandomBrightnessContrast", "HueSaturationValue", "RandomGamma",
              "GaussNoise", "MotionBlur", "Blur", "ImageCompression", "Rotate", "RandomResizedCrop", "Resize", "HorizontalFlip", "ColorJitter", "Affine"):
        setattr(a, n, lambda *x, **k: None)
    sys.path.insert(0, os.path.join(REFERENCE_ROOT, "scripts"))
    import importlib
    ev = importlib.import_module("eval_model")
    from oracle import pipeline as op
    cam = op.eval_camera(os.path.join(REPO, "config", "calibration.yaml"))
    truth, frames = scene()
    rows = []
    from rich.table import Table
    orig = Table.add_row
    Table.add_row = lambda self, *cells, **k: (rows.append(list(cells)), orig(self, *cells, **k))[1]
    res = ev.Results()
    res.set_calibration(cam)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        for fr in frames:
            res.add(fr["T_WC"], fr["objects"], truth)
        try:
            res.print_results()
        except Exception as e:              # the rich screen object needs a live terminal: the row was captured before it is shown
            if not rows:
                raise
    Table.add_row = orig
    cols = ["mean", "mean xy", "std", "< 3cm", "25th percentile", "75th percentile", "missing", "points"]
    row = dict(zip(cols, rows[-1]))
    out = {"generator": "tests/golden/make_goldens_eval.py (reference scripts/eval_model.py:129-232 Results)",
           "truth": truth.tolist(), "frames": to_json(frames), "row": row}
    with open(os.path.join(HERE, "evaluation.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(row)


if __name__ == "__main__":
    main()
