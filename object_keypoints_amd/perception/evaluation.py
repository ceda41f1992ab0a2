"""Accuracy bookkeeping for evaluation runs: what the reference's harness tabulates (scripts/eval_model.py:129-232).

Per frame, every detected object is matched to the ground-truth object whose centre is nearest in the camera's x/y plane
(depth is the least certain coordinate), every lifted keypoint to the nearest ground-truth keypoint of that object, and
the Euclidean errors are collected; keypoints that were not lifted count as missing.  `summary()` returns the columns of
the reference's table (centimetres; "< 3cm" and "missing" as the reference prints them).  Host NumPy: a handful of points
per frame, nothing here is on the per-frame device path.
"""
import numpy as np

from .utils import linalg

_MAX_RANGE = 2.0          # predictions with a coordinate beyond 2 m are treated as not lifted (eval_model.py:170)
_SMALL = 0.03             # the "< 3cm" column


class Results:
    COLUMNS = ("mean", "mean xy", "std", "< 3cm", "25th percentile", "75th percentile", "missing", "points")

    def __init__(self):
        self._err = []            # metres, one entry per matched keypoint
        self._err_xy = []
        self._missing = 0
        self._points = 0
        self.camera = None

    def set_calibration(self, camera):
        self.camera = camera

    # -- helpers -------------------------------------------------------------------------------------------------
    def _visible(self, points_C):
        px = self.camera.project(np.atleast_2d(points_C))
        return self.camera.in_frame(px)

    def _match_object(self, detection, truth_C):
        centre_xy = np.asarray(detection["p_C"][0][0])[:2]
        return truth_C[np.linalg.norm(truth_C[:, 0, :2] - centre_xy, axis=1).argmin()]

    # -- API of the reference's class ---------------------------------------------------------------------------------
    def add(self, T_WC, objects, scene_points):
        """T_WC: camera-to-world transform (4x4); objects: the keypoint pipeline's output for one frame (dicts with 'p_C':
        per keypoint type an (n,3) array, a list with None entries, or None); scene_points: (n_objects, n_keypoints, 3)
        ground truth in the world frame, index 0 = object centre."""
        truth_C = linalg.transform_points(linalg.inv_transform(np.asarray(T_WC)), np.asarray(scene_points, dtype=np.float64))
        for det in objects:
            target = self._match_object(det, truth_C)
            if not self._visible(target[0])[0]:
                continue                                           # the object's centre is outside the frame: skipped
            for group in det["p_C"]:
                if group is None:
                    continue
                for p in group:
                    if p is None or not (np.asarray(p) < _MAX_RANGE).all():
                        self._points += 1
                        self._missing += 1
                        continue
                    p = np.asarray(p, dtype=np.float64)
                    nearest = target[np.linalg.norm(target - p, axis=1).argmin()]
                    if not self._visible(nearest)[0]:
                        continue                                   # its ground truth is out of view: not scored
                    self._points += 1
                    self._err.append(float(np.linalg.norm(nearest - p)))
                    self._err_xy.append(float(np.linalg.norm(nearest[:2] - p[:2])))

    def summary(self):
        n = self._points
        if n == 0 or not self._err:
            return {"points": n, "missing": 100.0 if n else 0.0}
        cm, cm_xy = np.array(self._err) * 100.0, np.array(self._err_xy) * 100.0
        q25, q75 = np.percentile(cm, [25, 75])
        return {"mean": float(cm.mean()), "mean xy": float(cm_xy.mean()), "std": float(cm.std()),
                "< 3cm": float((np.array(self._err) < _SMALL).sum()) / n, "25th percentile": float(q25),
                "75th percentile": float(q75), "missing": 100.0 * self._missing / n, "points": n}

    def print_results(self):
        s = self.summary()
        print(" | ".join(self.COLUMNS))
        print(" | ".join(f"{s[c]:.4g}" if isinstance(s.get(c), float) else str(s.get(c, "-")) for c in self.COLUMNS))
