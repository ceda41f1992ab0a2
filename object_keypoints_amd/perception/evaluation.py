"""Accuracy bookkeeping of the reference's evaluation harness (scripts/eval_model.py:129-232, class `Results`):
pairs every predicted 3D keypoint with the nearest ground-truth keypoint of the nearest object and reports the error
statistics of the reference's table.  Host-side NumPy (a few points per frame); same `add` / `set_calibration`
signatures, `summary()` returns the numbers the reference prints (`print_results` renders them as text)."""
import numpy as np

from .utils import linalg


class Results:
    COLUMNS = ("mean", "mean xy", "std", "< 3cm", "25th percentile", "75th percentile", "missing", "points")

    def __init__(self):
        self.gt_keypoints = []
        self.predicted_keypoints = []
        self.camera = None

    def set_calibration(self, camera):
        self.camera = camera

    def add(self, T_WC, objects, scene_points):
        """T_WC: camera-to-world transform; objects: output of the keypoint pipeline for one frame (dicts with 'p_C');
        scene_points: (n_objects, n_keypoints, 3) ground-truth keypoints in the world frame, index 0 = object centre."""
        gt_keypoints, keypoints = [], []
        T_CW = linalg.inv_transform(T_WC)
        scene_points_C = linalg.transform_points(T_CW, scene_points)
        centers_C = scene_points_C[:, 0]
        for obj in objects:
            p_CK = obj['p_C']
            # depth is disregarded when choosing the object: it is the least certain coordinate (eval_model.py:152-153)
            object_distances = np.linalg.norm(centers_C[:, :2] - p_CK[0][0][:2], axis=1)
            object_points = scene_points_C[object_distances.argmin()]
            gt_center = self.camera.project(object_points[0:1])
            if not self.camera.in_frame(gt_center)[0]:
                continue                                   # object centre not in view
            gt_points, object_keypoints = [], []
            for points in p_CK:
                if points is None:
                    continue
                for point in points:
                    if point is not None and (point < 2.0).all():
                        gt_point = object_points[np.linalg.norm(object_points - point, axis=1).argmin()]
                        if (self.camera.in_frame(self.camera.project(gt_point[None])) == False).any():   # noqa: E712
                            continue
                        object_keypoints.append(point)
                        gt_points.append(gt_point)
                    else:
                        object_keypoints.append(None)
                        gt_points.append(None)
            gt_keypoints.append(gt_points)
            keypoints.append(object_keypoints)
        self.gt_keypoints.append(gt_keypoints)
        self.predicted_keypoints.append(keypoints)

    def summary(self):
        """dict with the columns of the reference's table (errors in cm, ratios as the reference prints them)."""
        errors, errors_xy = [], []
        missing = n_points = small_error = 0
        for gt, predicted in zip(self.gt_keypoints, self.predicted_keypoints):
            assert len(gt) == len(predicted)
            for gt_points, p_points in zip(gt, predicted):
                assert len(gt_points) == len(p_points)
                for gt_point, p_point in zip(gt_points, p_points):
                    n_points += 1
                    if p_point is not None:
                        error = float(np.linalg.norm(gt_point - p_point))
                        errors.append(error)
                        errors_xy.append(float(np.linalg.norm(gt_point[:2] - p_point[:2])))
                        small_error += error < 0.03
                    else:
                        missing += 1
        if n_points == 0 or not errors:
            return {"points": n_points, "missing": 100.0 if n_points else 0.0}
        errors = np.array(errors) * 100.0
        errors_xy = np.array(errors_xy) * 100.0
        return {"mean": float(errors.mean()), "mean xy": float(errors_xy.mean()), "std": float(errors.std()),
                "< 3cm": float(small_error) / float(n_points), "25th percentile": float(np.percentile(errors, 25)),
                "75th percentile": float(np.percentile(errors, 75)), "missing": float(missing) / float(n_points) * 100.0,
                "points": n_points}

    def print_results(self):
        s = self.summary()
        print(" | ".join(self.COLUMNS))
        print(" | ".join(f"{s[c]:.4g}" if isinstance(s.get(c), float) else str(s.get(c, "-")) for c in self.COLUMNS))
