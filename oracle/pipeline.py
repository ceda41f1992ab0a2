"""ORACLE (test infrastructure, not product code): NumPy restatement of the reference's
post-network pipeline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Reference being restated (perception/pipeline.py, file:line):
  KeypointExtractionComponent   :30-91   (5x5 ones box-sum, nms, >0.5 gate, per-peak centroid)
  ObjectExtraction              :93-153  (centre-vector voting, 20 px gate, arg-max / k-means de-dup)
  DetectionToPoint              :155-171 (undistort, round, clip, depth gather, unproject)
  ObjectKeypointPipeline        :173-200
  nms                           perception/models.py:55-58

Parity status: PINNED for extraction / object grouping / depth lifting by tests/golden/pipeline.npz,
produced by running the reference classes (tests/golden/make_goldens_pipeline.py).  The k-means
de-dup branch (pipeline.py:143-148) is nondeterministic in the reference (init='random', no seed)
and is checked at set level only.
"""
import numpy as np

from . import geometry


RGB_MEAN = np.array([0.40789654, 0.44719302, 0.47026115], dtype=np.float32)   # perception/datasets/video.py:55
RGB_STD = np.array([0.28863828, 0.27408164, 0.27809835], dtype=np.float32)    # perception/datasets/video.py:56


def normalize_frames(u8_nhwc):
    """uint8 RGB [N,H,W,3] -> float32 [N,3,H,W], exactly the expression of perception/datasets/video.py:215."""
    x = u8_nhwc.astype(np.float32).transpose([0, 3, 1, 2])
    return ((x / 255.0 - RGB_MEAN[None, :, None, None]) / RGB_STD[None, :, None, None]).astype(np.float32)


def box_sum5(p):
    """5x5 ones convolution with zero padding, accumulated in fp32 in row-major tap order — the
    association the reference's conv2d produces bit-for-bit."""
    p = np.asarray(p, dtype=np.float32)
    h, w = p.shape
    padded = np.zeros((h + 4, w + 4), dtype=np.float32)
    padded[2:-2, 2:-2] = p
    acc = np.zeros((h, w), dtype=np.float32)
    for dy in range(5):
        for dx in range(5):
            acc = (acc + padded[dy:dy + h, dx:dx + w]).astype(np.float32)
    return acc


def nms5(x, size=5):
    """x * (x == max over the size x size window), window clipped at the border."""
    r = size // 2
    h, w = x.shape
    padded = np.full((h + 2 * r, w + 2 * r), -np.inf, dtype=x.dtype)
    padded[r:r + h, r:r + w] = x
    m = np.full((h, w), -np.inf, dtype=x.dtype)
    for dy in range(size):
        for dx in range(size):
            m = np.maximum(m, padded[dy:dy + h, dx:dx + w])
    return x * (x == m).astype(x.dtype)


def peak_indices(p):
    """(P,2) int32 (y,x) of the peaks of one map in row-major order."""
    s = nms5(box_sum5(p))
    ys, xs = np.nonzero(s > np.float32(0.5))
    return np.stack([ys, xs], axis=1).astype(np.int32)


def refine_peaks(p, indices):
    """Probability-weighted centroid over the clipped 5x5 window and its mass (fp32)."""
    p = np.asarray(p, dtype=np.float32)
    h, w = p.shape
    points, conf = [], []
    for y, x in indices:
        y0, y1, x0, x1 = max(y - 2, 0), min(y + 3, h), max(x - 2, 0), min(x + 3, w)
        win = p[y0:y1, x0:x1]
        yy, xx = np.meshgrid(np.arange(y0, y1, dtype=np.float32), np.arange(x0, x1, dtype=np.float32), indexing="ij")
        mass = win.sum(dtype=np.float32)
        cy = (win * yy).sum(dtype=np.float32) / mass
        cx = (win * xx).sum(dtype=np.float32) / mass
        points.append(np.array([cx, cy], dtype=np.float32))      # (x, y)
        conf.append(np.float32(mass))
    return points, conf


class KeypointExtractionComponent:
    def __init__(self, keypoint_config, prediction_size, bandwidth=1.0):
        self.keypoint_config = [1] + keypoint_config["keypoint_config"]
        self.n_keypoints = sum(self.keypoint_config)
        self.prediction_size = prediction_size

    def _extract_keypoints(self, heatmap):
        assert heatmap.shape[0] == len(self.keypoint_config)
        out_points, confidences = [], []
        for i in range(len(self.keypoint_config)):
            p = heatmap[i].astype(np.float32)
            pts, conf = refine_peaks(p, peak_indices(p))
            out_points.append(pts)
            confidences.append(conf)
        return out_points, confidences

    def __call__(self, frames):
        keypoints, confidence = [], []
        for i in range(frames.shape[0]):
            kp, c = self._extract_keypoints(frames[i])
            keypoints.append(kp)
            confidence.append(c)
        return keypoints, confidence


def _lloyd(points, centers, iters=50):
    for _ in range(iters):
        assign = np.argmin(np.linalg.norm(points[:, None] - centers[None], axis=2), axis=1)
        new = np.stack([points[assign == j].mean(axis=0) if (assign == j).any() else centers[j] for j in range(len(centers))])
        if np.allclose(new, centers):
            break
        centers = new
    assign = np.argmin(np.linalg.norm(points[:, None] - centers[None], axis=2), axis=1)
    return centers, float(((points - centers[assign]) ** 2).sum())


def _kmeans(points, k, iters=50):
    """Deterministic stand-in for the reference's sklearn KMeans(init='random', n_clusters=k) (perception/pipeline.py:146), which under its
    pinned scikit-learn 0.24.1 runs n_init = 10 random starts and keeps the run of least inertia: Lloyd's algorithm from a farthest-point
    start seeded by EVERY point in turn, least inertia kept (earlier seeds win ties).  On the handful of votes an object receives this finds
    the optimum the reference's ten restarts find (tests/test_oracle_pipeline.py checks it against the exhaustive minimum)."""
    points = np.asarray(points)
    best = None
    for seed in range(len(points)):
        centers = [points[seed]]
        for _ in range(1, k):
            d = np.min([np.linalg.norm(points - c, axis=1) for c in centers], axis=0)
            centers.append(points[int(d.argmax())])
        centers, inertia = _lloyd(points.astype(np.float64), np.stack(centers).astype(np.float64), iters)
        if best is None or inertia < best[1] - 1e-12:
            best = (centers, inertia)
    return best[0]


class ObjectExtraction:
    def __init__(self, keypoint_config, prediction_size):
        self.keypoint_config = keypoint_config["keypoint_config"]
        self.prediction_size = prediction_size
        self.max = np.array(prediction_size[::-1], dtype=np.int32) - 1
        self.min = np.zeros(2, dtype=np.int32)
        h, w = prediction_size
        ys, xs = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
        self.image_indices = np.stack([xs, ys])          # (2,h,w): pixel centres (x, y)

    def __call__(self, keypoints, confidence, centers):
        if len(keypoints[0]) == 0:
            return []
        p_centers = self.image_indices + centers          # (K-1, 2, h, w) predicted object centre per pixel
        center_points = np.stack(keypoints[0])
        objects = [{"center": c, "heatmap_points": [[] for _ in keypoints[1:]],
                    "confidence": [[] for _ in keypoints[1:]], "p_centers": []} for c in center_points]
        for i, points in enumerate(keypoints[1:]):
            for j, point in enumerate(points):
                xy = np.clip(point.round().astype(np.int32), self.min, self.max)
                predicted = p_centers[i, :, xy[1], xy[0]]
                dist = np.linalg.norm(center_points - predicted[None], 2, axis=1)
                if dist.min() > 20.0:
                    continue                               # outlier: votes for no known centre
                obj = objects[int(dist.argmin())]
                obj["p_centers"].append(predicted)
                obj["heatmap_points"][i].append(point)
                obj["confidence"][i].append(confidence[i + 1][j])
        for obj in objects:
            for i in range(len(obj["heatmap_points"])):
                if len(obj["heatmap_points"][i]) == 0:
                    obj["heatmap_points"][i] = np.array([])
                    continue
                pts = np.stack(obj["heatmap_points"][i])
                conf = np.stack(obj["confidence"][i])
                want = self.keypoint_config[i]
                if pts.shape[0] > want:
                    pts = pts[conf.argmax(axis=0)][None] if want == 1 else _kmeans(pts, want)
                obj["heatmap_points"][i] = pts
        return objects


class DetectionToPoint:
    def reset(self, camera):
        self.camera = camera
        self.min_index = np.zeros(2, dtype=int)
        self.max_index = camera.image_size.astype(int) - 1

    def __call__(self, xy, p_depth):
        if xy.shape[0] == 0:
            return None
        xy = self.camera.undistort(xy)
        xy_int = np.clip(xy.round().astype(int), self.min_index, self.max_index)
        zs = p_depth[xy_int[:, 1], xy_int[:, 0]]
        return self.camera.unproject(xy, zs)


class ObjectKeypointPipeline:
    def __init__(self, prediction_size, points_3d, keypoint_config):
        self.keypoint_extraction = KeypointExtractionComponent(keypoint_config, prediction_size)
        self.object_extraction = ObjectExtraction(keypoint_config, prediction_size)
        self.detection_to_point = DetectionToPoint()

    def reset(self, camera):
        self.detection_to_point.reset(camera)

    def __call__(self, heatmap, p_depth, p_centers):
        heatmap, p_depth, p_centers = (np.asarray(t) for t in (heatmap, p_depth, p_centers))
        assert heatmap.shape[0] == 1, "One at the time, please."
        points, confidence = self.keypoint_extraction(heatmap)
        detected = self.object_extraction(points[0], confidence[0], p_centers[0])
        objects = []
        for obj in detected:
            world = [self.detection_to_point(obj["center"][None], p_depth[0][0])]
            for i, pts in enumerate(obj["heatmap_points"]):
                world.append(self.detection_to_point(pts, p_depth[0][1 + i]))
            objects.append({"p_centers": obj["p_centers"], "keypoints": [obj["center"][None]] + obj["heatmap_points"],
                            "p_C": world})
        return objects


class TriangulationComponent:
    """API the reference's test expects (test/test_pipeline.py:171-177): reset(stereo); __call__(p_L, p_R) -> (n,3)."""

    def reset(self, stereo_camera):
        self.stereo_camera = stereo_camera

    def __call__(self, left_keypoints, right_keypoints):
        return self.stereo_camera.triangulate(left_keypoints, right_keypoints)


def epipolar_cost(stereo_camera, left_points, right_points):
    """[n, m] symmetric point-to-epipolar-line distance (pixels, undistorted image space) between every left and right point."""
    L, R = stereo_camera.left_camera, stereo_camera.right_camera
    ul = geometry.fisheye_undistort(np.asarray(left_points, dtype=np.float64), L.K, L.D, P=L.K)
    ur = geometry.fisheye_undistort(np.asarray(right_points, dtype=np.float64), R.K, R.D, P=R.K)
    hl = np.concatenate([ul, np.ones((ul.shape[0], 1))], axis=1)
    hr = np.concatenate([ur, np.ones((ur.shape[0], 1))], axis=1)
    F = stereo_camera.F
    lines_r = hl @ F.T                      # F x_l: the epipolar line of each left point in the right image
    lines_l = hr @ F                        # F^T x_r
    num = np.abs(hl @ F.T @ hr.T)           # |x_r^T F x_l|
    d_r = num / np.maximum(np.linalg.norm(lines_r[:, :2], axis=1), 1e-300)[:, None]
    d_l = num / np.maximum(np.linalg.norm(lines_l[:, :2], axis=1), 1e-300)[None, :]
    return 0.5 * (d_r + d_l)


class AssociationComponent:
    """Left/right keypoint matching before triangulation.  There is NO implementation in the reference tree; the
    contract is its test (test/test_pipeline.py:208-261): reset(stereo_camera); __call__(points_left (n,2),
    points_right (m,2)) -> int array (n,), associations[i] = index of the right point matched to left point i or -1.
    Built here as: minimum-cost one-to-one assignment (Hungarian) on the symmetric epipolar distance, pairs
    farther than `max_distance` pixels from their epipolar lines being inadmissible."""

    def __init__(self, max_distance=20.0):
        self.max_distance = float(max_distance)

    def reset(self, stereo_camera):
        self.stereo_camera = stereo_camera

    def __call__(self, points_left, points_right):
        from scipy.optimize import linear_sum_assignment
        points_left, points_right = np.asarray(points_left), np.asarray(points_right)
        out = np.full(points_left.shape[0], -1, dtype=np.int64)
        if points_left.shape[0] == 0 or points_right.shape[0] == 0:
            return out
        cost = epipolar_cost(self.stereo_camera, points_left, points_right)
        # gate first: a point without any admissible partner must not steal one (all its pairs cost the same)
        rows, cols = linear_sum_assignment(np.where(cost <= self.max_distance, cost, 1e6))
        keep = cost[rows, cols] <= self.max_distance
        out[rows[keep]] = cols[keep]
        return out


def eval_camera(calibration_file, prediction_size=64, height=720, width=1280, resized=511):
    """The 64x64-space camera of scripts/eval_model.py:61-69 (scale to 511, centre-crop, scale to 64)."""
    params = geometry.load_calibration_params(calibration_file)
    camera = geometry.FisheyeCamera(params["K"], params["D"], params["image_size"])
    camera = camera.scale(resized / height)
    offset = np.array([(resized / height * width - float(resized)) / 2.0, 0.0])
    return camera.cut(offset).scale(prediction_size / resized)
