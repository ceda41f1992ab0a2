"""ORACLE (test infrastructure, not product code): NumPy fp64 restatement of the reference's
camera geometry.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Reference being restated (file:line under the reference tree):
  PinholeCamera / FisheyeCamera      perception/utils/camera_utils.py:7-81
  StereoCamera.triangulate           perception/utils/camera_utils.py:84-110
  fundamental_matrix                 perception/utils/camera_utils.py:184-189
  camera_matrix / load_calibration   perception/utils/camera_utils.py:119-170
  scale_camera_matrix                perception/utils/camera_utils.py:172-182
  skew_matrix / inv_transform / transform_points   perception/utils/linalg.py:4-20

The arithmetic of project / undistort / correctMatches / triangulatePoints lives in OpenCV, which
is NOT vendored in the reference tree and is not installed here (the reference's env pins
opencv=3.4.2, corner_net_lite/conda_packagelist.txt:53).  Those four are restated from OpenCV's
published algorithms:
  cv2.fisheye.projectPoints    theta = atan(r); theta_d = theta(1 + k1 t^2 + k2 t^4 + k3 t^6 + k4 t^8)
  cv2.fisheye.undistortPoints  3.4.2: clip theta_d to +-pi/2, 10 fixed-point iterations, scale = tan(theta)/theta_d
  cv2.triangulatePoints        per-point 4x4 homogeneous DLT, SVD null vector
  cv2.correctMatches           Hartley-Sturm (Hartley & Zisserman, Multiple View Geometry, alg. 12.1)
Parity status: project / undistort / DLT are PINNED by the known-answer vectors of the reference's
own test (test/test_pipeline.py:9-33,171-177; tests/golden/known_answers.json).  correct_matches
on noisy correspondences is parity-UNPINNED (no reference vector exercises it); it is checked by
property tests (epipolar constraint satisfied, displacement minimal).
"""
import numpy as np
import yaml


# ---- linalg -----------------------------------------------------------------------------------

def skew_matrix(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]], dtype=np.asarray(v).dtype)


def inv_transform(T):
    out = np.eye(4, dtype=T.dtype)
    R = T[:3, :3]
    out[:3, :3] = R.T
    out[:3, 3] = -R.T @ T[:3, 3]
    return out


def transform_points(T, points):
    return np.einsum("ij,...j->...i", T[:3, :3], points) + T[:3, 3]


# ---- OpenCV restatements ------------------------------------------------------------------------

def fisheye_project(X, K, D, T_CW=None):
    """Points (N,3) in the frame that T_CW maps to the camera -> (N,2) pixels (equidistant model)."""
    X = np.asarray(X, dtype=np.float64)
    if T_CW is not None:
        X = transform_points(np.asarray(T_CW, dtype=np.float64), X)
    a, b = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
    r = np.sqrt(a * a + b * b)
    theta = np.arctan(r)
    t2 = theta * theta
    theta_d = theta * (1 + D[0] * t2 + D[1] * t2 ** 2 + D[2] * t2 ** 3 + D[3] * t2 ** 4)
    scale = np.where(r > 1e-8, theta_d / np.where(r > 1e-8, r, 1.0), 1.0)
    x, y = a * scale, b * scale
    return np.stack([K[0, 0] * x + K[0, 1] * y + K[0, 2], K[1, 1] * y + K[1, 2]], axis=1)


def fisheye_undistort(xy, K, D, P=None):
    """cv2.fisheye.undistortPoints(xy, K, D, P=P) of OpenCV 3.4.2; the result is in the dtype of `xy`
    (cv2 returns the type it was given)."""
    xy_in = np.asarray(xy)
    out_dtype = xy_in.dtype if xy_in.dtype in (np.float32, np.float64) else np.float64
    pts = xy_in.astype(np.float64)
    P = K if P is None else P
    pw = np.stack([(pts[:, 0] - K[0, 2]) / K[0, 0], (pts[:, 1] - K[1, 2]) / K[1, 1]], axis=1)
    theta_d = np.clip(np.sqrt((pw * pw).sum(axis=1)), -np.pi / 2, np.pi / 2)
    theta = theta_d.copy()
    for _ in range(10):
        t2 = theta * theta
        theta = theta_d / (1 + D[0] * t2 + D[1] * t2 ** 2 + D[2] * t2 ** 3 + D[3] * t2 ** 4)
    ok = theta_d > 1e-8
    scale = np.where(ok, np.tan(theta) / np.where(ok, theta_d, 1.0), 1.0)
    pu = pw * scale[:, None]
    out = np.stack([P[0, 0] * pu[:, 0] + P[0, 1] * pu[:, 1] + P[0, 2], P[1, 1] * pu[:, 1] + P[1, 2]], axis=1)
    return out.astype(out_dtype)


def triangulate_points(P1, P2, x1, x2):
    """cv2.triangulatePoints: x1, x2 are (N,2); returns homogeneous (N,4)."""
    out = np.zeros((x1.shape[0], 4))
    for i in range(x1.shape[0]):
        A = np.stack([x1[i, 0] * P1[2] - P1[0], x1[i, 1] * P1[2] - P1[1],
                      x2[i, 0] * P2[2] - P2[0], x2[i, 1] * P2[2] - P2[1]])
        out[i] = np.linalg.svd(A)[2][-1]
    return out


def correct_matches(F, x1, x2):
    """cv2.correctMatches: move (x1[i], x2[i]) the least (sum of squared pixel distances) so that
    x2^T F x1 = 0 holds exactly.  Hartley-Sturm, HZ alg. 12.1."""
    F = np.asarray(F, dtype=np.float64)
    o1, o2 = np.array(x1, dtype=np.float64), np.array(x2, dtype=np.float64)
    for i in range(o1.shape[0]):
        T1i = np.array([[1, 0, o1[i, 0]], [0, 1, o1[i, 1]], [0, 0, 1.0]])
        T2i = np.array([[1, 0, o2[i, 0]], [0, 1, o2[i, 1]], [0, 0, 1.0]])
        Fp = T2i.T @ F @ T1i
        e1 = np.linalg.svd(Fp)[2][-1]          # F e1 = 0
        e2 = np.linalg.svd(Fp.T)[2][-1]        # e2^T F = 0
        e1 = e1 / np.hypot(e1[0], e1[1])
        e2 = e2 / np.hypot(e2[0], e2[1])
        R1 = np.array([[e1[0], e1[1], 0], [-e1[1], e1[0], 0], [0, 0, 1.0]])
        R2 = np.array([[e2[0], e2[1], 0], [-e2[1], e2[0], 0], [0, 0, 1.0]])
        Fpp = R2 @ Fp @ R1.T
        # plain Python floats: a NumPy scalar times a poly1d silently degrades to array arithmetic
        f1, f2 = float(e1[2]), float(e2[2])
        a, b, c, d = float(Fpp[1, 1]), float(Fpp[1, 2]), float(Fpp[2, 1]), float(Fpp[2, 2])
        t = np.poly1d([1.0, 0.0])
        u, v = a * t + b, c * t + d
        g = t * (u * u + f2 * f2 * v * v) ** 2 - (a * d - b * c) * (1 + f1 * f1 * t * t) ** 2 * u * v
        coeffs = np.array(g.coeffs, dtype=np.float64)
        keep = np.nonzero(np.abs(coeffs) > 1e-14 * np.abs(coeffs).max())[0]
        cands = [r.real for r in np.roots(coeffs[keep[0]:])]     # near-rectified rigs: f ~ 0 kills the top coefficients

        def cost(tt):
            uu, vv = a * tt + b, c * tt + d
            return tt * tt / (1 + f1 * f1 * tt * tt) + vv * vv / (uu * uu + f2 * f2 * vv * vv)

        best_t, best_s = None, np.inf
        for tt in cands:
            s = cost(tt)
            if s < best_s:
                best_t, best_s = tt, s
        den = a * a + f2 * f2 * c * c
        s_inf = 1.0 / (f1 * f1) + c * c / den if (f1 != 0 and den != 0) else np.inf
        if s_inf < best_s:
            l1 = np.array([f1, 0.0, -1.0])
            l2 = np.array([-f2 * c, a, c])
        else:
            l1 = np.array([best_t * f1, 1.0, -best_t])
            l2 = np.array([-f2 * (c * best_t + d), a * best_t + b, c * best_t + d])
        p1 = np.array([-l1[0] * l1[2], -l1[1] * l1[2], l1[0] ** 2 + l1[1] ** 2])
        p2 = np.array([-l2[0] * l2[2], -l2[1] * l2[2], l2[0] ** 2 + l2[1] ** 2])
        q1 = T1i @ R1.T @ p1
        q2 = T2i @ R2.T @ p2
        o1[i] = q1[:2] / q1[2]
        o2[i] = q2[:2] / q2[2]
    return o1, o2


# ---- cameras ---------------------------------------------------------------------------------------

def camera_matrix(intrinsics):
    fx, fy, cx, cy = intrinsics
    return np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])


def scale_camera_matrix(K, scaling_factor):
    out = K.copy()
    out[0, 0] *= scaling_factor[0]
    out[0, 2] *= scaling_factor[0]
    out[1, 1] *= scaling_factor[1]
    out[1, 2] *= scaling_factor[1]
    return out


def projection_matrix(K, T_CW):
    return K @ T_CW[:3, :]


def fundamental_matrix(T_RL, K, Kp):
    R, t = T_RL[:3, :3], T_RL[:3, 3]
    return np.linalg.inv(Kp).T @ R @ K.T @ skew_matrix(K @ R.T @ t)


class PinholeCamera:
    def __init__(self, K, D, image_size):
        self.K = K
        self.Kinv = np.linalg.inv(K)
        self.D = D
        self.image_size = np.array(image_size)        # (height, width)
        assert np.abs(K[0, 2] * 2.0 - image_size[1]) < 0.05 * image_size[1]

    def scale(self, scale):
        return FisheyeCamera(scale_camera_matrix(self.K, np.ones(2) * scale), self.D, self.image_size * scale)

    def cut(self, offset):
        K = self.K.copy()
        K[0, 2] -= offset[0]
        K[1, 2] -= offset[1]
        return FisheyeCamera(K, self.D, self.image_size - 2.0 * offset[::-1])

    def unproject(self, xys, zs):
        xs = np.concatenate([xys, np.ones((xys.shape[0], 1))], axis=1)
        return (self.Kinv @ xs[:, :, None])[:, :, 0] * zs[:, None]

    def in_frame(self, x):
        return ~((x <= 0.0).any(axis=1) | (x >= self.image_size).any(axis=1))


def radtan_project(X, K, D, T_CW=None):
    """cv2.projectPoints for the plumb-bob model with D = (k1, k2, p1, p2) (reference camera_utils.py:46-55)."""
    X = np.asarray(X, dtype=np.float64)
    if T_CW is not None:
        X = transform_points(np.asarray(T_CW, dtype=np.float64), X)
    x, y = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
    k1, k2, p1, p2 = [float(v) for v in np.asarray(D).reshape(-1)[:4]]
    r2 = x * x + y * y
    radial = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], axis=1)


def radtan_undistort(xy, K, D, P=None, iterations=5):
    """cv2.undistortPoints(xy, K, D, P=P) of OpenCV 3.4.2 (reference camera_utils.py:57-62): `iterations` fixed-point steps of
    x <- (x0 - tangential(x, y)) / radial(x, y); the result is in the dtype of `xy`.  Parity-unpinned against OpenCV itself
    (cv2 is not importable here and the reference's tests hold no radtan vector): pinned by project(undistort) properties."""
    xy_in = np.asarray(xy)
    out_dtype = xy_in.dtype if xy_in.dtype in (np.float32, np.float64) else np.float64
    pts = xy_in.astype(np.float64)
    P = K if P is None else P
    k1, k2, p1, p2 = [float(v) for v in np.asarray(D).reshape(-1)[:4]]
    x0, y0 = (pts[:, 0] - K[0, 2]) / K[0, 0], (pts[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iterations):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + (k2 * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    return np.stack([P[0, 0] * x + P[0, 2], P[1, 1] * y + P[1, 2]], axis=1).astype(out_dtype)


class RadTanPinholeCamera(PinholeCamera):
    def project(self, X, T_CW=np.eye(4)):
        return radtan_project(X, self.K, self.D, T_CW)

    def undistort(self, xy):
        return radtan_undistort(xy, self.K, self.D, P=self.K)


class FisheyeCamera(PinholeCamera):
    def project(self, X, T_CW=np.eye(4)):
        return fisheye_project(X, self.K, self.D, T_CW)

    def undistort(self, xy):
        return fisheye_undistort(xy, self.K, self.D, P=self.K)


class StereoCamera:
    def __init__(self, left_camera, right_camera, T_RL):
        self.left_camera, self.right_camera = left_camera, right_camera
        self.T_RL = T_RL
        self.T_LR = inv_transform(T_RL)
        self.F = fundamental_matrix(T_RL, left_camera.K, right_camera.K)

    def triangulate(self, left_keypoints, right_keypoints, correct=True):
        l = fisheye_undistort(left_keypoints.astype(np.float32), self.left_camera.K, self.left_camera.D, P=self.left_camera.K)
        r = fisheye_undistort(right_keypoints.astype(np.float32), self.right_camera.K, self.right_camera.D, P=self.right_camera.K)
        if correct:
            l, r = correct_matches(self.F, l, r)
            l, r = l.astype(np.float32), r.astype(np.float32)      # cv2 returns the input dtype
        P1 = self.left_camera.K @ np.eye(3, 4)
        P2 = self.right_camera.K @ self.T_RL[:3]
        p = triangulate_points(P1, P2, l.astype(np.float64), r.astype(np.float64))
        return p[:, :3] / p[:, 3:4]

    @classmethod
    def from_file(cls, calibration_file):
        c = load_calibration_params(calibration_file)
        return cls(FisheyeCamera(c["K"], c["D"], c["image_size"]), FisheyeCamera(c["Kp"], c["Dp"], c["image_size"]), c["T_RL"])


def load_calibration_params(calibration_file):
    with open(calibration_file, "rt") as f:
        calibration = yaml.safe_load(f.read())
    left, right = calibration["cam0"], calibration["cam1"]
    T_RL = np.array(right["T_cn_cnm1"])
    return {"K": camera_matrix(left["intrinsics"]), "Kp": camera_matrix(right["intrinsics"]),
            "D": np.array(left["distortion_coeffs"]), "Dp": np.array(right["distortion_coeffs"]),
            "T_LR": inv_transform(T_RL), "T_RL": T_RL, "image_size": right["resolution"][::-1]}
