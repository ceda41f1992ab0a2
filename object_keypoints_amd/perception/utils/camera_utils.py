"""Drop-in for the hot-path part of the reference's perception/utils/camera_utils.py (:7-110,119-189).

Same classes and method signatures (PinholeCamera, RadTanPinholeCamera, FisheyeCamera, StereoCamera, camera_matrix,
projection_matrix, load_calibration_params, scale_camera_matrix, fundamental_matrix); the OpenCV
calls of the reference (cv2.fisheye.undistortPoints, cv2.correctMatches, cv2.triangulatePoints) are
replaced by fp64 HIP kernels behind the C ABI (okp_camera_undistort, okp_triangulate_dlt).
NumPy arrays in, NumPy arrays out, as in the reference; there is no CPU implementation of the hot path here.
`FisheyeCamera.project` (reference :47-56: cv2.fisheye.projectPoints; used by the evaluation harness, the labelling tool
and the data sets, never per frame on the inference path) is plain host NumPy: the Kalibr equidistant model, which
reproduces the reference's known-answer projections (test/test_pipeline.py:26-33) to 5e-9 px.
"""
import numpy as np
import torch
import yaml

from ... import ops
from . import linalg


def _device():
    if not torch.cuda.is_available():
        raise ops.OkpError("camera geometry runs on the HIP device; no GPU is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _upper_triangular_inverse(K):
    """Inverse of a camera matrix [[fx, s, cx], [0, fy, cy], [0, 0, 1]] in closed form; anything else goes to LAPACK."""
    K = np.asarray(K, dtype=np.float64)
    if K.shape != (3, 3) or K[1, 0] != 0.0 or K[2, 0] != 0.0 or K[2, 1] != 0.0 or K[2, 2] != 1.0:
        return np.linalg.inv(K)
    fx, s, cx, fy, cy = K[0, 0], K[0, 1], K[0, 2], K[1, 1], K[1, 2]
    return np.array([[1.0 / fx, -s / (fx * fy), (s * cy - cx * fy) / (fx * fy)],
                     [0.0, 1.0 / fy, -cy / fy],
                     [0.0, 0.0, 1.0]])


class PinholeCamera:
    """Interface of the reference's PinholeCamera (camera_utils.py:7-43): attributes K, Kinv, D, image_size = (height, width)."""

    PRINCIPAL_POINT_TOLERANCE = 0.05      # the principal point sits within 5 % of the image width from the centre column

    def __init__(self, K, D, image_size):
        self.K, self.D = K, D
        self.Kinv = _upper_triangular_inverse(K)
        self.image_size = np.asarray(image_size).copy()     # (height, width)
        width = image_size[1]
        off_centre = abs(2.0 * K[0, 2] - width)
        assert off_centre < self.PRINCIPAL_POINT_TOLERANCE * width

    def scale(self, scale):
        """The camera of the same view resampled by `scale` (a FisheyeCamera whatever the receiver is, as in the reference)."""
        return FisheyeCamera(scale_camera_matrix(self.K, (scale, scale)), self.D, self.image_size * scale)

    def cut(self, offset):
        """The camera of the view cropped by offset = (x, y) pixels on every side."""
        shift = np.zeros_like(self.K)
        shift[:2, 2] = offset[:2]
        return FisheyeCamera(self.K - shift, self.D, self.image_size - 2.0 * np.asarray(offset)[::-1])

    def unproject(self, xys, zs):
        """(n, 2) pixels and (n,) depths -> (n, 3): z * K^-1 [x, y, 1]^T."""
        n = xys.shape[0]
        homogeneous = np.empty((n, 3), dtype=np.result_type(xys, np.float64))
        homogeneous[:, :2] = xys
        homogeneous[:, 2] = 1.0
        rays = np.einsum("ij,nj->ni", self.Kinv, homogeneous)
        return rays * np.asarray(zs)[:, None]

    def in_frame(self, x):
        """Row mask: no coordinate at or below 0, none at or beyond image_size."""
        outside = np.logical_or(x <= 0.0, x >= self.image_size)
        return np.logical_not(outside.any(axis=1))

    MODEL = 0          # okp_camera.model: 0 = equidistant (FisheyeCamera), 1 = radtan (RadTanPinholeCamera)

    def okp(self):
        return ops.make_camera(self.K, self.D, self.MODEL)

    def undistort(self, xy):
        """xy: N x 2 image points -> N x 2 undistorted points (device kernel), in the dtype given (as cv2 does)."""
        xy = np.asarray(xy)
        out = ops.camera_undistort(self.okp(), torch.from_numpy(np.ascontiguousarray(xy, dtype=np.float32)).to(_device()))
        out = out.cpu().numpy()
        return out.astype(xy.dtype) if xy.dtype in (np.float32, np.float64) else out


class RadTanPinholeCamera(PinholeCamera):
    """Kalibr "radtan" = OpenCV plumb-bob with D = (k1, k2, p1, p2) (reference camera_utils.py:45-62).  `undistort` is the device
    kernel (cv2.undistortPoints with P = K: five fixed-point iterations); `project` is host NumPy like FisheyeCamera.project.
    As in the reference, scale() / cut() return a FisheyeCamera (PinholeCamera's methods construct one, camera_utils.py:18-29)."""
    MODEL = 1

    def project(self, X, T_CW=np.eye(4)):
        X = linalg.transform_points(np.asarray(T_CW, dtype=np.float64), np.asarray(X, dtype=np.float64))
        x, y = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
        k1, k2, p1, p2 = [float(v) for v in np.asarray(self.D).reshape(-1)[:4]]
        r2 = x * x + y * y
        radial = 1 + k1 * r2 + k2 * r2 * r2
        xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        K = self.K
        return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], axis=1)


class FisheyeCamera(PinholeCamera):
    def project(self, X, T_CW=np.eye(4)):
        """(N,3) points in the frame that T_CW maps to the camera -> (N,2) pixels (host NumPy; not on the per-frame path)."""
        X = linalg.transform_points(np.asarray(T_CW, dtype=np.float64), np.asarray(X, dtype=np.float64))
        a, b = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
        r = np.sqrt(a * a + b * b)
        theta = np.arctan(r)
        t2 = theta * theta
        D = self.D
        theta_d = theta * (1 + D[0] * t2 + D[1] * t2 ** 2 + D[2] * t2 ** 3 + D[3] * t2 ** 4)
        scale = np.where(r > 1e-8, theta_d / np.where(r > 1e-8, r, 1.0), 1.0)
        x, y = a * scale, b * scale
        K = self.K
        return np.stack([K[0, 0] * x + K[0, 1] * y + K[0, 2], K[1, 1] * y + K[1, 2]], axis=1)


class StereoCamera:
    """A calibrated pair: T_RL maps left-camera coordinates to the right camera's (reference camera_utils.py:83-117)."""

    def __init__(self, left_camera, right_camera, T_RL):
        self.left_camera, self.right_camera = left_camera, right_camera
        self.T_RL, self.T_LR = T_RL, linalg.inv_transform(T_RL)
        self.F = fundamental_matrix(T_RL, left_camera.K, right_camera.K)

    def triangulate(self, left_keypoints, right_keypoints, correct_matches=True, per_camera_model=False):
        """Like the reference (camera_utils.py:92-110) the key points are undistorted with the FISHEYE (equidistant) model and the
        cameras' K / D whatever class the two cameras are - the reference calls cv2.fisheye.undistortPoints unconditionally.
        per_camera_model=True (not in the reference) undistorts each view with its own camera's model instead."""
        dev = _device()
        l = torch.from_numpy(np.ascontiguousarray(left_keypoints, dtype=np.float32)).to(dev)
        r = torch.from_numpy(np.ascontiguousarray(right_keypoints, dtype=np.float32)).to(dev)
        if per_camera_model:
            cam_l, cam_r = self.left_camera.okp(), self.right_camera.okp()
        else:
            cam_l = ops.make_camera(self.left_camera.K, self.left_camera.D)       # CAM_EQUIDISTANT
            cam_r = ops.make_camera(self.right_camera.K, self.right_camera.D)
        out = ops.triangulate_dlt(cam_l, cam_r, self.T_RL, l, r, F=self.F if correct_matches else None)
        return out.cpu().numpy()

    @classmethod
    def from_file(cls, calibration_file):
        """A Kalibr stereo calibration (cam0 = left, cam1 = right) as two FisheyeCameras."""
        p = load_calibration_params(calibration_file)
        cameras = [FisheyeCamera(p[k], p[d], p['image_size']) for k, d in (('K', 'D'), ('Kp', 'Dp'))]
        return cls(cameras[0], cameras[1], p['T_RL'])


def camera_matrix(intrinsics):
    """(fx, fy, cx, cy) -> 3 x 3."""
    K = np.eye(3)
    K[[0, 1], [0, 1]] = intrinsics[:2]
    K[:2, 2] = intrinsics[2:4]
    return K


def projection_matrix(camera_matrix, T_CW):
    return np.matmul(camera_matrix, np.asarray(T_CW)[:3])


def _read_yaml(path):
    with open(path, 'rt') as f:
        return yaml.load(f.read(), Loader=yaml.SafeLoader)


def _intrinsics_of(entry):
    """(K, D, (height, width)) of one Kalibr camera entry (resolution is stored width first)."""
    width, height = entry['resolution']
    return camera_matrix(entry['intrinsics']), np.array(entry['distortion_coeffs']), [height, width]


_CAMERA_CLASSES = {('pinhole', 'equidistant'): FisheyeCamera, ('pinhole', 'radtan'): RadTanPinholeCamera}


def from_calibration(calibration_file):
    entry = _read_yaml(calibration_file)['cam0']
    camera_class = _CAMERA_CLASSES.get((entry['camera_model'], entry['distortion_model']))
    if camera_class is None:
        raise ValueError(f"Unrecognized calibration type {entry['distortion_model']}.")
    return camera_class(*_intrinsics_of(entry))


def load_calibration_params(calibration_file):
    calibration = _read_yaml(calibration_file)
    (K, D, _), (Kp, Dp, size) = _intrinsics_of(calibration['cam0']), _intrinsics_of(calibration['cam1'])
    T_RL = np.array(calibration['cam1']['T_cn_cnm1'])
    return dict(K=K, Kp=Kp, D=D, Dp=Dp, T_RL=T_RL, T_LR=linalg.inv_transform(T_RL), image_size=size)


def scale_camera_matrix(K, scaling_factor):
    """Focal lengths and principal point scaled by (sx, sy); a skew entry is left as it is, as the reference does."""
    sx, sy = scaling_factor[0], scaling_factor[1]
    scaled = np.array(K, copy=True)
    scaled[0, [0, 2]] = K[0, [0, 2]] * sx
    scaled[1, [1, 2]] = K[1, [1, 2]] * sy
    return scaled


def fundamental_matrix(T_RL, K, Kp):
    """F with x_R^T F x_L = 0 for pixels of the same point: Kp^-T R K^T [K R^T t]_x (the reference's form, camera_utils.py:184-189)."""
    R, t = T_RL[:3, :3], T_RL[:3, 3]
    epipole = K @ (R.T @ t)
    return np.linalg.inv(Kp).T @ (R @ K.T) @ linalg.skew_matrix(epipole)
