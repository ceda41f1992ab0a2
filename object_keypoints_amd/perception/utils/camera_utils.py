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


class PinholeCamera:
    def __init__(self, K, D, image_size):
        self.K = K
        self.Kinv = np.linalg.inv(K)
        self.D = D
        self.image_size = np.array(image_size)     # height, width
        assert np.abs(K[0, 2] * 2.0 - image_size[1]) < 0.05 * image_size[1]

    def scale(self, scale):
        K = scale_camera_matrix(self.K, np.ones(2) * scale)
        return FisheyeCamera(K, self.D, self.image_size * scale)

    def cut(self, offset):
        K = self.K.copy()
        K[0, 2] = self.K[0, 2] - offset[0]
        K[1, 2] = self.K[1, 2] - offset[1]
        return FisheyeCamera(K, self.D, self.image_size - 2.0 * offset[::-1])

    def unproject(self, xys, zs):
        xs = np.concatenate([xys, np.ones((xys.shape[0], 1))], axis=1)
        return (self.Kinv @ xs[:, :, None])[:, :, 0] * zs[:, None]

    def in_frame(self, x):
        under = (x <= 0.0).any(axis=1)
        over = (x >= self.image_size).any(axis=1)
        return np.bitwise_or(under, over) == False  # noqa: E712

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
    def __init__(self, left_camera, right_camera, T_RL):
        self.left_camera = left_camera
        self.right_camera = right_camera
        self.T_RL = T_RL
        self.T_LR = linalg.inv_transform(T_RL)
        self.F = fundamental_matrix(T_RL, self.left_camera.K, self.right_camera.K)

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
        camera = load_calibration_params(calibration_file)
        left_camera = FisheyeCamera(camera['K'], camera['D'], camera['image_size'])
        right_camera = FisheyeCamera(camera['Kp'], camera['Dp'], camera['image_size'])
        return cls(left_camera, right_camera, camera['T_RL'])


def camera_matrix(intrinsics):
    fx, fy, cx, cy = intrinsics
    return np.array([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]])


def projection_matrix(camera_matrix, T_CW):
    return camera_matrix @ T_CW[:3, :]


def from_calibration(calibration_file):
    with open(calibration_file, 'rt') as f:
        camera = yaml.load(f.read(), Loader=yaml.SafeLoader)['cam0']
    K = camera_matrix(camera['intrinsics'])
    D = np.array(camera['distortion_coeffs'])
    if camera['distortion_model'] == 'equidistant' and camera['camera_model'] == 'pinhole':
        return FisheyeCamera(K, D, camera['resolution'][::-1])
    if camera['distortion_model'] == 'radtan' and camera['camera_model'] == 'pinhole':
        return RadTanPinholeCamera(K, D, camera['resolution'][::-1])
    raise ValueError(f"Unrecognized calibration type {camera['distortion_model']}.")


def load_calibration_params(calibration_file):
    with open(calibration_file, 'rt') as f:
        calibration = yaml.load(f.read(), Loader=yaml.SafeLoader)
    left, right = calibration['cam0'], calibration['cam1']
    T_RL = np.array(right['T_cn_cnm1'])
    return {
        'K': camera_matrix(left['intrinsics']), 'Kp': camera_matrix(right['intrinsics']),
        'D': np.array(left['distortion_coeffs']), 'Dp': np.array(right['distortion_coeffs']),
        'T_LR': linalg.inv_transform(T_RL), 'T_RL': T_RL, 'image_size': right['resolution'][::-1],
    }


def scale_camera_matrix(K, scaling_factor):
    out = K.copy()
    out[0, 0] = K[0, 0] * scaling_factor[0]
    out[1, 1] = K[1, 1] * scaling_factor[1]
    out[0, 2] = K[0, 2] * scaling_factor[0]
    out[1, 2] = K[1, 2] * scaling_factor[1]
    return out


def fundamental_matrix(T_RL, K, Kp):
    R = T_RL[:3, :3]
    t = T_RL[:3, 3]
    C = linalg.skew_matrix(K @ R.T @ t)
    return np.linalg.inv(Kp).T @ R @ K.T @ C
