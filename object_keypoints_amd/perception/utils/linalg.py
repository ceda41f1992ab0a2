"""Mirror of the reference's perception/utils/linalg.py:4-20 (small host-side rigid-transform helpers)."""
import numpy as np


def skew_matrix(v):
    x, y, z = v
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]], dtype=np.asarray(v).dtype)


def inv_transform(T):
    Rt = T[:3, :3].T
    out = np.eye(4, dtype=T.dtype)
    out[:3, :3] = Rt
    out[:3, 3] = -Rt @ T[:3, 3]
    return out


def transform_points(T, points):
    """T: 4x4; points: ... x 3."""
    return (T[:3, :3] @ points[..., None])[..., 0] + T[:3, 3]
