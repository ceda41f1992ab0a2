"""Rigid-transform helpers with the names and signatures of the reference's perception/utils/linalg.py:4-20 (host side, fp64)."""
import numpy as np


def skew_matrix(v):
    """[v]_x: skew_matrix(v) @ w == np.cross(v, w)."""
    v = np.asarray(v)
    m = np.zeros((3, 3), dtype=v.dtype)
    m[[2, 0, 1], [1, 2, 0]] = v           # (z,y)=x  (x,z)=y  (y,x)=z
    m[[1, 2, 0], [2, 0, 1]] = -v
    return m


def inv_transform(T):
    """Inverse of a rigid 4 x 4 transform: [R^T | -R^T t]."""
    inverse = np.eye(4, dtype=T.dtype)
    inverse[:3, :3] = T[:3, :3].T
    inverse[:3, 3] = -(inverse[:3, :3] @ T[:3, 3])
    return inverse


def transform_points(T, points):
    """T: 4 x 4; points: ... x 3 -> R p + t."""
    return np.einsum("ij,...j->...i", T[:3, :3], points) + T[:3, 3]
