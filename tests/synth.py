"""Synthetic scene generators shared by tests, smoke() and bench.py.

No reference data exists offline, so PnP is exercised on rendered NOCS maps of an ellipsoid whose
half-axes are the object's ``obj_scale``: for a pixel ray hitting the ellipsoid, the normalised
object coordinate (x/sx, y/sy, z/sz) is what a perfect Pix2Pose network would output there.
"""
from __future__ import annotations

import numpy as np

# LINEMOD intrinsics hard-coded in the reference at rendering/gpu_render.py:15
LM_K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], np.float64)
# approximate public LM obj_01 half extents (mm) -- NOT in the reference; synthetic stand-in
OBJ_PARAM = np.array([37.9, 38.8, 45.9, 0.0, 0.0, 0.0], np.float64)


def random_rotation(rs: np.random.RandomState) -> np.ndarray:
    q = rs.randn(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def render_ellipsoid_nocs(R, t, K, scale, u0, v0, w, h):
    """Ray-cast the ellipsoid sum((p_i/scale_i)^2)=1 under pose (R,t) over the pixel window
    [v0,v0+h) x [u0,u0+w).  Returns nocs [h,w,3] in [-1,1] (0 where missed) and hit mask [h,w]."""
    us, vs = np.meshgrid(np.arange(u0, u0 + w), np.arange(v0, v0 + h))
    d = np.stack([(us - K[0, 2]) / K[0, 0], (vs - K[1, 2]) / K[1, 1], np.ones_like(us, float)], -1)   # cam rays
    # object frame: p = R^T (s d - t);  unit-sphere coords q = p / scale
    Rt = R.T
    o = -(Rt @ t) / scale
    dd = (d @ Rt.T) / scale
    a = (dd * dd).sum(-1)
    b = 2 * (dd * o).sum(-1)
    c = (o * o).sum() - 1.0
    disc = b * b - 4 * a * c
    hit = disc > 0
    s = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0))) / (2 * a), 0)
    nocs = np.where(hit[..., None], o + s[..., None] * dd, 0.0)
    return nocs, hit


def project(K, R, t, P):
    X = P @ R.T + t
    return np.stack([K[0, 0] * X[:, 0] / X[:, 2] + K[0, 2], K[1, 1] * X[:, 1] / X[:, 2] + K[1, 2]], -1)


def pose_error(R0, t0, R1, t1):
    """(translation error in mm, rotation error in degrees)."""
    dt = float(np.linalg.norm(np.asarray(t0) - np.asarray(t1)))
    c = (np.trace(np.asarray(R0).T @ np.asarray(R1)) - 1) / 2
    return dt, float(np.degrees(np.arccos(np.clip(c, -1, 1))))
