"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front for oracle/pnp_oracle.c
(restatement of cv2.solvePnPRansac(EPNP) + cv2.Rodrigues, reference recognition.py:216-223)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        dp = C.POINTER(C.c_double)
        _lib.p2po_solve_pnp_ransac.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp,
                                               C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
        _lib.p2po_solve_pnp_ransac.restype = C.c_int
        _lib.p2po_solve_pnp_epnp.argtypes = [dp, dp, dp, C.c_int, dp, dp]
        _lib.p2po_rng_sequence.argtypes = [C.c_uint64, C.c_int, C.POINTER(C.c_uint)]
        _lib.p2po_rodrigues_v2r.argtypes = [dp, dp]
        _lib.p2po_solve_pnp_ransac_batch.argtypes = [dp, dp, dp, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_double,
                                                     C.c_double, dp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def solve_pnp_ransac(obj_pts, img_pts, camK, iterations=100, reproj_err=5.0, confidence=0.99):
    """-> (ok, R[3,3], t[3], inlier_idx or None, info dict).  ok False == cv2 returning inliers=None."""
    obj = np.ascontiguousarray(obj_pts, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img_pts, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(camK, np.float64).reshape(9)
    n = obj.shape[0]
    R = np.eye(3)
    t = np.zeros(3)
    mask = np.zeros(max(n, 1), np.uint8)
    info = np.zeros(3, np.int32)
    ok = lib().p2po_solve_pnp_ransac(_d(K), _d(obj), _d(img), n, iterations, reproj_err, confidence, _d(R), _d(t),
                                     mask.ctypes.data_as(C.POINTER(C.c_ubyte)), info.ctypes.data_as(C.POINTER(C.c_int)))
    meta = {"n_inliers": int(info[0]), "iterations": int(info[1]), "best_iter": int(info[2])}
    if not ok:
        return False, np.eye(3), np.zeros(3), None, meta
    return True, R, t, np.nonzero(mask[:n])[0], meta


def solve_pnp_epnp(obj_pts, img_pts, camK):
    obj = np.ascontiguousarray(obj_pts, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img_pts, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(camK, np.float64).reshape(9)
    R = np.eye(3)
    t = np.zeros(3)
    lib().p2po_solve_pnp_epnp(_d(K), _d(obj), _d(img), obj.shape[0], _d(R), _d(t))
    return R, t


def rng_sequence(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint32)
    lib().p2po_rng_sequence(seed, n, out.ctypes.data_as(C.POINTER(C.c_uint)))
    return out


def rodrigues(rvec) -> np.ndarray:
    r = np.ascontiguousarray(rvec, np.float64).reshape(3)
    R = np.eye(3)
    lib().p2po_rodrigues_v2r(_d(r), _d(R))
    return R


def solve_pnp_ransac_batch(Ks, objs, imgs, iterations=100, reproj_err=5.0, confidence=0.99):
    """Independent problems on all host cores (OpenMP).  objs/imgs: lists of arrays."""
    n_prob = len(objs)
    offsets = np.zeros(n_prob + 1, np.int32)
    offsets[1:] = np.cumsum([len(o) for o in objs])
    obj = np.ascontiguousarray(np.concatenate(objs) if n_prob else np.zeros((0, 3)), np.float64)
    img = np.ascontiguousarray(np.concatenate(imgs) if n_prob else np.zeros((0, 2)), np.float64)
    K = np.ascontiguousarray(Ks, np.float64).reshape(n_prob, 9)
    R = np.zeros((n_prob, 9)); t = np.zeros((n_prob, 3))
    info = np.zeros((n_prob, 3), np.int32); ok = np.zeros(n_prob, np.int32)
    lib().p2po_solve_pnp_ransac_batch(_d(K), _d(obj), _d(img), offsets.ctypes.data_as(C.POINTER(C.c_int)), n_prob,
                                      iterations, reproj_err, confidence, _d(R), _d(t),
                                      info.ctypes.data_as(C.POINTER(C.c_int)), ok.ctypes.data_as(C.POINTER(C.c_int)))
    return ok.astype(bool), R.reshape(-1, 3, 3), t, info
