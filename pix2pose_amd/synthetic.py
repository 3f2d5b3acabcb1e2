"""Synthetic detection scenes for tests, smoke() and bench.py.

No trained weights or datasets exist offline, and random weights give meaningless masks, so the
PnP stage is driven by *injected decoder outputs*: the normalised-object-coordinate (NOCS) image a
perfect Pix2Pose network would emit for an ellipsoid whose half-axes are the object's obj_scale,
ray-cast at a known pose, plus a per-pixel error map.  A share of the object pixels carries wrong
coordinates: half of those are flagged by a high predicted error (removed by the outlier
thresholds), half are not (left for RANSAC).  The generator passes still run on the real crops; the
pipeline then overwrites their output with these maps (p2p_est_pose_opts.inject1/2).
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


def render_nocs_at(R, t, K, scale, us, vs):
    """Ray-cast the ellipsoid sum((p_i/scale_i)^2)=1 under pose (R,t) at (sub-)pixel positions
    ``us``/``vs`` (same shape).  -> nocs [...,3] in [-1,1] (0 where missed), hit mask [...]."""
    d = np.stack([(us - K[0, 2]) / K[0, 0], (vs - K[1, 2]) / K[1, 1], np.ones_like(us, float)], -1)
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


def render_ellipsoid_nocs(R, t, K, scale, u0, v0, w, h):
    us, vs = np.meshgrid(np.arange(u0, u0 + w, dtype=float), np.arange(v0, v0 + h, dtype=float))
    return render_nocs_at(R, t, K, scale, us, vs)


def project(K, R, t, P):
    X = P @ R.T + t
    return np.stack([K[0, 0] * X[:, 0] / X[:, 2] + K[0, 2], K[1, 1] * X[:, 1] / X[:, 2] + K[1, 2]], -1)


def pose_error(R0, t0, R1, t1):
    """(translation error in mm, rotation error in degrees)."""
    dt = float(np.linalg.norm(np.asarray(t0) - np.asarray(t1)))
    c = (np.trace(np.asarray(R0).T @ np.asarray(R1)) - 1) / 2
    return dt, float(np.degrees(np.arccos(np.clip(c, -1, 1))))


def decoder_map(R, t, K, obj_param, v1_ori, u1_ori, side, rs, outlier_frac=0.2):
    """The [128,128,4] (x,y,z,prob) map a perfect network would emit for the square crop whose
    top-left is (v1_ori,u1_ori) and side ``side`` (sampled at the centres skimage's resize uses)."""
    g = (np.arange(128) + 0.5) * (side / 128.0) - 0.5
    us, vs = np.meshgrid(u1_ori + g, v1_ori + g)
    P0, hit = render_nocs_at(R, t, K, np.asarray(obj_param[:3], float), us, vs)
    nocs = P0 - np.asarray(obj_param[3:], float) / np.asarray(obj_param[:3], float)     # ct = 0 in practice
    out = np.zeros((128, 128, 4), np.float32)
    out[..., 3] = 0.9
    prob = np.where(hit, 0.05, 0.9)
    idx = np.flatnonzero(hit)
    if len(idx):
        # graded confidence so the outlier thresholds (e.g. 0.2/0.3/0.35) select different masks
        grade = rs.rand(len(idx))
        pr = np.where(grade < 0.06, 0.25, np.where(grade < 0.10, 0.32, 0.05))
        bad = rs.rand(len(idx)) < outlier_frac
        flagged = bad & (rs.rand(len(idx)) < 0.5)
        wrong = rs.uniform(-1, 1, (len(idx), 3))
        flat = nocs.reshape(-1, 3)
        flat[idx[bad]] = wrong[bad]
        pr = np.where(flagged, 0.9, pr)
        prob.reshape(-1)[idx] = pr
    out[..., :3] = np.where(hit[..., None], nocs, 0.0)
    out[..., 3] = prob
    return out


def stage2_box(map1, box1, bbox, H, W, box_size=1.5):
    """Stage-2 crop geometry implied by a stage-1 decoder map -- what the pipeline derives on the
    device (reference recognition.py:98-110), restated here with the shim's get_boxes."""
    from .recognition import get_boxes
    non_gray = np.linalg.norm(map1[..., :3].astype(np.float32), axis=2) > 0.3
    vs, us = np.where(non_gray)
    if len(vs) == 0:
        return None
    side_v, side_u = box1[1] - box1[0], box1[3] - box1[2]
    bb = np.array([vs.min(), us.min(), vs.max(), us.max()]) * np.array([side_v / 128, side_u / 128] * 2)
    cx_o, cy_o = (bbox[3] + bbox[1]) / 2, (bbox[2] + bbox[0]) / 2
    cx_m = int((np.mean(us) - (127 / 2)) + cx_o)
    cy_m = int((np.mean(vs) - (127 / 2)) + cy_o)
    return get_boxes(bb, H, W, box_size, ct=np.array([cy_m, cx_m]), max_w=side_v)


def make_scene(n_det, seed=0, n_slots=3, H=480, W=640, n_images=4, bbox_side=(86, 86), obj_param=OBJ_PARAM,
               K=LM_K, outlier_frac=0.2, z_range=(450.0, 900.0)):
    """-> dict(images [n_images,H,W,3] u8, dets [(img, obj, bbox, K)], gt [(R,t)], inject1 [n,128,128,4],
    inject2 [n,n_slots,128,128,4]).  bbox_side=(86,86) gives 128-px stage-1 crops (all resizes
    are the identity, BASELINE.json configs[1-3]); other sizes exercise the general resize path."""
    from .recognition import get_boxes
    rs = np.random.RandomState(seed)
    images = rs.randint(0, 256, (n_images, H, W, 3)).astype(np.uint8)
    dets, gt = [], []
    inj1 = np.zeros((n_det, 128, 128, 4), np.float32)
    inj2 = np.zeros((n_det, n_slots, 128, 128, 4), np.float32)
    inj2[..., 3] = 0.9
    scale = np.asarray(obj_param[:3], float)
    for i in range(n_det):
        side = int(rs.randint(bbox_side[0], bbox_side[1] + 1))
        crop = 2 * int(1.5 * side / 2)
        half = crop // 2 + 2
        cv_ = int(rs.randint(min(half, H // 2), max(H - half, H // 2 + 1)))
        cu_ = int(rs.randint(min(half, W // 2), max(W - half, W // 2 + 1)))
        bbox = [cv_ - side // 2, cu_ - side // 2, cv_ - side // 2 + side, cu_ - side // 2 + side]
        # depth chosen so the ellipsoid fills roughly the bbox
        z = float(rs.uniform(*z_range)) * (86.0 / side)
        t = np.array([(cu_ + rs.uniform(-4, 4) - K[0, 2]) * z / K[0, 0], (cv_ + rs.uniform(-4, 4) - K[1, 2]) * z / K[1, 1], z])
        R = random_rotation(rs)
        b1 = get_boxes(bbox, H, W, 1.5)
        inj1[i] = decoder_map(R, t, K, obj_param, b1[0], b1[2], b1[1] - b1[0], rs, outlier_frac)
        b2 = stage2_box(inj1[i], b1, bbox, H, W)
        if b2 is not None and b2[1] - b2[0] >= 5:
            for k in range(n_slots):
                inj2[i, k] = decoder_map(R, t, K, obj_param, b2[0], b2[2], b2[1] - b2[0], rs, outlier_frac)
        dets.append((int(rs.randint(0, n_images)), 0, bbox, K))
        gt.append((R, t))
    return {"images": images, "dets": dets, "gt": gt, "inject1": inj1, "inject2": inj2, "obj_param": np.asarray(obj_param, float)}
