"""Drop-in for the reference's ``pix2pose_model/recognition.py``: same class name, constructor
signature, attributes and ``est_pose`` / ``get_boxes`` / ``pnp_ransac`` /
``generator_train.predict`` surface (reference recognition.py:10-224), backed by the MI355X HIP
pipeline through the C ABI (include/p2p_mi355.h).  There is no CPU fallback.

Differences a caller can observe (all documented in INTEGRATION.md):
  * ``weight_fn`` names a ``.npz`` artefact (pix2pose_amd.weights) or ``synthetic:<backbone>:<seed>``,
    not a Keras HDF5 file (h5py/Keras are not part of this stack; the converter is tools-side).
  * an unknown ``backbone`` raises ValueError (the reference silently leaves ``generator_train``
    undefined, recognition.py:21-26).
  * on failure the first tuple element is what the reference returns there (np.zeros(1) at :79, the stage-1 / last-candidate preview at
    :127 / :191) -- fetched by a second debug-tap call on that path only (no caller reads it: tools/5_evaluation_bop_basic.py:303-305
    tests ``frac_inlier == -1`` and continues).
  * boxes entirely outside the frame (where the reference crashes on a shape mismatch) fail cleanly.
"""
from __future__ import annotations

import numpy as np

from . import runtime
from . import weights as W


def get_boxes(bbox, v_max, u_max, box_size=1.5, ct=np.array([-1]), max_w=9999):
    """Square crop geometry with Python ``int()`` truncation (reference recognition.py:28-69).
    Returns the same 12-tuple: v1_ori,v2_ori,u1_ori,u2_ori,v1,v2,u1,u2,vv1,vv2,uu1,uu2."""
    if ct[0] == -1:
        ct_v, ct_u = int((bbox[0] + bbox[2]) / 2), int((bbox[1] + bbox[3]) / 2)
    else:
        ct_v, ct_u = ct[0], ct[1]
    width, height = bbox[3] - bbox[1], bbox[2] - bbox[0]
    half = int(min(max_w, max(width * box_size, height * box_size)) / 2)
    v1o, v2o, u1o, u2o = ct_v - half, ct_v + half, ct_u - half, ct_u + half
    v1, v2, u1, u2 = v1o, v2o, u1o, u2o
    s = [0, 0, 0, 0]
    if v1o < 0:
        s[0], v1 = abs(v1o), 0
    if v2o > v_max:
        s[1], v2 = -abs(v2o - v_max), v_max
    if u1o < 0:
        s[2], u1 = abs(u1o), 0
    if u2o > u_max:
        s[3], u2 = -abs(u2o - u_max), u_max
    return tuple(int(x) for x in (v1o, v2o, u1o, u2o, v1, v2, u1, u2, s[0], s[1] + (v2o - v1o), s[2], s[3] + (u2o - u1o)))


class pix2pose():
    """One instance per object, like the reference (tools/5_evaluation_bop_basic.py:206-225)."""

    def __init__(self, weight_fn, camK, res_x, res_y, obj_param, th_ransac=3.0, th_outlier=[0.1, 0.2, 0.3],
                 th_inlier=0.1, box_size=1.5, dist_coeff=None, backbone="paper", **kwargs):
        self.camK = camK                       # callers re-assign it per image (5_evaluation_bop_basic.py:302)
        self.res_x = res_x
        self.res_y = res_y
        self.th_ransac = th_ransac             # stored, never used -- as in the reference (:14, :216-217)
        self.th_o = th_outlier
        self.th_i = th_inlier
        self.obj_scale = obj_param[:3]
        self.obj_ct = obj_param[3:]
        self.box_size = box_size
        self.dist_coeff = dist_coeff
        if backbone not in W.BACKBONES:
            raise ValueError("backbone must be 'paper' or 'resnet50', got %r" % (backbone,))
        self.backbone = backbone
        ctx = kwargs.get("ctx") or runtime.default_context(kwargs.get("device", 0))
        weights = weight_fn if isinstance(weight_fn, dict) else W.load_weights(weight_fn, backbone)
        self.ctx = ctx
        # skimage.transform.resize semantics: False = scikit-image <= 0.14 (no anti-aliasing), True = 0.17 - 0.18 (Gaussian
        # pre-filter when down-scaling); the reference does not pin the version
        # skimage="0.14" / "0.18" names the generation outright ("0.18" is pinned bit for bit to the real scikit-image 0.18.3)
        # "0.15" / "0.16": anti-aliasing also on the bool keep mask of recognition.py:103 (what the reference's own python-3.5 image resolves to).
        # Nothing named: generation 0 with a one-time warning that the choice was made for the caller (the reference pins no version).
        gen = kwargs.get("skimage")
        if gen is not None:
            self.anti_aliasing = runtime.resize_generation(str(gen))
        elif "anti_aliasing" in kwargs:
            self.anti_aliasing = runtime.resize_generation(kwargs["anti_aliasing"])
        else:
            self.anti_aliasing = 0
            runtime.warn_unpinned_generation("pix2pose(...)")
        # precision="f16x3" (default), "f32", or "auto" (split-f16 with an fp32 twin the object falls back to on a range event)
        self.generator_train = runtime.Generator(weights, backbone, ctx, precision=kwargs.get("precision", "f16x3"))
        self._inject = None                    # TEST / BENCH ONLY: (inject1_ptr, inject2_ptr, slots) device maps that replace the decoder outputs

    def _spec(self):
        return runtime.ObjectSpec(self.generator_train, np.concatenate([np.asarray(self.obj_scale, float),
                                                                        np.asarray(self.obj_ct, float)]),
                                  self.th_o, self.th_i, self.box_size)

    def get_boxes(self, bbox, v_max, u_max, ct=np.array([-1]), max_w=9999):
        return get_boxes(bbox, v_max, u_max, self.box_size, ct, max_w)

    def est_pose(self, rgb, bbox, gt_trans=np.eye((4)), z_iter=False):
        """-> (img_pred u8[h,w,3], valid_mask bool[H,W], R[3,3], t[3] (mm), frac_inlier, [v1,v2,u1,u2])
        or (placeholder, -1, -1, -1, -1, box) on failure (reference recognition.py:79,127,191,193)."""
        rgb = np.asarray(rgb)
        H, Wd = rgb.shape[0], rgb.shape[1]
        inj = {} if self._inject is None else dict(inject1=self._inject[0], inject2=self._inject[1], inject_slots=self._inject[2])
        poses, ex = runtime.est_pose_batch(self.ctx, [self._spec()], [rgb], [(0, 0, [int(b) for b in bbox], self.camK)],
                                           want_masks=True, anti_aliasing=self.anti_aliasing, **inj)
        p = poses[0]
        box = np.array(list(p.bbox_t), int)
        if p.status != 0:
            return self._failure_preview(p.status, rgb, bbox, inj), -1, -1, -1, -1, box
        v1, v2, u1, u2 = box
        img_pred = ex["img_pred"][0][:(v2 - v1) * (u2 - u1) * 3].reshape(v2 - v1, u2 - u1, 3).copy()
        mask = ex["valid_mask"][0][:H * Wd].reshape(H, Wd).astype(bool)
        return img_pred, mask, np.array(p.R, float).reshape(3, 3), np.array(p.t, float), p.frac_inlier, box

    def _failure_preview(self, status, rgb, bbox, inj):
        """First element of the reference's failure returns: np.zeros(1) at recognition.py:79; the stage-1 preview (decode + 1) / 2, clipped, at
        :127; at :191 `img_pred` is what the candidate loop assigned last (:141-143: the last stage-2 answer, gray pixels zeroed first).  No
        caller reads it, so it is fetched by a second (debug-tap) call on this path only."""
        if status == 1:                                    # _lib: P2P_POSE_CROP_TOO_SMALL
            return np.zeros((1))
        _, ex = runtime.est_pose_batch(self.ctx, [self._spec()], [rgb], [(0, 0, [int(b) for b in bbox], self.camK)],
                                       debug=True, anti_aliasing=self.anti_aliasing, **inj)
        if status == 2:                                    # P2P_POSE_NO_CANDIDATE
            decode = ex["y1"][0][..., :3].copy()
        else:                                              # P2P_POSE_PNP_FAILED
            k = max(i for i in range(ex["cand"].shape[1]) if ex["cand"][0, i, 0])
            decode = ex["y2"][0, k][..., :3].copy()
            decode[np.linalg.norm(decode, axis=2) < 0.3] = 0
        img_pred = (decode + 1) / 2
        img_pred[img_pred > 1] = 1
        img_pred[img_pred < 0] = 0
        return img_pred

    def est_pose_batch(self, rgbs, bboxes, camKs=None):
        """Many detections of this object at once (the reason this library exists).
        -> list of p2p_pose records (pix2pose_amd._lib.Pose)."""
        dets = [(i, 0, [int(b) for b in bboxes[i]], self.camK if camKs is None else camKs[i]) for i in range(len(bboxes))]
        return runtime.est_pose_batch(self.ctx, [self._spec()], list(rgbs), dets, anti_aliasing=self.anti_aliasing)[0]

    def pnp_ransac(self, rgb_aug_test, img_prob_ori, non_zero, v1, v2, u1, u2):
        """Reference recognition.py:195-224 with the solve on the GPU."""
        xyz = np.copy(rgb_aug_test[v1:v2, u1:u2]).astype(np.float64) / 255 * 2 - 1
        for k in range(3):
            xyz[:, :, k] = xyz[:, :, k] * self.obj_scale[k] + self.obj_ct[k]
        valid_mask = np.logical_and(non_zero, img_prob_ori < self.th_i)
        vs, us = np.where(valid_mask == 1)
        if len(vs) < 6:
            return np.eye(3), np.array([0, 0, 0]), valid_mask, -1
        obj = xyz[vs, us]
        img = np.stack((us + u1, vs + v1), axis=1).astype(np.float64)
        ok, R, t, info, _ = runtime.pnp_ransac_batch(self.ctx, [np.asarray(self.camK, float)], [obj], [img])
        if not ok[0]:
            return np.eye(3), np.array([0, 0, 0]), -1, -1
        return R[0], t[0], valid_mask, int(info[0, 0])
