"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (numpy) restatement of ``pix2pose.est_pose`` / ``get_boxes`` / ``pnp_ransac``
(reference pix2pose_model/recognition.py:28-69, :70-193, :195-224), including the behavioural
quirks listed in SURVEY.md section 8a-Q.  Third-party pieces are replaced by restatements:

  * ``skimage.transform.resize(order=1)``  -> :func:`resize_bilinear` (half-pixel
    centres, per-tap border handling 'reflect' / 'constant'+cval, ``clip=True`` output clipping to the input's
    range with cval preservation).  The scikit-image version is unpinned in the reference (requirements.txt does
    not list it): ``anti_aliasing=False`` (default here) is scikit-image <= 0.14, ``anti_aliasing=True`` is the
    0.17 - 0.18 default: ``scipy.ndimage.gaussian_filter`` with sigma = (in/out - 1)/2 per down-scaled axis --
    the REAL scipy routine skimage calls, not a restatement -- before the warp.  (>= 0.19 refuses the bool
    array recognition.py:103 passes, so the reference cannot run there at all.)
  * ``cv2.solvePnPRansac`` / ``cv2.Rodrigues`` -> oracle/pnp_oracle.c
  * ``generator_train.predict``            -> any callable (oracle/ae_oracle.forward, or injected
    decoder outputs for the synthetic PnP scenes)

PINNING: tests/golden/reference_est_pose.json holds outputs of the reference's own recognition.py
(est_pose / get_boxes / pnp_ransac executed unmodified in the build container, tests/golden/
make_reference_vectors.py) and this file reproduces them bit for bit (tests/test_reference_vectors_cpu.py).
The three library restatements above stay UNPINNED: none of keras / cv2 / skimage is installable here.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import pnp_oracle


# --------------------------------------------------------------------------------------
# skimage.transform.resize(order=1) restatement
# --------------------------------------------------------------------------------------
def _map_reflect(i, n):
    """numpy-pad 'reflect' (edge sample not repeated), any integer index -> [0, n)."""
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def _warp_bilinear_f32(a, out_shape, mode, cval):
    """scikit-image 0.17 / 0.18 ``_warp_fast[float32]`` + ``bilinear_interpolation[float32]`` for a float32 image, operation for
    operation as compiled in the 0.18.3 wheel (checked bit for bit against the real library: tests/golden/external_vectors.json,
    tools/make_external_vectors.py).  ``warp`` casts the 3x3 matrix to the image dtype; ``_transform_metric`` evaluates
    c = M00 * col + M02 in float32 (two roundings); taps floorf / ceilf; d = c - floor in float32; then, per pixel,
        top    = (1.0 - (double)dc) * (double)tl + (double)(dc * tr)        [dc * tr is a FLOAT32 product]
        bottom = (1.0 - (double)dc) * (double)bl + (double)(dc * br)
        out    = (float)((1.0 - (double)dr) * top + (double)dr * bottom)
    (Cython types the literal in ``1 - dc`` as a C double; ``dc * pixel`` stays float * float)."""
    f32, f64 = np.float32, np.float64
    h, w = a.shape[:2]
    oh, ow = out_shape

    def axis(n_in, n_out):
        s = n_in / n_out                                  # factors[i], float64
        m_s, m_t = f32(s), f32(s * 0.5 - 0.5)             # the affine's scale / translation, cast to the image dtype
        src = (m_s * np.arange(n_out, dtype=f32)).astype(f32) + m_t
        lo = np.floor(src)
        return lo.astype(np.int64), np.ceil(src).astype(np.int64), (src - lo).astype(f32)

    r0, r1, dr = axis(h, oh)
    c0, c1, dc = axis(w, ow)

    def tap(ri, ci):
        if mode == "reflect":
            return a[_map_reflect(ri, h)][:, _map_reflect(ci, w)]
        ok = ((ri >= 0) & (ri < h))[:, None] & ((ci >= 0) & (ci < w))[None, :]
        v = a[np.clip(ri, 0, h - 1)][:, np.clip(ci, 0, w - 1)]
        if a.ndim == 3:
            ok = ok[..., None]
        return np.where(ok, v, f32(cval)).astype(f32)

    if a.ndim == 3:
        dcb, drb = dc[None, :, None], dr[:, None, None]
    else:
        dcb, drb = dc[None, :], dr[:, None]
    dcd, drd = dcb.astype(f64), drb.astype(f64)
    top = (1.0 - dcd) * tap(r0, c0).astype(f64) + (dcb * tap(r0, c1)).astype(f32).astype(f64)
    bot = (1.0 - dcd) * tap(r1, c0).astype(f64) + (dcb * tap(r1, c1)).astype(f32).astype(f64)
    return ((1.0 - drd) * top + drd * bot).astype(f32)


def generation(spec):
    """scikit-image GENERATION of the six resize calls (p2p_est_pose_opts.resize_anti_aliasing): False / 0 = <= 0.14 (no anti-aliasing,
    everything in double); True / 1 = 0.17 - 0.18 (filter on for float images, off for bool ones; float32 images warped in float32);
    2 = 0.15 / 0.16 (filter on for EVERY image as passed -- the bool keep mask of recognition.py:103 included, with scipy's bool output --
    and ``_warp_fast`` converting every image to double)."""
    g = {False: 0, True: 1}.get(spec, spec) if isinstance(spec, bool) else int(spec)
    if g not in (0, 1, 2):
        raise ValueError("resize generation must be 0, 1 or 2")
    return g


def resize_gen(img, out_shape, mode, cval, gen):
    """One resize call of est_pose under scikit-image generation ``gen`` (see :func:`generation`)."""
    return resize_bilinear(img, out_shape, mode, cval, anti_aliasing=gen > 0, keep_float32=gen == 1, filter_bool=gen == 2)


def resize_bilinear(img, out_shape, mode, cval=0.0, anti_aliasing=False, clip=True, keep_float32=None, filter_bool=False):
    """img [H,W] or [H,W,C] (bool/float) -> float64 [oh,ow(,C)]  (float32 for a float32 image when ``keep_float32``).
    keep_float32 (default: = anti_aliasing, i.e. the two switches select a scikit-image GENERATION -- <= 0.14: no filter, everything
    in double; 0.17 / 0.18: filter on, and ``warp`` keeps a float32 image float32: matrix, coordinates and result in float32, see
    :func:`_warp_bilinear_f32`; the float32 images of the path are the prob map of recognition.py:134 and img_pred of :144).
    src = dst*scale + (0.5*scale - 0.5), scale = in/out; taps floor/ceil; out-of-range taps are
    reflected ('reflect') or replaced by cval ('constant').
    filter_bool: with anti_aliasing, a bool image is filtered too (scikit-image 0.15 / 0.16), see below.
    anti_aliasing (skimage 0.17-0.18 semantics: on by default for float images, off for bool ones): Gaussian pre-filter, sigma = max(0, (in/out - 1)/2) per axis, truncated at
    4 sigma, border mode 'mirror' (for 'reflect') or 'constant' with cval; the filter keeps the input's dtype (a float32
    map is rounded to float32 after each axis pass, exactly what scipy does for skimage).
    clip (skimage default): the output is clipped to [min, max] of the (filtered) input; in 'constant' mode with cval
    outside that range, pixels exactly equal to cval are kept (skimage._shared / transform._warps._clip_warp_output)."""
    a0 = np.asarray(img)
    if a0.dtype == bool and anti_aliasing and filter_bool:
        # scikit-image 0.15 / 0.16: resize() hands the image AS PASSED to scipy.ndimage.gaussian_filter, whose output takes the input's dtype:
        # for a bool array every axis pass ends in a C cast double -> npy_bool, i.e. a pixel survives a pass only where its weighted sum
        # reaches 1.0 -- an erosion that, depending on how the weights' sum rounds, leaves part of the mask or NOTHING of it.
        from scipy import ndimage as ndi
        hb, wb = a0.shape[:2]
        sig = [max(0.0, (hb / out_shape[0] - 1) / 2), max(0.0, (wb / out_shape[1] - 1) / 2)]
        if max(sig) > 0:
            a0 = ndi.gaussian_filter(a0, sig, cval=cval, mode="mirror" if mode == "reflect" else "constant")
            assert a0.dtype == bool
        anti_aliasing = False
    if a0.dtype == bool:
        # scikit-image 0.17 / 0.18: `anti_aliasing` defaults to "not a bool image" ("Gaussian convolution is not defined with bool data
        # type": a FutureWarning there, a ValueError from 0.19 on) -- the one bool input of the path is the keep mask of recognition.py:103.
        # (0.15 / 0.16 ran scipy's filter on the bool array, whose bool OUTPUT keeps only pixels whose weighted sum is exactly 1.0: an
        # erosion down to a few percent of the mask.  That generation is not modelled.)
        anti_aliasing = False
    if a0.dtype == bool or a0.dtype.kind not in "f":
        a0 = a0.astype(np.float64)                  # img_as_float(bool / ints used here) -> float64
    h, w = a0.shape[:2]
    oh, ow = out_shape
    if anti_aliasing:
        from scipy import ndimage as ndi
        sig = [max(0.0, (h / oh - 1) / 2), max(0.0, (w / ow - 1) / 2)] + [0.0] * (a0.ndim - 2)
        if max(sig) > 0:
            a0 = ndi.gaussian_filter(a0, sig, cval=cval, mode="mirror" if mode == "reflect" else "constant")
    lo_v, hi_v = (float(a0.min()), float(a0.max())) if a0.size else (0.0, 0.0)
    if keep_float32 is None:
        keep_float32 = bool(anti_aliasing)
    if keep_float32 and a0.dtype == np.float32:
        out = _warp_bilinear_f32(a0, (oh, ow), mode, cval)
        if clip:                                                    # _clip_warp_output on the float32 arrays
            preserve = mode == "constant" and not (lo_v <= cval <= hi_v)
            keep = (out == np.float32(cval)) if preserve else None
            out = np.clip(out, np.float32(lo_v), np.float32(hi_v))
            if preserve:
                out[keep] = cval
        return out
    a = a0.astype(np.float64)

    def axis(n_in, n_out):
        s = n_in / n_out
        src = np.arange(n_out, dtype=np.float64) * s + (0.5 * s - 0.5)
        lo = np.floor(src)
        hi = np.ceil(src)
        return lo.astype(np.int64), hi.astype(np.int64), src - lo

    r0, r1, dr = axis(h, oh)
    c0, c1, dc = axis(w, ow)

    def tap(ri, ci):
        if mode == "reflect":
            return a[_map_reflect(ri, h)][:, _map_reflect(ci, w)]
        ok_r = (ri >= 0) & (ri < h)
        ok_c = (ci >= 0) & (ci < w)
        v = a[np.clip(ri, 0, h - 1)][:, np.clip(ci, 0, w - 1)]
        ok = ok_r[:, None] & ok_c[None, :]
        if a.ndim == 3:
            ok = ok[..., None]
        return np.where(ok, v, cval)

    if a.ndim == 3:
        dcb = dc[None, :, None]
        drb = dr[:, None, None]
    else:
        dcb = dc[None, :]
        drb = dr[:, None]
    top = (1 - dcb) * tap(r0, c0) + dcb * tap(r0, c1)
    bot = (1 - dcb) * tap(r1, c0) + dcb * tap(r1, c1)
    out = (1 - drb) * top + drb * bot
    if clip:
        preserve = mode == "constant" and not (lo_v <= cval <= hi_v)
        keep = (out == cval) if preserve else None
        out = np.clip(out, lo_v, hi_v)
        if preserve:
            out[keep] = cval
    return out


# --------------------------------------------------------------------------------------
# crop geometry  (recognition.py:28-69)
# --------------------------------------------------------------------------------------
@dataclass
class Boxes:
    v1_ori: int
    v2_ori: int
    u1_ori: int
    u2_ori: int
    v1: int
    v2: int
    u1: int
    u2: int
    vv1: int
    vv2: int
    uu1: int
    uu2: int

    def as_list(self):
        return [self.v1_ori, self.v2_ori, self.u1_ori, self.u2_ori, self.v1, self.v2, self.u1, self.u2,
                self.vv1, self.vv2, self.uu1, self.uu2]


def get_boxes(bbox, v_max, u_max, box_size=1.5, ct=None, max_w=9999) -> Boxes:
    """Square crop of side 2*int(w/2), w = min(max_w, box_size*max(width,height)), centred on the
    bbox centre (or ``ct``), clipped to the image; plus paste offsets into a zero canvas."""
    if ct is None or ct[0] == -1:                            # :29-31 (the sentinel is ct[0] == -1: a genuine centre row of -1 also lands here)
        ct_v = int((bbox[0] + bbox[2]) / 2)
        ct_u = int((bbox[1] + bbox[3]) / 2)
    else:                                                    # :32-34
        ct_v, ct_u = ct[0], ct[1]
    width = bbox[3] - bbox[1]                                # :36-37
    height = bbox[2] - bbox[0]
    w = min(max_w, max(width * box_size, height * box_size))  # :38
    half = int(w / 2)
    v1o, v2o = ct_v - half, ct_v + half                      # :40-43
    u1o, u2o = ct_u - half, ct_u + half
    v1, v2, u1, u2 = v1o, v2o, u1o, u2o
    sv0 = su0 = sv1 = su1 = 0
    if v1o < 0:                                              # :53-55
        sv0 = abs(v1o); v1 = 0
    if v2o > v_max:                                          # :56-58
        sv1 = -abs(v2o - v_max); v2 = v_max
    if u1o < 0:                                              # :59-61
        su0 = abs(u1o); u1 = 0
    if u2o > u_max:                                          # :62-64
        su1 = -abs(u2o - u_max); u2 = u_max
    return Boxes(int(v1o), int(v2o), int(u1o), int(u2o), int(v1), int(v2), int(u1), int(u2),
                 int(sv0), int(sv1 + (v2o - v1o)), int(su0), int(su1 + (u2o - u1o)))   # :65-69


# --------------------------------------------------------------------------------------
# PnP front end  (recognition.py:195-224)
# --------------------------------------------------------------------------------------
def correspondences(rgb_aug, img_prob_ori, non_zero, v1, v2, u1, u2, obj_scale, obj_ct, th_i):
    """u8 XYZ canvas -> (obj_pts [n,3] mm, img_pts [n,2] (u,v), valid_mask)  (:196-213)."""
    xyz = np.copy(rgb_aug[v1:v2, u1:u2]).astype(np.float64)
    xyz = xyz / 255
    xyz = xyz * 2 - 1
    for k in range(3):
        xyz[:, :, k] = xyz[:, :, k] * obj_scale[k] + obj_ct[k]
    valid = np.logical_and(non_zero, img_prob_ori < float(th_i))     # a float32 img_prob_ori (scikit-image >= 0.17) is compared in float32
    vs, us = np.where(valid == 1)                     # row-major order
    obj_pts = xyz[vs, us]
    img_pts = np.stack((us + u1, vs + v1), axis=1).astype(np.float64)
    return obj_pts, img_pts, valid


def pnp_ransac(rgb_aug, img_prob_ori, non_zero, v1, v2, u1, u2, camK, obj_scale, obj_ct, th_i):
    obj_pts, img_pts, valid = correspondences(rgb_aug, img_prob_ori, non_zero, v1, v2, u1, u2, obj_scale, obj_ct, th_i)
    if len(obj_pts) < 6:                              # :214-215
        return np.eye(3), np.array([0, 0, 0]), valid, -1, None
    ok, R, t, inl, meta = pnp_oracle.solve_pnp_ransac(obj_pts, img_pts, camK, iterations=100, reproj_err=5.0)
    if not ok:                                        # :218-219 (inliers is None)
        return np.eye(3), np.array([0, 0, 0]), -1, -1, meta
    return R, t, valid, len(inl), meta                # :220-224


# --------------------------------------------------------------------------------------
# est_pose  (recognition.py:70-193)
# --------------------------------------------------------------------------------------
def est_pose(rgb, bbox, predict, camK, obj_param, th_outlier=(0.1, 0.2, 0.3), th_inlier=0.1, box_size=1.5,
             debug=None, anti_aliasing=False):
    """Returns the reference's 6-tuple.  ``predict(x[N,128,128,3]) -> [decode, prob]``.
    ``debug`` (dict) receives intermediates for stage-wise parity tests.  ``anti_aliasing``: the scikit-image generation, see
    :func:`generation` (False / True kept for the first two)."""
    camK = np.asarray(camK, np.float64).reshape(3, 3)
    obj_scale, obj_ct = np.asarray(obj_param[:3], float), np.asarray(obj_param[3:], float)
    H, W = rgb.shape[0], rgb.shape[1]
    dbg = debug if debug is not None else {}

    b1 = get_boxes(bbox, H, W, box_size)                                             # :71
    cx_o = (bbox[3] + bbox[1]) / 2                                                   # :72-73
    cy_o = (bbox[2] + bbox[0]) / 2
    w_stage_1 = b1.v2_ori - b1.v1_ori                                                # :74
    side = b1.v2_ori - b1.v1_ori
    base = np.zeros((side, b1.u2_ori - b1.u1_ori, 3))                               # :75
    crop = (np.copy(rgb[b1.v1:b1.v2, b1.u1:b1.u2]).astype(np.float32) - [128, 128, 128]) / 128     # :76-77
    fail_box = np.array([b1.v1, b1.v2, b1.u1, b1.u2], int)
    if base.shape[0] < 5 or base.shape[1] < 5 or crop.shape[0] < 5 or crop.shape[1] < 5:   # :78-79
        return np.zeros((1)), -1, -1, -1, -1, fail_box
    base[b1.vv1:b1.vv2, b1.uu1:b1.uu2] = crop                                        # :81
    gen = generation(anti_aliasing)
    x1 = resize_gen(base, (128, 128), "reflect", 0, gen)                             # :82
    dbg["x1"] = x1.astype(np.float32)
    decode, prob = predict(np.expand_dims(x1, 0), stage=1)                           # :84
    decode = np.array(decode, np.float32)
    img_pred = np.clip((decode[0] + 1) / 2, 0, 1)                                    # :85-87
    non_gray = np.linalg.norm(decode[0], axis=2) > 0.3                               # :89
    n_init_mask = int(np.sum(non_gray))                                              # :90
    dbg["n_init_mask"] = n_init_mask
    dbg["decode1"], dbg["prob1"] = decode[0], np.array(prob[0, :, :, 0], np.float32)

    inputs, boxes, slots = [], [], []
    for slot, th_o in enumerate(th_outlier):                                         # :93
        keep = np.logical_and(non_gray, prob[0, :, :, 0] < th_o)                     # :94-95
        if np.sum(keep) < 10:                                                        # :96-97
            continue
        vs, us = np.where(non_gray)                                                  # :98
        if len(vs) == 0:
            continue
        bb = np.array([vs.min(), us.min(), vs.max(), us.max()])                      # :101 (of non_gray, not keep)
        bb = bb * np.array([side / 128, (b1.u2_ori - b1.u1_ori) / 128] * 2)          # :102
        keep_ori = resize_gen(keep, (side, b1.u2_ori - b1.u1_ori), "constant", 0, gen) > 0.9     # :103 (a BOOL image)
        keep_ori = keep_ori[b1.vv1:b1.vv2, b1.uu1:b1.uu2]                            # :104
        bg_full = np.ones((H, W), bool)                                              # :105-106
        bg_full[b1.v1:b1.v2, b1.u1:b1.u2] = np.invert(keep_ori)
        cx_m = int((np.mean(us) - (127 / 2)) + cx_o)                                 # :108 (quirk: 128-res offset)
        cy_m = int((np.mean(vs) - (127 / 2)) + cy_o)                                 # :109
        b2 = get_boxes(bb, H, W, box_size, ct=np.array([cy_m, cx_m]), max_w=w_stage_1)   # :110
        base2 = np.zeros((b2.v2_ori - b2.v1_ori, b2.u2_ori - b2.u1_ori, 3))          # :113
        crop2 = (np.copy(rgb[b2.v1:b2.v2, b2.u1:b2.u2]) - [128, 128, 128]) / 128     # :114-115
        if crop2.shape[0] > 0 and crop2.shape[1] > 0:
            crop2[bg_full[b2.v1:b2.v2, b2.u1:b2.u2]] = 0                             # :116
        tgt = base2[b2.vv1:b2.vv2, b2.uu1:b2.uu2]
        if (base2.shape[0] < 5 or base2.shape[1] < 5 or crop2.shape[0] < 5 or crop2.shape[1] < 5
                or tgt.shape[0] == 0 or tgt.shape[1] == 0):                          # :117-119
            continue
        # NOTE the reference appends to box_refined (:111) BEFORE this `continue`, which would
        # mis-align boxes and inputs afterwards; we keep them aligned (the misaligned case needs
        # a <5 px re-crop and does not occur for boxes that passed the stage-1 check).
        base2[b2.vv1:b2.vv2, b2.uu1:b2.uu2] = crop2                                  # :120
        inputs.append(resize_gen(base2, (128, 128), "reflect", 0, gen))              # :121-122
        boxes.append(b2)
        slots.append(slot)
    dbg["slots"] = slots
    dbg["boxes2"] = [b.as_list() for b in boxes]
    if len(inputs) <= 0:                                                             # :125-127
        return img_pred, -1, -1, -1, -1, fail_box
    dbg["x2"] = np.array(inputs, np.float32)

    decode, prob = predict(np.array(inputs), stage=2, slots=slots)                   # :129
    decode = np.array(decode, np.float32)
    prob = np.array(prob, np.float32)
    max_inlier, min_dist = -1, 9999999                                               # :130-131
    rot_pred = tra_pred = valid_mask_full = img_pred_f = None
    dbg["cands"] = []
    last = b1
    for c in range(len(inputs)):                                                     # :132
        b = boxes[c]
        last = b
        side2, wid2 = b.v2_ori - b.v1_ori, b.u2_ori - b.u1_ori
        prob_ori = resize_gen(prob[c, :, :, 0], (side2, wid2), "constant", 1, gen)   # :134 (float32 map)
        prob_ori = prob_ori[b.vv1:b.vv2, b.uu1:b.uu2]                                # :135
        gray = np.linalg.norm(decode[c], axis=2) < 0.3                               # :137
        ng = np.invert(gray)                                                         # :138
        decode[c, gray, :] = 0                                                       # :139
        pred = np.clip((decode[c] + 1) / 2, 0, 1)                                    # :141-143
        pred_ori = resize_gen(pred, (side2, wid2), "constant", 0.5, gen) * 255       # :144 (float32 map)
        ng = resize_gen(ng.astype(float), (side2, wid2), "constant", 0, gen) > 0.9   # :146
        ng = ng[b.vv1:b.vv2, b.uu1:b.uu2]                                            # :147
        n_non_gray = int(np.sum(ng))                                                 # :148
        cd = {"slot": slots[c], "n_non_gray": n_non_gray}
        dbg["cands"].append(cd)
        if n_non_gray < 10:                                                          # :149-150
            continue
        pred_ori = pred_ori[b.vv1:b.vv2, b.uu1:b.uu2]                                # :151
        canvas = np.zeros((H, W, 3), np.uint8)                                       # :152
        canvas[b.v1:b.v2, b.u1:b.u2] = pred_ori                                      # :153-154 (float -> u8 truncation)
        R_c, t_c, valid, n_inl, meta = pnp_ransac(canvas, prob_ori, ng, b.v1, b.v2, b.u1, b.u2, camK,
                                                  obj_scale, obj_ct, th_inlier)      # :156
        ng_full = np.zeros((H, W), bool)                                             # :159-162
        ng_full[b.v1:b.v2, b.u1:b.u2] = ng
        fv, fu = np.where(ng_full)
        ct_pt = np.array([np.mean(fv), np.mean(fu)])
        if t_c[2] == 0:                                                              # :163-164
            dist = 99999
        else:                                                                        # :165-168
            pu = camK[0, 0] * t_c[0] / t_c[2] + camK[0, 2]
            pv = camK[1, 1] * t_c[1] / t_c[2] + camK[1, 2]
            dist = ((pv - ct_pt[0]) ** 2 + (pu - ct_pt[1]) ** 2) / (n_inl + 1E-6)
        cd.update(R=np.array(R_c, float), t=np.array(t_c, float), n_inliers=n_inl, dist=float(dist), meta=meta,
                  n_valid=int(np.sum(valid)) if not isinstance(valid, int) else -1)
        if dist < min_dist:                                                          # :170-178
            rot_pred, tra_pred, max_inlier, min_dist = R_c, t_c, n_inl, dist
            valid_mask_full = np.zeros((H, W), bool)
            valid_mask_full[b.v1:b.v2, b.u1:b.u2] = valid      # valid == -1 (int) broadcasts to True, as in the reference
            img_pred_f = pred_ori
            dbg["best"] = c
    last_box = np.array([last.v1, last.v2, last.u1, last.u2], int)   # :133 overwrites v1.. -> box of the LAST candidate
    if max_inlier == -1:                                                             # :189-191
        return img_pred, -1, -1, -1, -1, last_box
    return (img_pred_f.astype(np.uint8), valid_mask_full, np.array(rot_pred, float), np.array(tra_pred, float),
            max_inlier / n_init_mask, last_box)                                      # :193
