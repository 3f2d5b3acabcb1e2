#!/usr/bin/env python
"""Pin the THIRD-PARTY LIBRARY SEMANTICS the oracle restates -- to be run OUTSIDE the build container.

The reference's arithmetic lives in tensorflow-gpu 1.9 / keras 2.2.1 (requirements.txt:1-2), opencv-python 3.4.2.17
(:3-4) and an unpinned scikit-image -- none of which exists in the GPU image (no network), so the oracle under oracle/
is a restatement and DESIGN.md section 4 calls those semantics "unpinned".  Anyone with such an environment closes the
gap with this script: it executes the REAL libraries on deterministic inputs and writes

    tests/golden/external_vectors.json

which tests/test_external_vectors.py consumes when present (CPU: the oracle; GPU: the HIP path) and skips otherwise.

    python tools/make_external_vectors.py [--reference /path/to/Pix2Pose] [--out tests/golden/external_vectors.json]

Sections (each is skipped, and recorded as skipped, when its library is missing):
  resize    skimage.transform.resize(order=1) on bool / float32 / float64 inputs, 'reflect' and 'constant' + cval, up and
            down; the installed version decides whether anti-aliasing is on by default (>= 0.15), and is recorded.
            Call sites: recognition.py:82,103,121,134,144,146.
  pnp       cv2.solvePnPRansac(flags=SOLVEPNP_EPNP, reprojectionError=5, iterationsCount=100) + cv2.Rodrigues on ten
            synthetic problems (recognition.py:216-223).
  layers    one probe per Keras layer kind the graphs use: Conv2D 'same' stride 2, ZeroPadding2D + 7x7 'valid' stride 2,
            Conv2DTranspose 5x5 stride 2 'same', BatchNormalization (inference), LeakyReLU(), MaxPooling2D 3x3/2 'same',
            Flatten + Dense (ae_model.py / resnet50_mod.py).
  graphs    (--reference) the reference's own builders aemodel_unet_prob / aemodel_unet_resnet50 with this repository's
            synthetic weights assigned by layer name, predict() on two inputs; also save_weights()/save() to HDF5 and back
            through pix2pose_amd.convert_keras (row f-2: the h5py reader on real files).
  est_pose  (--reference) recognition.pix2pose.est_pose UNSHIMMED except for the network (decoder maps injected), i.e.
            with the real skimage and cv2 -- the scenes of tests/golden/make_reference_vectors.py.
Only inputs' seeds / parameters and outputs are stored; nothing of the reference is copied.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pix2pose_amd import synthetic  # noqa: E402  (pure numpy)
from pix2pose_amd import weights as W  # noqa: E402

TH_O, TH_I = [0.2, 0.3, 0.35], 0.2


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def version_tuple(v):
    out = []
    for p in v.split(".")[:3]:
        digits = "".join(ch for ch in p if ch.isdigit())
        out.append(int(digits) if digits else 0)
    return out


# ---------------------------------------------------------------------------------------------- resize
RESIZE_CASES = [  # (n_in, n_out, mode, cval, dtype[, channels])
    (128, 190, "constant", 1.0, "float32"), (128, 190, "constant", 0.5, "float32"), (128, 190, "constant", 0.0, "bool"),
    (128, 74, "constant", 1.0, "float32"), (128, 74, "constant", 0.0, "bool"), (128, 74, "constant", 0.5, "float32"),
    (300, 128, "reflect", 0.0, "float64"), (90, 128, "reflect", 0.0, "float64"), (128, 128, "reflect", 0.0, "float64"),
    (128, 160, "constant", 0.0, "float64"), (17, 128, "reflect", 0.0, "float64"), (128, 9, "constant", 1.0, "float32"),
    # round 4: the float32 images of the path at more scales -- prob (:134, cval 1) single channel, img_pred (:144, cval 0.5) 3 channels
    (128, 263, "constant", 1.0, "float32"), (128, 301, "constant", 0.5, "float32", True), (128, 450, "constant", 1.0, "float32"),
    (128, 129, "constant", 0.5, "float32", True), (128, 127, "constant", 1.0, "float32"), (128, 100, "constant", 0.5, "float32", True),
    (128, 41, "constant", 1.0, "float32"), (128, 128, "constant", 0.5, "float32", True), (128, 200, "constant", 0.0, "float64"),
    # 0/1 masks thresholded at > 0.9 (:103 bool, :146 float64) at sides that are NOT multiples of 10: there a border row's bilinear weight is
    # exactly 0.9 and the real library's own matrix noise decides (tests/golden/make_reference_vectors.py) -- nothing to hold anyone to
    (128, 173, "constant", 0.0, "bool"), (128, 251, "constant", 0.0, "float64"), (128, 97, "constant", 0.0, "float64"),
    (128, 333, "constant", 0.0, "bool")]


def case_channels(case):
    """3-channel input?  The float64 'reflect' cases are the canvases of :82,121; a 6th tuple element says so explicitly."""
    return bool(case[5]) if len(case) > 5 else (case[4] == "float64" and case[2] == "reflect")


def resize_input(case_idx, n_in, dtype, channels):
    rs = np.random.RandomState(9000 + case_idx)
    shape = (n_in, n_in, 3) if channels else (n_in, n_in)
    a = rs.rand(*shape)
    if dtype == "bool":
        return a > 0.6
    if case_idx >= 21 and dtype == "float64":          # the float64 masks of recognition.py:146: non_gray.astype(float), values 0.0 / 1.0 in blobs
        from scipy import ndimage as ndi
        return (ndi.uniform_filter(a, 9) > 0.5).astype(np.float64)
    return a.astype(dtype)


def section_resize():
    import skimage
    from skimage.transform import resize
    cases = []
    for i, case in enumerate(RESIZE_CASES):
        n_in, n_out, mode, cval, dtype = case[:5]
        channels = case_channels(case)
        a = resize_input(i, n_in, dtype, channels)
        raw = resize(a, (n_out, n_out), order=1, mode=mode, cval=cval)
        r = np.asarray(raw, np.float64)
        cases.append({"n_in": n_in, "n_out": n_out, "mode": mode, "cval": cval, "dtype": dtype, "channels": bool(channels),
                      "out_dtype": str(raw.dtype), "crc": crc(raw),            # the result's own bits (a float32 image stays float32 from 0.16 on)
                      "u8_crc": crc((raw * 255).astype(np.uint8)),            # recognition.py:144,152: (resize(...) * 255) stored into a uint8 canvas
                      "lt02_crc": crc(np.packbits(raw < 0.2)),                # recognition.py:203: img_prob_ori < th_inlier
                      "gt09_crc": crc(np.packbits(raw > 0.9)),                # recognition.py:103,146: resize(mask) > 0.9
                      "sum": float(r.sum()), "min": float(r.min()), "max": float(r.max()),
                      "diag": [float(v) for v in (r[np.arange(n_out), np.arange(n_out)].reshape(n_out, -1)[:, 0])],
                      "first_row": [float(v) for v in r[0].reshape(n_out, -1)[:, 0]]})
    v = skimage.__version__
    import scipy
    return {"version": v, "scipy_version": scipy.__version__, "numpy_version": np.__version__,
            "anti_aliasing_default": version_tuple(v) >= [0, 15, 0], "cases": cases}


# ---------------------------------------------------------------------------------------------- PnP
def pnp_problems(n_prob=10, seed0=100):
    rs = np.random.RandomState(seed0)
    out = []
    for _ in range(n_prob):
        n = int(rs.randint(6, 3000))
        R = synthetic.random_rotation(rs)
        t = np.array([rs.uniform(-60, 60), rs.uniform(-60, 60), rs.uniform(400, 1200)])
        P = rs.uniform(-1, 1, (n, 3)) * synthetic.OBJ_PARAM[:3]
        uv = synthetic.project(synthetic.LM_K, R, t, P) + 0.3 * rs.randn(n, 2)
        n_out = int(rs.uniform(0.0, 0.4) * n)
        uv[:n_out] += rs.uniform(20, 60, (n_out, 2)) * rs.choice([-1, 1], (n_out, 2))
        out.append((P, uv))
    return out


def section_pnp():
    import cv2
    res = []
    for P, uv in pnp_problems():
        ok, rvec, tvec, inl = cv2.solvePnPRansac(P, np.ascontiguousarray(uv).reshape(-1, 1, 2), synthetic.LM_K, None,
                                                 flags=cv2.SOLVEPNP_EPNP, reprojectionError=5, iterationsCount=100)
        if inl is None:
            res.append({"ok": False})
            continue
        R = np.eye(3)
        cv2.Rodrigues(rvec, R)
        res.append({"ok": True, "n": int(len(P)), "rvec": rvec.reshape(3).tolist(), "t": tvec.reshape(3).tolist(), "R": R.tolist(),
                    "n_inliers": int(len(inl)), "inliers_crc": crc(np.asarray(inl, np.int32).reshape(-1))})
    return {"version": cv2.__version__, "seed0": 100, "problems": res}


# ---------------------------------------------------------------------------------------------- Keras layer probes
def section_layers():
    import keras
    from keras.layers import (BatchNormalization, Conv2D, Conv2DTranspose, Dense, Flatten, Input, LeakyReLU, MaxPooling2D,
                              ZeroPadding2D)
    from keras.models import Model
    rs = np.random.RandomState(7)
    out = {"version": keras.__version__}

    def run(layers, x, weights):
        inp = Input(x.shape[1:])
        y = inp
        for lyr in layers:
            y = lyr(y)
        m = Model(inp, y)
        k = 0
        for lyr in layers:
            n = len(lyr.get_weights())
            if n:
                lyr.set_weights(weights[k:k + n])
                k += n
        return m.predict(x)

    x = rs.randn(1, 8, 8, 3).astype(np.float32)
    k, b = rs.randn(5, 5, 3, 4).astype(np.float32), rs.randn(4).astype(np.float32)
    out["conv_same_s2"] = {"x": x.tolist(), "kernel": k.tolist(), "bias": b.tolist(),
                           "y": run([Conv2D(4, (5, 5), strides=(2, 2), padding="same")], x, [k, b]).tolist()}
    x = rs.randn(1, 10, 10, 3).astype(np.float32)
    k, b = rs.randn(7, 7, 3, 2).astype(np.float32), rs.randn(2).astype(np.float32)
    out["zeropad3_conv7_valid_s2"] = {"x": x.tolist(), "kernel": k.tolist(), "bias": b.tolist(),
                                      "y": run([ZeroPadding2D((3, 3)), Conv2D(2, (7, 7), strides=(2, 2))], x, [k, b]).tolist()}
    x = rs.randn(1, 6, 6, 3).astype(np.float32)
    k, b = rs.randn(5, 5, 4, 3).astype(np.float32), rs.randn(4).astype(np.float32)       # (kh, kw, Cout, Cin)
    out["deconv_same_s2"] = {"x": x.tolist(), "kernel": k.tolist(), "bias": b.tolist(),
                             "y": run([Conv2DTranspose(4, (5, 5), strides=(2, 2), padding="same")], x, [k, b]).tolist()}
    x = rs.randn(1, 4, 4, 5).astype(np.float32)
    g, be, mu, var = (rs.rand(5).astype(np.float32) + 0.5, rs.randn(5).astype(np.float32), rs.randn(5).astype(np.float32),
                      rs.rand(5).astype(np.float32) * 2 + 1e-3)
    out["batchnorm_leaky"] = {"x": x.tolist(), "gamma": g.tolist(), "beta": be.tolist(), "mean": mu.tolist(), "var": var.tolist(),
                              "y": run([BatchNormalization(), LeakyReLU()], x, [g, be, mu, var]).tolist()}
    x = rs.randn(1, 8, 8, 2).astype(np.float32)
    out["maxpool_3x3_s2_same"] = {"x": x.tolist(), "y": run([MaxPooling2D((3, 3), strides=(2, 2), padding="same")], x, []).tolist()}
    x = rs.randn(2, 2, 2, 3).astype(np.float32)
    k, b = rs.randn(12, 5).astype(np.float32), rs.randn(5).astype(np.float32)
    out["flatten_dense"] = {"x": x.tolist(), "kernel": k.tolist(), "bias": b.tolist(), "y": run([Flatten(), Dense(5)], x, [k, b]).tolist()}
    return out


# ---------------------------------------------------------------------------------------------- graphs (+ HDF5 round trip)
def keras_name_of(canon, backbone):
    """canonical layer name (pix2pose_amd.weights) -> ('conv'|'bn', keras layer name or None for auto-named layers)."""
    if canon.startswith("res"):
        blk, br = canon[3:5], canon.split("_")[1]
        return ("res%s_branch%s" % (blk, br), "bn%s_branch%s" % (blk, br))
    if canon == "conv1" and backbone == "resnet50":
        return ("conv1", "bn_conv1")
    return (canon, None)


def assign_weights(model, w, backbone):
    """Set the canonical weights on a model built by the reference's builder (named layers by name, auto-named layers in
    creation order -- the same rule pix2pose_amd.convert_keras applies in the other direction)."""
    from pix2pose_amd import convert_keras as CK

    def all_layers(m):
        for lyr in m.layers:
            if hasattr(lyr, "layers"):
                for sub in all_layers(lyr):
                    yield sub
            else:
                yield lyr
    layers = list(all_layers(model))
    by_name = {lyr.name: lyr for lyr in layers}

    def numbered(prefix):
        found = []
        for lyr in layers:
            if lyr.name.startswith(prefix + "_") and lyr.name[len(prefix) + 1:].isdigit():
                found.append((int(lyr.name[len(prefix) + 1:]), lyr))
        return [lyr for _, lyr in sorted(found, key=lambda t: t[0])]

    def conv(canon, lyr):
        lyr.set_weights([w[canon + ".kernel"], w[canon + ".bias"]])

    def bn(canon, lyr):
        lyr.set_weights([w[canon + "." + k] for k in ("gamma", "beta", "mean", "var")])
    canons = sorted({k.rsplit(".", 1)[0] for k in w})
    for canon, lyr in zip(CK._BN_ORDER[backbone], numbered("batch_normalization")):
        bn(canon, lyr)
    for canon, lyr in zip(CK._DENSE_ORDER, numbered("dense")):
        conv(canon, lyr)
    for canon, lyr in zip(CK._DECONV_ORDER, numbered("conv2d_transpose")):
        conv(canon, lyr)
    auto = set(CK._DENSE_ORDER) | set(CK._DECONV_ORDER)
    for canon in canons:
        if canon in auto:
            continue
        cname, bname = keras_name_of(canon, backbone)
        conv(canon, by_name[cname])
        if bname is not None:
            bn(canon, by_name[bname])


def section_graphs(ref):
    sys.path.insert(0, ref)
    from pix2pose_model import ae_model as ae
    from pix2pose_amd import convert_keras as CK
    out = {}
    x = (np.random.RandomState(0).randint(0, 256, (2, 128, 128, 3)).astype(np.float32) - 128) / 128
    idx = np.random.RandomState(1).choice(2 * 128 * 128, 64, replace=False)
    for backbone in ("paper", "resnet50"):
        built = ae.aemodel_unet_prob(p=1.0) if backbone == "paper" else ae.aemodel_unet_resnet50(p=1.0)
        model = built[0] if isinstance(built, (tuple, list)) else built
        w = W.synthetic_weights(backbone, 11)
        assign_weights(model, w, backbone)
        dec, prob = model.predict(x)
        rec = {"weights_seed": 11, "pixel_index": idx.tolist(), "decode": dec.reshape(-1, 3)[idx].tolist(),
               "prob": prob.reshape(-1)[idx].tolist(), "decode_abs_mean": float(np.abs(dec).astype(np.float64).mean())}
        # HDF5 round trip through the converter (both artefact kinds of the reference: tools/3_train_pix2pose.py:273-276)
        for kind in ("weights", "model"):
            with tempfile.TemporaryDirectory() as td:
                fn = os.path.join(td, "x.hdf5")
                (model.save_weights if kind == "weights" else model.save)(fn)
                try:
                    back = CK.convert_named(CK.read_hdf5(fn), backbone)
                    rec["hdf5_%s_roundtrip_exact" % kind] = bool(all(np.array_equal(back[k], w[k]) for k in w))
                except Exception as e:  # noqa: BLE001 -- recorded, the test reports it
                    rec["hdf5_%s_roundtrip_exact" % kind] = "error: %s" % (e,)
        out[backbone] = rec
    return out


# ---------------------------------------------------------------------------------------------- est_pose with the real libraries
SCENES = [dict(seed=501, n_det=3, bbox_side=(86, 86)), dict(seed=502, n_det=4, bbox_side=(60, 140)),
          dict(seed=503, n_det=3, bbox_side=(150, 210)), dict(seed=512, n_det=4, bbox_side=(40, 84))]


class _Predict:
    def __init__(self, inj1, inj2):
        self.inj1, self.inj2, self.calls = inj1, inj2, 0

    def predict(self, x):
        self.calls += 1
        m = self.inj1[None] if self.calls == 1 else self.inj2[:x.shape[0]]
        return [m[..., :3].astype(np.float32).copy(), m[..., 3:].astype(np.float32).copy()]


def section_est_pose(ref):
    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, ref)
    from pix2pose_model import recognition as R
    scenes = []
    for spec in SCENES:
        sc = synthetic.make_scene(spec["n_det"], seed=spec["seed"], bbox_side=spec["bbox_side"])
        dets = []
        for i, (img_i, _, bbox, K) in enumerate(sc["dets"]):
            p = object.__new__(R.pix2pose)
            p.camK, p.res_x, p.res_y = np.asarray(K, float), 640, 480
            p.th_ransac, p.th_o, p.th_i = 3.0, TH_O, TH_I
            p.obj_scale, p.obj_ct = sc["obj_param"][:3], sc["obj_param"][3:]
            p.box_size, p.dist_coeff = 1.5, None
            p.generator_train = _Predict(sc["inject1"][i], sc["inject2"][i])
            r = p.est_pose(sc["images"][img_i], np.asarray(bbox))
            d = {"bbox": [int(b) for b in bbox], "bbox_t": [int(v) for v in r[5]]}
            if isinstance(r[1], int) and r[1] == -1:
                d["ok"] = False
            else:
                d.update({"ok": True, "R": np.asarray(r[2]).tolist(), "t": np.asarray(r[3]).tolist(), "frac_inlier": float(r[4]),
                          "mask_sum": int(np.sum(r[1])), "mask_crc": crc(np.packbits(r[1])), "img_pred_crc": crc(r[0])})
            dets.append(d)
        scenes.append({"spec": {k: (list(v) if isinstance(v, tuple) else v) for k, v in spec.items()}, "dets": dets})
    return {"scenes": scenes, "th_outlier": TH_O, "th_inlier": TH_I}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None, help="checkout of kirumang/Pix2Pose (enables the graphs and est_pose sections)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "external_vectors.json"))
    args = ap.parse_args()
    out = {"note": "outputs of the real third-party libraries (tools/make_external_vectors.py)", "skipped": {},
           "interpreter": "%s (python %s)" % (sys.executable, sys.version.split()[0])}
    sections = [("resize", section_resize), ("pnp", section_pnp), ("layers", section_layers)]
    if args.reference:
        sections += [("graphs", lambda: section_graphs(args.reference)), ("est_pose", lambda: section_est_pose(args.reference))]
    else:
        out["skipped"]["graphs"] = out["skipped"]["est_pose"] = "no --reference checkout given"
    for name, fn in sections:
        try:
            out[name] = fn()
            print("section %-9s ok" % name)
        except ImportError as e:
            out["skipped"][name] = "ImportError: %s" % (e,)
            print("section %-9s skipped (%s)" % (name, e))
    with open(args.out, "w") as f:
        json.dump(out, f)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
