"""Vectors produced by the REAL third-party libraries (tools/make_external_vectors.py, run by anyone who has TensorFlow 1.x /
Keras 2.2, OpenCV 3.4 and scikit-image -- none is installable in the build container).  When
tests/golden/external_vectors.json is present these tests hold the oracle (CPU) and the HIP path (GPU) to it, which closes the
"library semantics unpinned" gap of DESIGN.md section 4; when it is absent they are skipped.  A self-check runs the
consumer code on a file fabricated from the oracle itself, so that the plumbing is known to work the day a real file arrives."""
import json
import os
import sys
import types
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
FN = os.path.join(ROOT, "tests", "golden", "external_vectors.json")
EXT = json.load(open(FN)) if os.path.exists(FN) else None
needs_file = pytest.mark.skipif(EXT is None, reason="tests/golden/external_vectors.json not generated (needs TF/Keras/cv2/skimage)")


def _crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


# ------------------------------------------------------------------------------------------ consumers
def check_resize(sec):
    """float32 images (recognition.py:134 prob, :144 img_pred): scikit-image >= 0.16 keeps them float32 through the warp, and the oracle
    reproduces the result BIT FOR BIT (crc of the float32 array, of its uint8 truncation and of the < 0.2 decision).  float64 / bool
    images: the affine matrix comes out of a least-squares fit (LAPACK: ~1e-16 relative noise that differs between installs), so those
    are held to 1e-12 -- and their uint8 / threshold decisions to exact equality."""
    import make_external_vectors as M
    from oracle.est_pose_oracle import resize_bilinear
    aa = bool(sec["anti_aliasing_default"])
    n_f32 = 0
    for i, c in enumerate(sec["cases"]):
        a = M.resize_input(i, c["n_in"], c["dtype"], c["channels"])
        raw = resize_bilinear(a, (c["n_out"], c["n_out"]), c["mode"], c["cval"], anti_aliasing=aa)
        r = np.asarray(raw, np.float64)
        r2 = r.reshape(c["n_out"], c["n_out"], -1)[:, :, 0]
        assert abs(float(r.sum()) - c["sum"]) < 1e-9 * max(1.0, abs(c["sum"])), (i, c)
        assert np.abs(r2[np.arange(c["n_out"]), np.arange(c["n_out"])] - np.array(c["diag"])).max() < 1e-12, (i, c)
        assert np.abs(r2[0] - np.array(c["first_row"])).max() < 1e-12, (i, c)
        assert abs(float(r.min()) - c["min"]) < 1e-12 and abs(float(r.max()) - c["max"]) < 1e-12, (i, c)
        if "crc" not in c:
            continue                                       # a file written before round 4
        assert str(raw.dtype) == c["out_dtype"], (i, c["dtype"], raw.dtype, c["out_dtype"])
        assert _crc((raw * 255).astype(np.uint8)) == c["u8_crc"], (i, "uint8 truncation differs")
        assert _crc(np.packbits(raw < 0.2)) == c["lt02_crc"], (i, "< 0.2 decisions differ")
        if c["out_dtype"] == "float32":
            assert _crc(raw) == c["crc"], (i, "float32 result is not bit-identical to scikit-image's")
            n_f32 += 1
    return n_f32


def check_pnp(sec, solver):
    """solver(P, uv, K) -> ok, R, t, inlier index array"""
    import make_external_vectors as M
    from pix2pose_amd import synthetic
    for (P, uv), ref in zip(M.pnp_problems(seed0=sec["seed0"]), sec["problems"]):
        ok, R, t, inl = solver(P, uv, synthetic.LM_K)
        assert bool(ok) == ref["ok"]
        if not ref["ok"]:
            continue
        assert len(inl) == ref["n_inliers"] and _crc(np.asarray(inl, np.int32).reshape(-1)) == ref["inliers_crc"]
        dt, dr = synthetic.pose_error(np.array(ref["R"]), np.array(ref["t"]), R, t)
        assert dt < 1e-3 and dr < 1e-3, (dt, dr)          # north_star: 1 mm / 1 deg; same hypothesis => ~1e-9


def check_layers(sec):
    from oracle import ae_oracle as O
    f = lambda k, n: np.array(sec[k][n], np.float32)
    tol = 2e-5                                            # TF accumulates in fp32, the oracle in double
    c = "conv_same_s2"
    assert np.abs(O.conv2d(f(c, "x"), f(c, "kernel"), f(c, "bias"), stride=2, padding="same") - f(c, "y")).max() < tol
    c = "zeropad3_conv7_valid_s2"
    xp = np.pad(f(c, "x"), ((0, 0), (3, 3), (3, 3), (0, 0)))
    assert np.abs(O.conv2d(xp, f(c, "kernel"), f(c, "bias"), stride=2, padding="valid") - f(c, "y")).max() < tol
    c = "deconv_same_s2"
    assert np.abs(O.conv2d_transpose(f(c, "x"), f(c, "kernel"), f(c, "bias"), stride=2) - f(c, "y")).max() < tol
    c = "batchnorm_leaky"
    w = {"l." + k: f(c, k) for k in ("gamma", "beta", "mean", "var")}
    assert np.abs(O.bn_act(f(c, "x"), w, "l", "leaky") - f(c, "y")).max() < tol
    c = "maxpool_3x3_s2_same"
    assert np.array_equal(O.maxpool_3x3_s2_same(f(c, "x")), f(c, "y"))
    c = "flatten_dense"
    x = f(c, "x")
    assert np.abs(O.dense(x.reshape(x.shape[0], -1), f(c, "kernel"), f(c, "bias")) - f(c, "y")).max() < tol


def check_graphs(sec, predict):
    """predict(weights, x, backbone) -> decode, prob"""
    from pix2pose_amd import weights as W
    x = (np.random.RandomState(0).randint(0, 256, (2, 128, 128, 3)).astype(np.float32) - 128) / 128
    for backbone, rec in sec.items():
        d, p = predict(W.synthetic_weights(backbone, rec["weights_seed"]), x, backbone)
        idx = np.array(rec["pixel_index"])
        assert np.abs(d.reshape(-1, 3)[idx] - np.array(rec["decode"])).max() < 1e-3           # north_star: XYZ within 1e-3 abs of the TF path
        assert np.abs(p.reshape(-1)[idx] - np.array(rec["prob"])).max() < 1e-3
        for kind in ("weights", "model"):
            assert rec["hdf5_%s_roundtrip_exact" % kind] is True, (backbone, kind, rec["hdf5_%s_roundtrip_exact" % kind])


def check_est_pose(sec, aa, run):
    """run(scene, anti_aliasing) -> list of (ok, R, t, frac, bbox_t, mask, img_pred)"""
    from pix2pose_amd import synthetic
    n = 0
    for s in sec["scenes"]:
        sp = s["spec"]
        sc = synthetic.make_scene(sp["n_det"], seed=sp["seed"], bbox_side=tuple(sp["bbox_side"]))
        for got, ref in zip(run(sc, aa), s["dets"]):
            ok, R, t, frac, box, mask, img = got
            assert ok == ref["ok"] and [int(v) for v in box] == ref["bbox_t"]
            if not ok:
                continue
            n += 1
            dt, dr = synthetic.pose_error(np.array(ref["R"]), np.array(ref["t"]), R, t)
            assert dt < 1.0 and dr < 1.0                  # north_star bar; identical decisions give ~1e-9
            assert int(mask.sum()) == ref["mask_sum"] and _crc(np.packbits(mask)) == ref["mask_crc"]
            assert _crc(img) == ref["img_pred_crc"]
    assert n > 0


def _oracle_est_pose_runner(sc, aa):
    from oracle import est_pose_oracle as E
    out = []
    for i, (img_i, _, bbox, K) in enumerate(sc["dets"]):
        def predict(x, stage, slots=None, i=i):
            m = sc["inject1"][i][None] if stage == 1 else sc["inject2"][i][slots]
            return [m[..., :3].copy(), m[..., 3:].copy()]
        r = E.est_pose(sc["images"][img_i], bbox, predict, K, sc["obj_param"], [0.2, 0.3, 0.35], 0.2, anti_aliasing=aa)
        ok = not (isinstance(r[1], (int, np.integer)) and r[1] == -1)
        out.append((ok, r[2], r[3], r[4], r[5], r[1], r[0]))
    return out


def _oracle_pnp(P, uv, K):
    from oracle import pnp_oracle
    ok, R, t, inl, _ = pnp_oracle.solve_pnp_ransac(P, uv, K)
    return ok, R, t, inl


# ------------------------------------------------------------------------------------------ real file: the oracle
_ALL = ["resize", "pnp", "layers", "graphs", "est_pose"]
_PRESENT = [k for k in _ALL if EXT is not None and k in EXT]


def test_which_real_libraries_the_committed_file_covers():
    """The committed file was written under /opt/conda/bin/python3.9 of the build image: scikit-image 0.18.3 is there, OpenCV and Keras /
    TensorFlow are not -- their sections are absent and SAY so (the tests below exist only for the sections the file holds; a file
    written where those libraries exist brings its tests with it)."""
    assert EXT is not None and "resize" in EXT and EXT["resize"]["version"] == "0.18.3"
    absent = sorted(set(_ALL) - set(_PRESENT))
    assert absent == sorted(EXT.get("skipped", {})), (absent, EXT.get("skipped"))
    assert all("No module named" in v or "no --reference" in v for v in EXT["skipped"].values()), EXT["skipped"]


@needs_file
@pytest.mark.parametrize("section", _PRESENT or ["resize"])
def test_oracle_matches_real_libraries(section):
    if section not in EXT:
        pytest.skip("section skipped by the generator: %s" % EXT.get("skipped", {}).get(section))
    if section == "resize":
        check_resize(EXT["resize"])
    elif section == "pnp":
        check_pnp(EXT["pnp"], _oracle_pnp)
    elif section == "layers":
        check_layers(EXT["layers"])
    elif section == "graphs":
        from oracle import ae_oracle
        check_graphs(EXT["graphs"], lambda w, x, b: ae_oracle.forward(w, x, b))
    else:
        aa = bool(EXT["resize"]["anti_aliasing_default"]) if "resize" in EXT else False
        check_est_pose(EXT["est_pose"], aa, _oracle_est_pose_runner)


# ------------------------------------------------------------------------------------------ real file: the HIP path
def _hip_path_matches_real_libraries(section):
    if section not in EXT:
        pytest.skip("section skipped by the generator")
    from pix2pose_amd import runtime
    if section == "pnp":
        def solver(P, uv, K):
            ok, R, t, info, masks = runtime.pnp_ransac_batch(runtime.default_context(), [K], [P], [uv], want_mask=True)
            return ok[0], R[0], t[0], np.nonzero(masks[0])[0]
        check_pnp(EXT["pnp"], solver)
    elif section == "graphs":
        check_graphs(EXT["graphs"], lambda w, x, b: runtime.Generator(w, b).predict(x))
    else:
        import torch
        from pix2pose_amd import synthetic, weights as W
        aa = bool(EXT["resize"]["anti_aliasing_default"]) if "resize" in EXT else False
        ctx = runtime.Context(0, max_batch=16)
        spec = runtime.ObjectSpec(runtime.Generator(W.synthetic_weights("paper", 1), "paper", ctx), synthetic.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2)

        def run(sc, aa):
            j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
            torch.cuda.synchronize()
            poses, ex = runtime.est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=j1.data_ptr(), inject2=j2.data_ptr(),
                                               inject_slots=3, want_masks=True, anti_aliasing=aa)
            H, Wd = sc["images"].shape[1:3]
            out = []
            for i, p in enumerate(poses):
                v1, v2, u1, u2 = p.bbox_t
                out.append((p.status == 0, np.array(p.R).reshape(3, 3), np.array(p.t), p.frac_inlier, list(p.bbox_t),
                            ex["valid_mask"][i][:H * Wd].reshape(H, Wd).astype(bool),
                            ex["img_pred"][i][:max(v2 - v1, 0) * max(u2 - u1, 0) * 3].reshape(max(v2 - v1, 0), max(u2 - u1, 0), 3)))
            return out
        check_est_pose(EXT["est_pose"], aa, run)


# the HIP path against the sections the file holds (none of pnp / graphs / est_pose in the committed file: no cv2, no Keras in the image)
_GPU_SECTIONS = [k for k in ("pnp", "graphs", "est_pose") if k in _PRESENT]
if _GPU_SECTIONS:
    test_hip_path_matches_real_libraries = pytest.mark.gpu(pytest.mark.parametrize("section", _GPU_SECTIONS)(_hip_path_matches_real_libraries))


@needs_file
@pytest.mark.gpu
def test_hip_back_resize_matches_real_skimage():
    """The HIP back-resize code (cand_pixel + the anti-aliasing filter, reached through the test hook p2p_debug_back_resize) on the inputs of
    the real scikit-image 0.18.3 vectors: float32 maps through the float32 warp -- `(resize(x) * 255)` truncated to uint8 for img_pred-like
    inputs (cval 0.5), `resize(x) < 0.2` for prob-like ones (cval 1) -- and 0/1 masks through the float64 warp (`> 0.9`).  Bit for bit
    (CRC of all pixels)."""
    import ctypes as C
    import make_external_vectors as M
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Context
    sec = EXT["resize"]
    assert sec["anti_aliasing_default"] and sec["version"].startswith("0.18")
    ctx = Context(0, max_batch=8)
    L = _lib.lib()
    n = 0
    for i, c in enumerate(sec["cases"]):
        if c["n_in"] != 128 or c["mode"] != "constant" or "gt09_crc" not in c:
            continue
        a = M.resize_input(i, c["n_in"], c["dtype"], c["channels"])
        no = c["n_out"]
        prob = np.full((128, 128), 0.5, np.float32)
        pred = np.full((128, 128, 3), 0.5, np.float32)
        ng = np.ones((128, 128), np.float32)
        kind = None
        if c["dtype"] == "float32" and c["cval"] == 1.0 and not c["channels"]:
            prob, kind = a, "prob"
        elif c["dtype"] == "float32" and c["cval"] == 0.5:
            pred, kind = (a if c["channels"] else np.repeat(a[:, :, None], 3, axis=2)), "pred"
        elif c["cval"] == 0.0 and (c["dtype"] == "float64" or no >= 128) and not c["channels"] and no % 10:      # (sides that are multiples of 10: the
            # real library's own matrix noise decides `> 0.9` on rows whose weight is exactly 0.9 -- DESIGN.md section 4)
            ng, kind = a.astype(np.float32), "mask"          # a bool image is not filtered by 0.18 (no >= 128: nothing is), a float64 one is
        if kind is None:
            continue
        prob, pred, ng = (np.ascontiguousarray(v, np.float32) for v in (prob, pred, ng))
        q = np.zeros((no, no, 3), np.uint8)
        below = np.zeros((no, no), np.uint8)
        g = np.zeros((no, no), np.uint8)
        _lib.check(L.p2p_debug_back_resize(ctx.handle, prob.ctypes.data, pred.ctypes.data, ng.ctypes.data, no, no, 0.2, 1,
                                           q.ctypes.data, below.ctypes.data, g.ctypes.data), "p2p_debug_back_resize")
        if kind == "prob":
            assert _crc(np.packbits(below.astype(bool))) == c["lt02_crc"], (i, c["n_out"])
        elif kind == "pred":
            assert _crc(q if c["channels"] else q[:, :, 0]) == c["u8_crc"], (i, c["n_out"])
        else:
            assert _crc(np.packbits(g.astype(bool))) == c["gt09_crc"], (i, c["n_out"])
        n += 1
    assert n >= 14, n


# ------------------------------------------------------------------------------------------ self-check of the plumbing
def test_consumers_work_on_a_fabricated_file(monkeypatch):
    """tools/make_external_vectors.py run with stand-in skimage / cv2 modules served by the oracle: the generator's sections and
    the consumers above execute end to end (this says nothing about the real libraries -- that is what the real file is for)."""
    import make_external_vectors as M
    from oracle import est_pose_oracle, pnp_oracle
    sk, skt = types.ModuleType("skimage"), types.ModuleType("skimage.transform")
    sk.__version__ = "0.16.2"
    skt.resize = lambda img, shape, order=1, mode="reflect", cval=0: est_pose_oracle.resize_bilinear(np.asarray(img), tuple(shape), mode, cval, anti_aliasing=True)
    sk.transform = skt
    cv2 = types.ModuleType("cv2")
    cv2.__version__, cv2.SOLVEPNP_EPNP = "3.4.2-standin", 1

    class _Rvec:
        def __init__(self, R):
            self.R = R

        def reshape(self, *a):
            return np.zeros(3)

    def solve(obj, img, K, dist, flags=None, reprojectionError=8.0, iterationsCount=100):
        ok, R, t, inl, _ = pnp_oracle.solve_pnp_ransac(obj, np.asarray(img).reshape(-1, 2), K, iterations=iterationsCount, reproj_err=reprojectionError)
        return (True, _Rvec(R), t.reshape(3, 1), inl.reshape(-1, 1)) if ok else (False, None, None, None)

    def rodrigues(rvec, dst=None):
        dst[:] = rvec.R
        return dst, None
    cv2.solvePnPRansac, cv2.Rodrigues = solve, rodrigues
    for k, v in {"skimage": sk, "skimage.transform": skt, "cv2": cv2}.items():
        monkeypatch.setitem(sys.modules, k, v)
    r = M.section_resize()
    assert r["anti_aliasing_default"] is True and len(r["cases"]) == len(M.RESIZE_CASES)
    check_resize(r)
    r["anti_aliasing_default"] = False                     # the consumer really looks at the flag
    with pytest.raises(AssertionError):
        check_resize(r)
    check_pnp(M.section_pnp(), _oracle_pnp)
    assert M.version_tuple("0.14.2") < [0, 15, 0] <= M.version_tuple("0.15.0") and M.version_tuple("0.19.0rc1") >= [0, 15, 0]
