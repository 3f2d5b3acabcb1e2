"""Golden vectors from the REFERENCE'S OWN evaluation script (tools/5_evaluation_bop_basic.py + tools/bop_io.py).

    python tests/golden/make_reference_eval_vectors.py      (needs /root/reference; writes reference_eval.json)

The script is EXECUTED unmodified (runpy, argv `[gpu_id] [cfg] [dataset]`) on a tiny synthetic BOP-style dataset in a
temporary directory.  Everything it pulls in that does not exist in this image is stood in:
    bop_toolkit_lib.inout   load_json / load_cam_params / load_scene_camera / load_im (returns an image that encodes
                            (scene_id, im_id)) / save_bop_results (captures the rows the script hands over)
    mrcnn.* , tools.mask_rcnn_util   a detector whose `detect()` returns pre-drawn boxes / classes / scores / masks per image
    pix2pose_model.recognition       a `pix2pose` class whose est_pose() returns pre-drawn (R, t, frac_inlier, mask) per box
    tensorflow, cv2, skimage, matplotlib, transforms3d, pix2pose_util.*, pix2pose_model.ae_model      empty modules
So the fixture pins this repository's restatement of the script's OWN logic (pix2pose_amd/eval_bop.py, SURVEY.md
section 8 row f-1): per-object threshold selection, target grouping, candidate limiting, both score types, the
per-image normalise / sort / ViVo truncation (including the string-vs-int `task_type` quirk) and the output file name.
The CSV byte format belongs to bop_toolkit (un-vendored) and stays unpinned.
"""
import json
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
H, W = 120, 160
MODEL_IDS = [1, 5, 6, 8]            # object ids with a .ply in the fake dataset (5 has none in the targets of scene 3)


def make_world(seed=11):
    """Targets, per-image detections and per-box poses: plain data, stored in the fixture."""
    rs = np.random.RandomState(seed)
    targets, images = [], {}
    for scene in (2, 3):
        for im in (4, 9, 17):
            objs = sorted(rs.choice(MODEL_IDS, int(rs.randint(1, 4)), replace=False).tolist())
            for o in objs:
                targets.append({"scene_id": scene, "im_id": im, "obj_id": int(o), "inst_count": int(rs.randint(1, 3))})
            dets = []
            for _ in range(int(rs.randint(3, 9))):
                v0, u0 = int(rs.randint(0, H - 30)), int(rs.randint(0, W - 30))
                roi = [v0, u0, v0 + int(rs.randint(10, 30)), u0 + int(rs.randint(10, 30))]
                if rs.rand() < 0.1:
                    roi = [-1, -1, -1, -1]
                cls = int(rs.randint(0, len(MODEL_IDS)))                       # index into the sorted model ids
                mv0, mu0 = max(roi[0] + int(rs.randint(-4, 5)), 0), max(roi[1] + int(rs.randint(-4, 5)), 0)
                det_mask = [mv0, mu0, mv0 + int(rs.randint(8, 30)), mu0 + int(rs.randint(8, 30))]
                fail = bool(rs.rand() < 0.15)
                q = rs.randn(4); q /= np.linalg.norm(q)
                w_, x_, y_, z_ = q
                R = [[1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_)],
                     [2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_)],
                     [2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_ * x_ + y_ * y_)]]
                pv0, pu0 = max(roi[0] + int(rs.randint(-3, 4)), 0), max(roi[1] + int(rs.randint(-3, 4)), 0)
                dets.append({"roi": roi, "class_id": cls + 1, "score": float(np.round(rs.uniform(0.3, 1.0), 6)), "det_mask": det_mask,
                             "pose": {"fail": fail, "R": np.round(R, 9).tolist(), "t": np.round(rs.uniform(-300, 900, 3), 6).tolist(),
                                      "frac_inlier": float(np.round(rs.uniform(0.05, 0.9), 6)),
                                      "pred_mask": [pv0, pu0, pv0 + int(rs.randint(6, 28)), pu0 + int(rs.randint(6, 28))]}})
            images["%d/%d" % (scene, im)] = dets
    return targets, images


def rect_mask(r):
    m = np.zeros((H, W), bool)
    m[max(r[0], 0):max(r[2], 0), max(r[1], 0):max(r[3], 0)] = True
    return m


def run_reference(cfg_extra, targets, images, tmp):
    captured = {"ctor": [], "rows": None, "path": None}
    bop = os.path.join(tmp, "bop")
    ds = os.path.join(bop, "lmo")
    for d in ("models", "models_xyz", "test"):
        os.makedirs(os.path.join(ds, d), exist_ok=True)
    json.dump({str(m): {"diameter": 100.0} for m in MODEL_IDS}, open(os.path.join(ds, "models", "models_info.json"), "w"))
    for m in MODEL_IDS:
        open(os.path.join(ds, "models", "obj_%06d.ply" % m), "w").close()
    json.dump({str(m): {"x_scale": 30.0 + m, "y_scale": 31.0 + m, "z_scale": 32.0 + m, "x_ct": 0.5 * m, "y_ct": -0.25 * m, "z_ct": 1.0}
               for m in MODEL_IDS}, open(os.path.join(ds, "models_xyz", "norm_factor.json"), "w"))
    json.dump(targets, open(os.path.join(ds, "targets.json"), "w"))
    cfg = {"backbone": "resnet50", "dataset_dir": bop, "detection_pipeline": "rcnn", "path_to_detection_pipeline": os.path.join(tmp, "mrcnn_dir"),
           "path_to_output": os.path.join(tmp, "out"), "outlier_th": [0.2, 0.3, 0.35], "inlier_th": 0.2, "norm_factor_fn": "norm_factor.json",
           "score_type": 2, "task_type": 2, "cand_factor": 2, "test_target": "targets"}
    cfg.update(cfg_extra)
    cfg_fn = os.path.join(tmp, "cfg.json")
    json.dump(cfg, open(cfg_fn, "w"))

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float

    def load_im(path):
        parts = path.replace("\\\\", "/").split("/")
        scene, im = int(parts[-3]), int(os.path.splitext(parts[-1])[0])
        img = np.zeros((H, W, 3), np.uint8)
        img[0, 0] = (77, scene, im)
        return img
    K = [572.4, 0, 80.0, 0, 573.6, 60.0, 0, 0, 1]
    inout = mod("bop_toolkit_lib.inout", load_json=lambda p: json.load(open(p)),
                load_cam_params=lambda p: {"im_size": (W, H), "K": np.array(K).reshape(3, 3)},
                load_scene_camera=lambda p: {im: {"cam_K": np.array(K) + 0.001 * im, "depth_scale": 1.0} for im in (4, 9, 17)},
                load_im=load_im, save_bop_results=lambda path, rows: captured.update(rows=rows, path=path))
    mod("bop_toolkit_lib", inout=inout, renderer=mod("bop_toolkit_lib.renderer"))
    for name in ("cv2", "matplotlib", "transforms3d", "pix2pose_util", "pix2pose_util.data_io", "pix2pose_model.ae_model", "mrcnn", "mrcnn.config"):
        mod(name)
    mod("matplotlib.pyplot")
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    mod("skimage")
    mod("skimage.transform", resize=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("resize is not expected: masks already have the frame size")))
    mod("pix2pose_util.common_util", get_bbox_from_mask=None)
    gpu_options = types.SimpleNamespace(allow_growth=False)
    mod("tensorflow", ConfigProto=lambda: types.SimpleNamespace(gpu_options=gpu_options), Session=lambda config=None: None)
    sys.modules["mrcnn.config"].Config = object

    def key_of(image):
        return None if image[0, 0, 0] != 77 else "%d/%d" % (int(image[0, 0, 1]), int(image[0, 0, 2]))

    class MaskRCNN:
        def __init__(self, mode=None, config=None, model_dir=None):
            pass

        def find_last(self):
            return ""

        def load_weights(self, *a, **k):
            pass

        def detect(self, imgs, verbose=0):
            k = key_of(imgs[0])
            dets = images.get(k, []) if k else []
            return [{"rois": np.array([d["roi"] for d in dets], np.int64).reshape(-1, 4), "class_ids": np.array([d["class_id"] for d in dets], np.int64),
                     "scores": np.array([d["score"] for d in dets]),
                     "masks": np.stack([rect_mask(d["det_mask"]) for d in dets], -1) if dets else np.zeros((H, W, 0), bool)}]
    mod("mrcnn.utils", resize_image=lambda img, **k: (img, (0, 0, img.shape[0], img.shape[1]), 1, None, None))
    mod("mrcnn.model", MaskRCNN=MaskRCNN)

    class BopInferenceConfig:
        IMAGE_MIN_DIM = IMAGE_MAX_DIM = 0
        IMAGE_MIN_SCALE = 0
        IMAGE_RESIZE_MODE = "none"

        def __init__(self, dataset=None, num_classes=None, im_width=None, im_height=None):
            pass

        def display(self):
            pass

    class pix2pose:
        def __init__(self, weight_fn, camK, res_x, res_y, obj_param, th_ransac=3.0, th_outlier=None, th_inlier=0.1, backbone="paper", **kw):
            self.camK = camK
            captured["ctor"].append({"weight_fn": os.path.relpath(weight_fn, bop), "obj_param": np.asarray(obj_param).tolist(),
                                     "th_outlier": [float(v) for v in th_outlier], "th_inlier": th_inlier, "th_ransac": th_ransac, "backbone": backbone,
                                     "res": [res_x, res_y]})

        def est_pose(self, image, roi):
            k = key_of(image)
            for d in (images.get(k, []) if k else []):
                if list(d["roi"]) == [int(v) for v in roi]:
                    p = d["pose"]
                    if p["fail"]:
                        break
                    return np.zeros(1), rect_mask(p["pred_mask"]), np.array(p["R"]), np.array(p["t"]), p["frac_inlier"], np.array(roi)
            return np.zeros(1), -1, -1, -1, -1, np.array(roi)
    pm = mod("pix2pose_model")
    pm.__path__ = []
    mod("pix2pose_model.recognition", pix2pose=pix2pose)
    tools_pkg = mod("tools")
    tools_pkg.__path__ = [os.path.join(REF, "tools")]
    mod("tools.mask_rcnn_util", BopInferenceConfig=BopInferenceConfig)

    argv, cwd = sys.argv, os.getcwd()
    sys.argv = ["5_evaluation_bop_basic.py", "-1", cfg_fn, "lmo"]
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.modules.pop("tools.bop_io", None)
    try:
        runpy.run_path(os.path.join(REF, "tools", "5_evaluation_bop_basic.py"), run_name="__main__")
    finally:
        sys.argv = argv
        os.chdir(cwd)
        sys.path.remove(REF)
    rows = [{"scene_id": int(r["scene_id"]), "im_id": int(r["im_id"]), "obj_id": int(r["obj_id"]), "score": float(r["score"]),
             "R": np.asarray(r["R"]).tolist(), "t": np.asarray(r["t"]).tolist()} for r in captured["rows"]]
    return {"cfg": cfg_extra, "rows": rows, "output_name": os.path.basename(captured["path"]), "ctor": captured["ctor"]}


def main():
    targets, images = make_world()
    out = {"note": "rows handed to inout.save_bop_results by /root/reference/tools/5_evaluation_bop_basic.py run on a synthetic dataset with the "
                   "detector, the pose estimator and bop_toolkit stood in, see tests/golden/make_reference_eval_vectors.py",
           "frame": [H, W], "model_ids": MODEL_IDS, "targets": targets, "images": images, "runs": []}
    for extra in ({"score_type": 2, "task_type": 2}, {"score_type": 2, "task_type": "2"}, {"score_type": 1, "task_type": "2", "cand_factor": 1},
                  {"score_type": 1, "task_type": 1, "outlier_th": [[0.15], [0.25], [0.3], [0.35]], "cand_factor": 0.5}):
        tmp = tempfile.mkdtemp()
        try:
            out["runs"].append(run_reference(extra, targets, images, tmp))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        print(extra, "->", len(out["runs"][-1]["rows"]), "rows,", out["runs"][-1]["output_name"])
    fn = os.path.join(HERE, "reference_eval.json")
    with open(fn, "w") as f:
        json.dump(out, f)
    print("wrote", fn, os.path.getsize(fn), "bytes")


if __name__ == "__main__":
    main()
