"""From a BOP-format dataset directory and a file of pre-dumped 2D detections to the detection stream
``eval_bop.run()`` consumes -- the data formats on the input side of the hot path (SURVEY.md section 8f-1,
BASELINE.json configs[4]: "YCB-V full BOP test_targets stream, Mask-RCNN boxes pre-dumped").

Directory conventions are the reference's (nothing here reads a model file or renders anything):
  * dataset sub-directory and test split             tools/bop_io.py:51-113  (``tless`` -> test_primesense, everything else -> test)
  * global camera: ``camera.json``, ycbv ``camera_uw.json`` -> im_size = (width, height)       tools/bop_io.py:115-120
  * object list: the ids of ``models*/models_info.json`` that have an ``obj_<id:06d>.ply`` next to it (all ids when the
    directory holds no mesh at all: evaluation never reads one), narrowed by cfg ``target_obj``    tools/bop_io.py:122-135,
                                                                                                tools/5_evaluation_bop_basic.py:139-153
  * the script's ``bop_dir`` is the DATASET directory ``<cfg dataset_dir>/<dataset>`` (first value bop_io.get_dataset returns,
    tools/5_evaluation_bop_basic.py:122-125), and everything below hangs off it:
  * normalisation factors ``<bop_dir>/models_xyz/<cfg norm_factor_fn>``                        tools/5_evaluation_bop_basic.py:129
  * weights ``<bop_dir>/pix2pose_weights/<id:02d>/inference[_resnet_model|_resnet50].hdf5``    tools/5_evaluation_bop_basic.py:196-205
    (here: the ``.npz`` the converter wrote next to it, or the ``.hdf5`` itself when h5py is installed)
  * targets ``<bop_dir>/<cfg test_target>.json``                                               tools/5_evaluation_bop_basic.py:226-227
  * per scene ``<test_dir>/<scene:06d>/scene_camera.json`` -> cam_K of every image             tools/5_evaluation_bop_basic.py:246-254
  * frames ``rgb/<im:06d>.png``; itodd ``gray/<im:06d>.tif`` copied to three channels          tools/5_evaluation_bop_basic.py:256-266

Detections: the COCO-style list the BOP challenge distributes for its default detectors (and the reference's Mask R-CNN /
RetinaNet wrappers reduce to, tools/5_evaluation_bop_basic.py:35-95): one record per detection,
``{"scene_id", "image_id", "category_id", "bbox": [x, y, w, h], "score", "segmentation": {"size": [h, w], "counts": ...}}``
-- ``rois`` are the reference's ``[v1, u1, v2, u2]`` integers; the run-length masks (plain count lists or COCO's compressed
strings, column-major like pycocotools) are only needed for score_type 2 and decoded when a chunk is prepared.
"""
from __future__ import annotations

import json
import os

import numpy as np

_TEST_SPLIT = {"tless": "test_primesense"}


def dataset_dirs(bop_dir: str, dataset: str):
    """-> (dataset directory, test split directory)   (tools/bop_io.py:51-113)"""
    d = os.path.join(bop_dir, dataset)
    return d, os.path.join(d, _TEST_SPLIT.get(dataset, "test"))


def _models_dir(dataset_dir: str, dataset: str) -> str:
    # evaluation never reads a mesh; the id list comes from whichever models_info.json the reference would open
    for name in (("models_reconst", "models_cad", "models") if dataset == "tless" else ("models",)):
        if os.path.exists(os.path.join(dataset_dir, name, "models_info.json")):
            return os.path.join(dataset_dir, name)
    raise FileNotFoundError("no models*/models_info.json under %s" % dataset_dir)


def load_model_ids(dataset_dir: str, dataset: str, target_obj=None):
    """Sorted object ids of the dataset, narrowed to cfg['target_obj'] when given.  The reference keeps an id only if its mesh
    file exists (tools/bop_io.py:129-131); a deployment without any mesh keeps them all."""
    mdir = _models_dir(dataset_dir, dataset)
    info = json.load(open(os.path.join(mdir, "models_info.json")))
    ids = sorted(int(k) for k in info)
    with_mesh = [i for i in ids if os.path.exists(os.path.join(mdir, "obj_%06d.ply" % i))]
    if with_mesh:
        ids = with_mesh
    if target_obj is not None:
        ids = [i for i in ids if i in set(int(t) for t in target_obj)]
    return ids


def load_im_size(dataset_dir: str, dataset: str):
    cam = json.load(open(os.path.join(dataset_dir, "camera_uw.json" if dataset == "ycbv" else "camera.json")))
    return [int(cam["width"]), int(cam["height"])]


def weights_path(dataset_dir: str, model_id: int, backbone: str) -> str:
    """The converted ``.npz`` next to the reference's inference weights, else the ``.hdf5`` itself (read through
    convert_keras, needs h5py).  Candidate order as in the reference."""
    wdir = os.path.join(dataset_dir, "pix2pose_weights", "%02d" % model_id)
    stems = ["inference_resnet_model", "inference_resnet50"] if backbone == "resnet50" else ["inference"]
    for ext in (".npz", ".hdf5"):
        for s in stems:
            fn = os.path.join(wdir, s + ext)
            if os.path.exists(fn):
                return fn
    raise FileNotFoundError("no inference weights for object %d under %s (expected %s.npz or .hdf5)" % (model_id, wdir, " / ".join(stems)))


# ------------------------------------------------------------------------------------------ run-length masks
def _rle_counts_from_string(s: str):
    """COCO's compressed RLE string -> counts (pycocotools rleFrString: 5 bits per character, continuation bit 0x20,
    sign bit 0x10 of the last character, deltas against the count two places back from the third count on)."""
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(seg) -> np.ndarray:
    """{"size": [h, w], "counts": list | str} -> bool [h, w]; runs alternate 0, 1, 0, ... in column-major order."""
    h, w = (int(v) for v in seg["size"])
    counts = seg["counts"]
    if isinstance(counts, (str, bytes)):
        counts = _rle_counts_from_string(counts.decode() if isinstance(counts, bytes) else counts)
    counts = np.asarray(counts, np.int64)
    if counts.sum() != h * w or (counts < 0).any():
        raise ValueError("run-length mask does not cover %d x %d pixels" % (h, w))
    vals = np.zeros(len(counts), bool)
    vals[1::2] = True
    return np.repeat(vals, counts).reshape(w, h).T


def rle_encode(mask: np.ndarray) -> dict:
    """bool [h, w] -> uncompressed COCO RLE (tests and dump writers)."""
    m = np.asarray(mask, bool)
    flat = m.T.reshape(-1)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {"size": [int(m.shape[0]), int(m.shape[1])], "counts": counts}


# ------------------------------------------------------------------------------------------ detections
def group_detections(detections, model_ids):
    """COCO-style records -> {(scene_id, im_id): {"rois", "obj_ids", "scores", "segmentations"}}.  Detections of objects
    outside `model_ids` are dropped (the reference's detector only knows the dataset's objects); a box is the reference's
    integer [v1, u1, v2, u2] (tools/5_evaluation_bop_basic.py:46-50,85-89)."""
    known = set(model_ids)
    out = {}
    for d in detections:
        oid = int(d["category_id"])
        if oid not in known:
            continue
        x, y, w, h = (float(v) for v in d["bbox"])
        e = out.setdefault((int(d["scene_id"]), int(d["image_id"])), {"rois": [], "obj_ids": [], "scores": [], "segmentations": []})
        e["rois"].append([int(y), int(x), int(y + h), int(x + w)])
        e["obj_ids"].append(oid)
        e["scores"].append(float(d["score"]))
        e["segmentations"].append(d.get("segmentation"))
    return out


def build_dump(cfg: dict, dataset: str, detections) -> dict:
    """The dict eval_bop.run() takes, with absolute paths, from the BOP directory cfg['dataset_dir'] and a detection list."""
    bop_dir = cfg["dataset_dir"]
    dataset_dir, test_dir = dataset_dirs(bop_dir, dataset)
    model_ids = load_model_ids(dataset_dir, dataset, cfg.get("target_obj"))
    backbone = cfg.get("backbone", "paper")
    norm = json.load(open(os.path.join(dataset_dir, "models_xyz", cfg["norm_factor_fn"])))
    targets = json.load(open(os.path.join(dataset_dir, cfg["test_target"] + ".json")))
    per_image = group_detections(detections, model_ids)
    gray = dataset == "itodd"
    images, cams = [], {}
    seen = set()
    for t in targets:
        key = (int(t["scene_id"]), int(t["im_id"]))
        if key in seen:
            continue
        seen.add(key)
        sid, iid = key
        if sid not in cams:
            cams[sid] = json.load(open(os.path.join(test_dir, "%06d" % sid, "scene_camera.json")))
        det = per_image.get(key, {"rois": [], "obj_ids": [], "scores": [], "segmentations": []})
        im = {"scene_id": sid, "im_id": iid,
              "rgb": os.path.join(test_dir, "%06d" % sid, "gray" if gray else "rgb", "%06d.%s" % (iid, "tif" if gray else "png")),
              "cam_K": [float(v) for v in cams[sid][str(iid)]["cam_K"]],
              "rois": det["rois"], "obj_ids": det["obj_ids"], "scores": det["scores"]}
        if any(s is not None for s in det["segmentations"]):
            im["segmentations"] = det["segmentations"]
        images.append(im)
    return {"im_size": load_im_size(dataset_dir, dataset), "model_ids": model_ids,
            "norm_factor": {str(m): norm[str(m)] for m in model_ids},
            "weights": {str(m): weights_path(dataset_dir, m, backbone) for m in model_ids},
            "targets": targets, "images": images}
