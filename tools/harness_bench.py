"""Throughput of the evaluation harness on a synthetic BOP directory with PNG frames (run on a GPU box):
python tools/harness_bench.py -- frame decoding on one thread against the prefetching loader (eval_bop.FramePrefetcher)."""
import sys, os, time, json, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch; torch.cuda.init()
from pix2pose_amd import bop_dataset as B, eval_bop as E, synthetic as S
from test_bop_dataset import make_bop_dir
root = tempfile.mkdtemp()
n_sc, per = 8, 96
n_img = n_sc * per
sc = S.make_scene(n_img * 3, seed=5, n_images=8, bbox_side=(70, 140))
frames = [sc["images"][i % 8] for i in range(n_img)]
cfg, targets, paths = make_bop_dir(root, "ycbv", n_sc, per, (1, 4, 9), backbone="resnet50", frames=frames)
cfg["target_obj"] = [1, 4, 9]; cfg["score_type"] = 1; cfg["generator_chunk"] = 512
tl = E.group_targets(targets)
dets = []
for gi, (sid, iid, obj_t, inst) in enumerate(tl):
    for k in range(3):
        b = [int(v) for v in sc["dets"][(gi * 3 + k) % len(sc["dets"])][2]]
        dets.append({"scene_id": sid, "image_id": iid, "category_id": [1, 4, 9][k], "bbox": [b[1], b[0], b[3] - b[1], b[2] - b[0]], "score": 0.9})
dump = B.build_dump(cfg, "ycbv", dets)
for thr in (0, 8, 0, 8):
    c = dict(cfg, path_to_output=None, loader_threads=thr)
    t0 = time.time()
    rows = E.run(c, "ycbv", dump, base_dir="/", batch_images=32)
    dt = time.time() - t0
    nd = sum(len(E.select_detections(im["rois"], im["obj_ids"], t[2], t[3], 2.0)) for im, t in zip(dump["images"], tl))
    print("loader_threads %d: %d images, %d detections, %.2f s -> %.0f images/s, %.0f detections/s" % (thr, n_img, nd, dt, n_img / dt, nd / dt))
