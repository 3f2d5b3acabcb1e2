"""Harness counterpart of tools/5_evaluation_bop_basic.py: CPU tests of the per-image
logic and CSV format; a GPU end-to-end run on a synthetic pre-dumped detection stream."""
import json
import os

import numpy as np
import pytest

from pix2pose_amd import eval_bop as E


def test_threshold_selection_and_target_grouping():
    assert E.outlier_thresholds({"outlier_th": [0.2, 0.3, 0.35]}, 2) == [[0.2, 0.3, 0.35]] * 2
    assert E.outlier_thresholds({"outlier_th": [[0.1], [0.3], [0.2]]}, 3) == [[0.1], [0.3], [0.2]]   # cfg_tless_paper style
    tg = [{"scene_id": 1, "im_id": 3, "obj_id": 5, "inst_count": 1}, {"scene_id": 1, "im_id": 3, "obj_id": 8, "inst_count": 2},
          {"scene_id": 1, "im_id": 4, "obj_id": 5, "inst_count": 1}, {"scene_id": 2, "im_id": 4, "obj_id": 9, "inst_count": 1}]
    assert E.group_targets(tg) == [[1, 3, [5, 8], [1, 2]], [1, 4, [5], [1]], [2, 4, [9], [1]]]


def test_candidate_limiting_like_reference():
    rois = [[0, 0, 5, 5], [-1, -1, 3, 3], [1, 1, 6, 6], [2, 2, 7, 7], [3, 3, 8, 8], [4, 4, 9, 9]]
    obj_ids = [5, 5, 7, 5, 5, 5]
    # inst_count 1, cand_factor 2: candidates are taken while pred <= 2, i.e. three of object 5
    assert E.select_detections(rois, obj_ids, [5], [1], 2.0) == [0, 3, 4]
    assert E.select_detections(rois, obj_ids, [5, 7], [1, 1], 0.0) == [0, 2]


def test_scores_ranking_and_vivo_quirk():
    assert E.detection_score(0.9, 0.5, (30, 60), 2) == 0.9 * 0.5 * 0.5 * 60
    assert E.detection_score(0.9, 0.5, (0, 0), 2) == 0
    assert E.detection_score(0.9, 0.5, None, 1) == 0.9
    res = [{"obj_id": 5, "score": 2.0, "R": np.eye(3), "t": np.zeros(3)}, {"obj_id": 5, "score": 4.0, "R": np.eye(3), "t": np.ones(3)},
           {"obj_id": 7, "score": 1.0, "R": np.eye(3), "t": np.ones(3) * 2}]
    rows = E.rank_image_results(res, [5, 7], [1, 1], 2, 11, 22, 0.5)        # int 2: the reference's `=='2'` never fires
    assert [r["score"] for r in rows] == [1.0, 0.5, 0.25] and [r["obj_id"] for r in rows] == [5, 5, 7]
    rows = E.rank_image_results(res, [5, 7], [1, 1], '2', 11, 22, 0.5)      # string '2': ViVo truncation
    assert [(r["obj_id"], r["score"]) for r in rows] == [(5, 1.0), (7, 0.25)]
    assert E.rank_image_results([], [5], [1], 2, 1, 1, 0.0) == []


def test_csv_format(tmp_path):
    rows = [{"scene_id": 1, "im_id": 2, "obj_id": 3, "score": 0.5, "R": np.arange(9.0).reshape(3, 3), "t": np.array([1.5, -2.0, 700.25]), "time": 0.25}]
    fn = str(tmp_path / E.output_name("lmo"))
    E.save_bop_results(fn, rows)
    lines = open(fn).read().split("\n")
    assert lines[0] == "scene_id,im_id,obj_id,score,R,t,time"
    assert lines[1] == "1,2,3,0.5,0.0 1.0 2.0 3.0 4.0 5.0 6.0 7.0 8.0,1.5 -2.0 700.25,0.25"
    assert E.output_name("tless").endswith("tless-test-primesense.csv") and fn.endswith("pix2pose-iccv19_lmo-test.csv")


@pytest.mark.gpu
def test_harness_end_to_end_matches_per_detection_shim(tmp_path):
    """A synthetic detection stream through the batched harness == the reference's flow (one shim
    est_pose per detection + host-side mask IoU), including score_type 2 with detector masks."""
    import torch
    from pix2pose_amd import synthetic as S
    from pix2pose_amd.recognition import pix2pose
    from pix2pose_amd.runtime import Context
    sc = S.make_scene(6, seed=21, n_images=2)
    H, Wd = sc["images"].shape[1:3]
    rs = np.random.RandomState(0)
    images = []
    for fi in range(2):
        np.save(tmp_path / ("f%d.npy" % fi), sc["images"][fi])
        ids = [i for i, d in enumerate(sc["dets"]) if d[0] == fi]
        masks = np.zeros((H, Wd, len(ids)), bool)
        for k, i in enumerate(ids):
            b = sc["dets"][i][2]
            masks[b[0] + 10:b[2] - 5, b[1] + 8:b[3] - 12, k] = True
        np.save(tmp_path / ("m%d.npy" % fi), masks)
        images.append({"scene_id": 1, "im_id": fi, "rgb": "f%d.npy" % fi, "cam_K": S.LM_K.reshape(-1).tolist(),
                       "rois": [sc["dets"][i][2] for i in ids], "obj_ids": [1] * len(ids),
                       "scores": [float(rs.uniform(0.5, 1)) for _ in ids], "masks": "m%d.npy" % fi, "_ids": ids})
    dump = {"im_size": [Wd, H], "model_ids": [1], "weights": {"1": "synthetic:paper:1"},
            "norm_factor": {"1": dict(zip(["x_scale", "y_scale", "z_scale", "x_ct", "y_ct", "z_ct"], S.OBJ_PARAM.tolist()))},
            "targets": [{"scene_id": 1, "im_id": 0, "obj_id": 1, "inst_count": 5}, {"scene_id": 1, "im_id": 1, "obj_id": 1, "inst_count": 5}],
            "images": images}
    cfg = {"backbone": "paper", "outlier_th": [0.2, 0.3, 0.35], "inlier_th": 0.2, "score_type": 2, "task_type": 2,
           "cand_factor": 2, "path_to_output": str(tmp_path / "out"), "generator_chunk": 32}
    # injected decoder maps, in the harness's detection order (image 0 first, then image 1)
    order = [i for im in images for i in im["_ids"]]
    j1 = torch.from_numpy(sc["inject1"][order]).cuda()
    j2 = torch.from_numpy(sc["inject2"][order]).cuda()
    torch.cuda.synchronize()
    rows = E.run(cfg, "lmo", dump, base_dir=str(tmp_path),
                 est_pose_kwargs=dict(inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3))
    assert os.path.exists(tmp_path / "out" / "pix2pose-iccv19_lmo-test.csv")
    assert len(rows) >= 5
    # the reference's flow, one detection at a time through the batch API of size 1 (same injection)
    from pix2pose_amd import runtime
    ctx = Context(0, max_batch=16)
    p = pix2pose("synthetic:paper:1", S.LM_K, Wd, H, S.OBJ_PARAM, th_outlier=[0.2, 0.3, 0.35], th_inlier=0.2, backbone="paper", ctx=ctx)
    exp = {}
    for im in images:
        masks = np.load(tmp_path / im["masks"])
        res = []
        for k, i in enumerate(im["_ids"]):
            a1, a2 = torch.from_numpy(sc["inject1"][i:i + 1]).cuda(), torch.from_numpy(sc["inject2"][i:i + 1]).cuda()
            torch.cuda.synchronize()
            poses, ex = runtime.est_pose_batch(ctx, [p._spec()], [sc["images"][sc["dets"][i][0]]], [(0, 0, sc["dets"][i][2], S.LM_K)],
                                               inject1=a1.data_ptr(), inject2=a2.data_ptr(), inject_slots=3, want_masks=True)
            if poses[0].status != 0:
                continue
            vm = ex["valid_mask"][0][:H * Wd].reshape(H, Wd).astype(bool)
            union = np.sum(np.logical_or(masks[:, :, k], vm))
            iou = 0 if union <= 0 else np.sum(np.logical_and(masks[:, :, k], vm)) / union
            res.append({"obj_id": 1, "score": im["scores"][k] * poses[0].frac_inlier * iou * union,
                        "R": np.array(poses[0].R).reshape(3, 3), "t": np.array(poses[0].t)})
        for r in E.rank_image_results(res, [1], [5], 2, 1, im["im_id"], 0.0):
            exp.setdefault(im["im_id"], []).append(r)
    got = {}
    for r in rows:
        got.setdefault(r["im_id"], []).append(r)
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert len(got[k]) == len(exp[k])
        for a, b in zip(got[k], exp[k]):
            assert abs(a["score"] - b["score"]) < 1e-12
            np.testing.assert_array_equal(a["R"], b["R"])
            np.testing.assert_array_equal(a["t"], b["t"])


@pytest.mark.gpu
def test_harness_stream_many_images_objects_and_cli(tmp_path):
    """BASELINE.json configs[4]'s shape on synthetic data: a BOP test_targets stream -- 24 frames, 5 objects (per-object
    weights and obj_param), 72 detections with detector masks and scores -- goes through the harness in 6 pipelined chunks
    (p2p_est_pose_submit / collect across chunk boundaries, score_type 2 mask sums from the device) and out as the bop19 CSV,
    via the command-line entry point a user would run.  Every row equals what per-image BLOCKING calls give."""
    import subprocess
    import sys
    from pix2pose_amd import runtime, synthetic as S, weights as W
    rs = np.random.RandomState(5)
    n_img, per_img, n_obj = 24, 3, 5
    sc = S.make_scene(n_img * per_img, seed=77, n_images=n_img, bbox_side=(60, 150))
    H, Wd = sc["images"].shape[1:3]
    model_ids = [3, 5, 8, 13, 21]
    images, targets = [], []
    for fi in range(n_img):
        np.save(tmp_path / ("f%02d.npy" % fi), sc["images"][fi])
        ids = list(range(fi * per_img, (fi + 1) * per_img))
        obj = [model_ids[(fi + k) % n_obj] for k in range(per_img)]
        masks = np.zeros((H, Wd, per_img), bool)
        for k, i in enumerate(ids):
            b = sc["dets"][i][2]
            masks[max(b[0], 0) + 4:b[2] - 3, max(b[1], 0) + 5:b[3] - 6, k] = True
        np.save(tmp_path / ("m%02d.npy" % fi), masks)
        images.append({"scene_id": 2, "im_id": fi, "rgb": "f%02d.npy" % fi, "cam_K": S.LM_K.reshape(-1).tolist(),
                       "rois": [[int(v) for v in sc["dets"][i][2]] for i in ids], "obj_ids": obj,
                       "scores": [float(rs.uniform(0.5, 1)) for _ in ids], "masks": "m%02d.npy" % fi})
        for o in sorted(set(obj)):
            targets.append({"scene_id": 2, "im_id": fi, "obj_id": o, "inst_count": 1})
    dump = {"im_size": [Wd, H], "model_ids": model_ids, "weights": {str(m): "synthetic:paper:%d" % m for m in model_ids},
            "norm_factor": {str(m): dict(zip(["x_scale", "y_scale", "z_scale", "x_ct", "y_ct", "z_ct"], (S.OBJ_PARAM * (1 + 0.01 * m)).tolist())) for m in model_ids},
            "targets": targets, "images": images}
    cfg = {"backbone": "paper", "outlier_th": [0.2, 0.3, 0.35], "inlier_th": 0.2, "score_type": 2, "task_type": 2,
           "cand_factor": 2, "path_to_output": str(tmp_path / "out"), "generator_chunk": 64, "dataset_dir": str(tmp_path)}
    json.dump(dump, open(tmp_path / "detections.json", "w"))
    json.dump(cfg, open(tmp_path / "cfg.json", "w"))
    # the command-line entry point (random weights give no usable masks, so this checks the plumbing: it runs, streams its
    # chunks and writes the bop19 file; the row-level checks below inject decoder maps)
    r = subprocess.run([sys.executable, "-m", "pix2pose_amd.eval_bop", "0", str(tmp_path / "cfg.json"), "ycbv", str(tmp_path / "detections.json")],
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    csv = open(tmp_path / "out" / "pix2pose-iccv19_ycbv-test.csv").read().split("\n")
    assert csv[0] == "scene_id,im_id,obj_id,score,R,t,time"
    # stream order of the detections = target order x select_detections order; injected decoder maps follow it
    import torch
    order = []
    for scene_id, im_id, obj_t, inst in E.group_targets(targets):
        im = images[im_id]
        order += [im_id * per_img + k for k in E.select_detections(im["rois"], im["obj_ids"], obj_t, inst, 2.0)]
    j1 = torch.from_numpy(sc["inject1"][order]).cuda()
    j2 = torch.from_numpy(sc["inject2"][order]).cuda()
    torch.cuda.synchronize()
    inject = lambda start, n: dict(inject1=j1[start:start + n].data_ptr(), inject2=j2[start:start + n].data_ptr(), inject_slots=3)
    # the same stream in 6 pipelined chunks, in ONE chunk, and image by image through the blocking call
    cfg_q = dict(cfg, path_to_output=str(tmp_path / "out6"))
    rows6 = E.run(cfg_q, "ycbv", dump, base_dir=str(tmp_path), batch_images=4, est_pose_kwargs=inject)
    rows1 = E.run(dict(cfg, path_to_output=None), "ycbv", dump, base_dir=str(tmp_path), batch_images=64, est_pose_kwargs=inject)
    csv = open(tmp_path / "out6" / "pix2pose-iccv19_ycbv-test.csv").read().split("\n")
    assert len(csv) - 1 == len(rows6) == len(rows1) and len(rows6) >= 40
    ctx = runtime.Context(0, max_batch=64)
    specs = [runtime.ObjectSpec(runtime.Generator(W.load_weights("synthetic:paper:%d" % m, "paper"), "paper", ctx),
                                E.model_params_to_obj_param(dump["norm_factor"][str(m)]), cfg["outlier_th"], cfg["inlier_th"]) for m in model_ids]
    exp, pos = [], 0
    for scene_id, im_id, obj_t, inst in E.group_targets(targets):
        im = images[im_id]
        masks = np.load(tmp_path / im["masks"])
        sel = E.select_detections(im["rois"], im["obj_ids"], obj_t, inst, 2.0)
        dets = [(0, model_ids.index(im["obj_ids"][k]), im["rois"][k], S.LM_K) for k in sel]
        poses, ex = runtime.est_pose_batch(ctx, specs, [sc["images"][im_id]], dets, det_masks=[masks[:, :, k] for k in sel], **inject(pos, len(sel)))
        pos += len(sel)
        res = [{"obj_id": im["obj_ids"][k], "score": E.detection_score(im["scores"][k], p.frac_inlier, (int(ex["mask_stats"][j, 0]), int(ex["mask_stats"][j, 1])), 2, "rcnn"),
                "R": np.array(p.R).reshape(3, 3), "t": np.array(p.t)} for j, (k, p) in enumerate(zip(sel, poses)) if p.status == 0]
        exp += E.rank_image_results(res, obj_t, inst, 2, scene_id, im_id, 0.0)
    for rows in (rows6, rows1):
        assert len(rows) == len(exp)
        for a, b in zip(rows, exp):
            assert (a["scene_id"], a["im_id"], a["obj_id"]) == (b["scene_id"], b["im_id"], b["obj_id"])
            assert abs(a["score"] - b["score"]) < 1e-12
            np.testing.assert_array_equal(a["R"], b["R"])
            np.testing.assert_array_equal(a["t"], b["t"])
    # the CSV holds the same rows (text round trip of the floats)
    for line, b in zip(csv[1:], exp):
        f = line.split(",")
        assert [int(f[0]), int(f[1]), int(f[2])] == [b["scene_id"], b["im_id"], b["obj_id"]]
        assert np.allclose([float(v) for v in f[4].split()], np.asarray(b["R"]).flatten(), rtol=0, atol=1e-12)
        assert np.allclose([float(v) for v in f[5].split()], np.asarray(b["t"]).flatten(), rtol=0, atol=1e-9)
