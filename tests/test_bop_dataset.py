"""The data formats on the input side of the harness: BOP dataset directory + COCO-style detections -> detection stream
(pix2pose_amd/bop_dataset.py; reference tools/bop_io.py, tools/5_evaluation_bop_basic.py:120-270), and the harness sharded
over several ranks (BASELINE.json configs[4])."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pix2pose_amd import bop_dataset as B
from pix2pose_amd import eval_bop as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ run-length masks
def _rle_to_string(counts):
    """pycocotools rleToString (maskApi.c), restated to produce test vectors for the decoder."""
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def test_rle_round_trip_and_compressed_strings():
    rs = np.random.RandomState(3)
    for h, w in ((1, 1), (7, 5), (48, 64), (480, 640)):
        for density in (0.0, 0.03, 0.5, 1.0):
            m = rs.uniform(size=(h, w)) < density
            if h > 40:                                   # blobs: long runs exercise multi-character counts and negative deltas
                m = np.zeros((h, w), bool)
                for _ in range(6):
                    y, x = rs.randint(0, h - 8), rs.randint(0, w - 8)
                    m[y:y + rs.randint(1, h // 3), x:x + rs.randint(1, w // 3)] = density > 0
            seg = B.rle_encode(m)
            assert sum(seg["counts"]) == h * w
            np.testing.assert_array_equal(B.rle_decode(seg), m)
            packed = {"size": seg["size"], "counts": _rle_to_string(seg["counts"])}
            np.testing.assert_array_equal(B.rle_decode(packed), m)
            np.testing.assert_array_equal(B.rle_decode(dict(packed, counts=packed["counts"].encode())), m)
    # column-major order, first run = zeros: a 2 x 3 mask with only (row 1, col 0) set is runs 1, 1, 4
    np.testing.assert_array_equal(B.rle_decode({"size": [2, 3], "counts": [1, 1, 4]}), [[0, 0, 0], [1, 0, 0]])
    assert B.rle_encode(np.array([[1, 0], [1, 0]], bool))["counts"] == [0, 2, 2]
    with pytest.raises(ValueError):
        B.rle_decode({"size": [2, 3], "counts": [1, 1, 3]})


# ------------------------------------------------------------------------------------------ a BOP directory on disk
def make_bop_dir(root, dataset="ycbv", n_scenes=2, per_scene=3, model_ids=(1, 4, 9), backbone="paper", frames=None, weights=True):
    """Writes a miniature BOP dataset in the reference's layout; returns (cfg, targets, frame paths by (scene, im))."""
    from PIL import Image
    from pix2pose_amd import synthetic as S, weights as W
    ds = os.path.join(root, dataset)
    split = "test_primesense" if dataset == "tless" else "test"
    os.makedirs(os.path.join(ds, "models"), exist_ok=True)
    json.dump({str(m): {"diameter": 100.0 + m} for m in list(model_ids) + [77]}, open(os.path.join(ds, "models", "models_info.json"), "w"))
    H, Wd = 480, 640
    json.dump({"cx": 312.9869, "cy": 241.3109, "depth_scale": 0.1, "fx": 1066.778, "fy": 1067.487, "height": H, "width": Wd},
              open(os.path.join(ds, "camera_uw.json" if dataset == "ycbv" else "camera.json"), "w"))
    os.makedirs(os.path.join(ds, "models_xyz"), exist_ok=True)          # the script's bop_dir is the dataset directory
    keys = ["x_scale", "y_scale", "z_scale", "x_ct", "y_ct", "z_ct"]
    json.dump({str(m): dict(zip(keys, (S.OBJ_PARAM * (1 + 0.01 * m)).tolist())) for m in list(model_ids) + [77]},
              open(os.path.join(ds, "models_xyz", "norm_factor.json"), "w"))
    if weights:
        for m in model_ids:
            wdir = os.path.join(ds, "pix2pose_weights", "%02d" % m)
            os.makedirs(wdir, exist_ok=True)
            W.save_weights(os.path.join(wdir, ("inference_resnet_model" if backbone == "resnet50" else "inference") + ".npz"), backbone,
                           W.synthetic_weights(backbone, m))
    targets, paths = [], {}
    rs = np.random.RandomState(1)
    k = 0
    for s in range(n_scenes):
        sid = 48 + s
        sdir = os.path.join(ds, split, "%06d" % sid)
        os.makedirs(os.path.join(sdir, "rgb"), exist_ok=True)
        cam = {}
        for i in range(per_scene):
            iid = 10 * i + 1
            K = S.LM_K.copy()
            K[0, 2] += s; K[1, 2] -= i                   # per-image intrinsics, as scene_camera.json allows
            cam[str(iid)] = {"cam_K": K.reshape(-1).tolist(), "depth_scale": 0.1}
            img = frames[k] if frames is not None else rs.randint(0, 255, (H, Wd, 3)).astype(np.uint8)
            fn = os.path.join(sdir, "rgb", "%06d.png" % iid)
            Image.fromarray(img).save(fn)
            paths[(sid, iid)] = fn
            for m in model_ids[: 2 + (k % 2)]:
                targets.append({"im_id": iid, "inst_count": 1, "obj_id": int(m), "scene_id": sid})
            k += 1
        json.dump(cam, open(os.path.join(sdir, "scene_camera.json"), "w"))
    json.dump(targets, open(os.path.join(ds, "test_targets_bop19.json"), "w"))
    cfg = {"dataset_dir": root, "test_target": "test_targets_bop19", "norm_factor_fn": "norm_factor.json", "backbone": backbone,
           "outlier_th": [0.2, 0.3, 0.35], "inlier_th": 0.2, "score_type": 2, "task_type": 2, "cand_factor": 2,
           "path_to_output": os.path.join(root, "out"), "generator_chunk": 64}
    return cfg, targets, paths


def test_build_dump_from_a_bop_directory(tmp_path):
    root = str(tmp_path)
    cfg, targets, paths = make_bop_dir(root, "ycbv")
    mask = np.zeros((480, 640), bool)
    mask[100:180, 200:260] = True
    dets = [
        {"scene_id": 48, "image_id": 1, "category_id": 4, "bbox": [200.7, 100.2, 60.9, 80.4], "score": 0.9, "segmentation": B.rle_encode(mask)},
        {"scene_id": 48, "image_id": 1, "category_id": 1, "bbox": [10, 20, 30, 40], "score": 0.5},
        {"scene_id": 48, "image_id": 1, "category_id": 77, "bbox": [1, 2, 3, 4], "score": 0.99},      # not a target object
        {"scene_id": 49, "image_id": 21, "category_id": 9, "bbox": [300, 50, 100, 120], "score": 0.7},
        {"scene_id": 50, "image_id": 1, "category_id": 1, "bbox": [0, 0, 5, 5], "score": 0.1},         # image not in the target list
    ]
    cfg["target_obj"] = [1, 4, 9]
    dump = B.build_dump(cfg, "ycbv", dets)
    assert dump["im_size"] == [640, 480] and dump["model_ids"] == [1, 4, 9]
    assert dump["targets"] == targets
    assert [(im["scene_id"], im["im_id"]) for im in dump["images"]] == [(t[0], t[1]) for t in E.group_targets(targets)]
    im = dump["images"][0]
    assert im["rgb"] == paths[(48, 1)] and os.path.isabs(im["rgb"])
    assert im["rois"] == [[100, 200, 180, 261], [20, 10, 60, 40]]          # [v1, u1, v2, u2], truncated like the reference's int boxes
    assert im["obj_ids"] == [4, 1] and im["scores"] == [0.9, 0.5]
    np.testing.assert_array_equal(B.rle_decode(im["segmentations"][0]), mask)
    assert im["segmentations"][1] is None
    from pix2pose_amd import synthetic as S
    assert im["cam_K"] == S.LM_K.reshape(-1).tolist() and "segmentations" not in dump["images"][1]
    im49 = [i for i in dump["images"] if (i["scene_id"], i["im_id"]) == (49, 21)][0]
    assert im49["rois"] == [[50, 300, 170, 400]] and im49["cam_K"][2] == S.LM_K[0, 2] + 1 and im49["cam_K"][5] == S.LM_K[1, 2] - 2
    assert dump["weights"]["4"].endswith(os.path.join("pix2pose_weights", "04", "inference.npz"))
    assert set(dump["norm_factor"]) == {"1", "4", "9"} and dump["norm_factor"]["9"]["x_scale"] > dump["norm_factor"]["1"]["x_scale"]
    # object filter, split names, missing weights
    cfg["target_obj"] = [4]
    assert B.build_dump(cfg, "ycbv", dets)["model_ids"] == [4]
    assert B.dataset_dirs("/d", "tless")[1] == "/d/tless/test_primesense" and B.dataset_dirs("/d", "lmo")[1] == "/d/lmo/test"
    os.remove(dump["weights"]["4"])
    with pytest.raises(FileNotFoundError, match="inference"):
        B.build_dump(cfg, "ycbv", dets)


def test_directory_conventions_match_the_reference_script(tmp_path):
    """tests/golden/reference_eval.json holds what the reference's OWN evaluation script (run unmodified on a synthetic BOP-style
    directory, tests/golden/make_reference_eval_vectors.py) passed to every pix2pose(...) it constructed: the weight file it
    looked for, obj_param from the norm-factor file, the frame size.  The same directory, rebuilt here from the fixture, must
    give the same object list, weight paths, parameters, targets and per-image intrinsics through build_dump."""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_eval.json")))
    H, Wd = fx["frame"]
    model_ids = fx["model_ids"]
    run = fx["runs"][0]                                # backbone resnet50, the generating script's default cfg
    bop = str(tmp_path / "bop")
    ds = os.path.join(bop, "lmo")
    for d in ("models", "models_xyz"):
        os.makedirs(os.path.join(ds, d))
    json.dump({str(m): {"diameter": 100.0} for m in model_ids + [99]}, open(os.path.join(ds, "models", "models_info.json"), "w"))
    for m in model_ids:                                # object 99 has no mesh: the reference drops it
        open(os.path.join(ds, "models", "obj_%06d.ply" % m), "w").close()
    json.dump({str(m): {"x_scale": 30.0 + m, "y_scale": 31.0 + m, "z_scale": 32.0 + m, "x_ct": 0.5 * m, "y_ct": -0.25 * m, "z_ct": 1.0}
               for m in model_ids}, open(os.path.join(ds, "models_xyz", "norm_factor.json"), "w"))
    json.dump(fx["targets"], open(os.path.join(ds, "targets.json"), "w"))
    json.dump({"cx": 80.0, "cy": 60.0, "depth_scale": 1.0, "fx": 572.4, "fy": 573.6, "height": H, "width": Wd}, open(os.path.join(ds, "camera.json"), "w"))
    K = [572.4, 0, 80.0, 0, 573.6, 60.0, 0, 0, 1]
    for scene in sorted({t["scene_id"] for t in fx["targets"]}):
        sdir = os.path.join(ds, "test", "%06d" % scene)
        os.makedirs(os.path.join(sdir, "rgb"))
        json.dump({str(im): {"cam_K": (np.array(K) + 0.001 * im).tolist(), "depth_scale": 1.0} for im in (4, 9, 17)},
                  open(os.path.join(sdir, "scene_camera.json"), "w"))
    for c in run["ctor"]:                              # the files the reference would open (inference_resnet_model.hdf5 first; recorded: its fallback name, relative to dataset_dir)
        fn = os.path.join(bop, c["weight_fn"])
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        open(fn, "w").close()
    cfg = {"backbone": "resnet50", "dataset_dir": bop, "norm_factor_fn": "norm_factor.json", "test_target": "targets"}
    dump = B.build_dump(cfg, "lmo", [])
    assert dump["model_ids"] == sorted(model_ids) and len(run["ctor"]) == len(model_ids)
    assert dump["im_size"] == run["ctor"][0]["res"]
    for m, c in zip(dump["model_ids"], run["ctor"]):
        assert os.path.relpath(dump["weights"][str(m)], bop) == c["weight_fn"]
        np.testing.assert_array_equal(E.model_params_to_obj_param(dump["norm_factor"][str(m)]), c["obj_param"])
    assert dump["targets"] == fx["targets"]
    for im in dump["images"]:
        np.testing.assert_array_equal(im["cam_K"], (np.array(K) + 0.001 * im["im_id"]).tolist())
        assert im["rgb"] == os.path.join(ds, "test", "%06d" % im["scene_id"], "rgb", "%06d.png" % im["im_id"])
    # a converted .npz next to the .hdf5 takes precedence
    first = dump["weights"][str(dump["model_ids"][0])]
    open(os.path.splitext(first)[0] + ".npz", "w").close()
    assert B.build_dump(cfg, "lmo", [])["weights"][str(dump["model_ids"][0])].endswith(".npz")


def test_frame_prefetcher_returns_what_a_direct_load_returns(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(2)
    paths, imgs = [], []
    for i in range(7):
        a = rs.randint(0, 255, (48, 64, 3)).astype(np.uint8) if i != 3 else rs.randint(0, 255, (48, 64)).astype(np.uint8)   # one gray frame
        fn = str(tmp_path / ("%d.png" % i))
        Image.fromarray(a).save(fn)
        paths.append(fn); imgs.append(a)
    np.save(tmp_path / "n.npy", imgs[0])
    for threads in (0, 3):
        pf = E.FramePrefetcher(threads)
        pf.request(paths[:4])
        pf.request(paths[2:6])                       # overlapping requests are decoded once
        for i in (1, 0, 3, 2, 6, 5, 4):              # any order, requested or not
            got = pf.get(paths[i])
            exp = imgs[i] if imgs[i].ndim == 3 else np.repeat(imgs[i][:, :, None], 3, axis=2)
            np.testing.assert_array_equal(got, exp)
        np.testing.assert_array_equal(pf.get(str(tmp_path / "n.npy")), imgs[0])
        assert not pf._pending
        pf.close()


def test_rows_survive_the_gather_records():
    rs = np.random.RandomState(0)
    rows = [{"scene_id": 48 + i % 2, "im_id": 7 * i, "obj_id": 3 + i, "score": float(rs.uniform()), "R": rs.normal(size=(3, 3)),
             "t": rs.normal(size=3) * 1000, "time": 0.125 * i, "_order": (i // 2, i % 2)} for i in range(7)]
    back = E.records_to_rows(E.rows_to_records(rows))
    for a, b in zip(rows, back):
        assert (a["scene_id"], a["im_id"], a["obj_id"], a["score"], a["time"], a["_order"]) == (b["scene_id"], b["im_id"], b["obj_id"], b["score"], b["time"], b["_order"])
        np.testing.assert_array_equal(a["R"], b["R"])
        np.testing.assert_array_equal(a["t"], b["t"])


# ------------------------------------------------------------------------------------------ the stream on the GPU
def _read_csv(fn):
    lines = open(fn).read().split("\n")
    assert lines[0] == "scene_id,im_id,obj_id,score,R,t,time"
    return [ln.split(",")[:6] for ln in lines[1:]]


@pytest.mark.gpu
def test_bop_directory_stream_one_and_two_ranks(tmp_path):
    """A BOP directory (PNG frames, scene_camera.json, targets, norm factors, per-object weights) and COCO-style detections with
    run-length masks go through the command-line harness as one process and as two ranks (gloo, both on cuda:0) that shard
    the images and gather the rows: the two CSV files hold the same rows in the same order, and they are the rows of the
    in-process run over the dict build_dump returns."""
    from pix2pose_amd import synthetic as S
    n_scenes, per_scene, model_ids = 2, 4, (1, 4, 9)
    n_img = n_scenes * per_scene
    sc = S.make_scene(n_img * 3, seed=31, n_images=n_img, bbox_side=(70, 140))
    H, Wd = sc["images"].shape[1:3]
    root = str(tmp_path)
    cfg, targets, paths = make_bop_dir(root, "ycbv", n_scenes, per_scene, model_ids, frames=[sc["images"][i] for i in range(n_img)])
    cfg["target_obj"] = list(model_ids)         # models_info.json also lists an object without weights
    tl = E.group_targets(targets)
    dets, key, rows1, rows2 = [], [], [], []
    for gi, (sid, iid, obj_t, inst) in enumerate(tl):
        for k in range(3):                      # three detections per image: two target objects and one the image does not ask for
            i = gi * 3 + k
            b = [int(v) for v in sc["dets"][i][2]]
            mask = np.zeros((H, Wd), bool)
            mask[max(b[0], 0) + 6:b[2] - 4, max(b[1], 0) + 3:b[3] - 7] = True
            seg = B.rle_encode(mask)
            if k == 1:
                seg = {"size": seg["size"], "counts": _rle_to_string(seg["counts"])}
            dets.append({"scene_id": sid, "image_id": iid, "category_id": int(model_ids[k]), "bbox": [b[1], b[0], b[3] - b[1], b[2] - b[0]],
                         "score": 0.5 + 0.01 * i, "segmentation": seg})
            key.append((gi, k)); rows1.append(sc["inject1"][i]); rows2.append(sc["inject2"][i])
    json.dump(dets, open(os.path.join(root, "detections.json"), "w"))
    json.dump(cfg, open(os.path.join(root, "cfg.json"), "w"))
    np.savez(os.path.join(root, "inject.npz"), key=np.array(key), inject1=np.stack(rows1), inject2=np.stack(rows2))
    env = dict(os.environ, P2P_EVAL_INJECT=os.path.join(root, "inject.npz"))
    args = ["0", os.path.join(root, "cfg.json"), "ycbv", os.path.join(root, "detections.json")]
    r = subprocess.run([sys.executable, "-m", "pix2pose_amd.eval_bop"] + args, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    csv_fn = os.path.join(root, "out", "pix2pose-iccv19_ycbv-test.csv")
    one = _read_csv(csv_fn)
    os.rename(csv_fn, csv_fn + ".1")
    env2 = dict(env, P2P_EVAL_BACKEND="gloo", P2P_EVAL_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", "-m", "pix2pose_amd.eval_bop"] + args, capture_output=True, text=True, cwd=ROOT, env=env2, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    two = _read_csv(csv_fn)
    assert len(one) >= 10 and one == two
    # ragged shards: three ranks, two images per step -> 3 / 3 / 2 images = 2 / 2 / 1 steps: rank 2 has NO batch at the second step and
    # joins that step's collective as an empty shard (the per-step exchange of pose records; over RCCL it is the C ABI's
    # p2p_est_pose_collect_gathered with P2P_TICKET_NONE, here -- three ranks on one device -- torch.distributed carries the same records)
    os.rename(csv_fn, csv_fn + ".2")
    json.dump(dict(cfg, batch_images=2), open(os.path.join(root, "cfg3.json"), "w"))
    args3 = ["0", os.path.join(root, "cfg3.json"), "ycbv", os.path.join(root, "detections.json")]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", "-m", "pix2pose_amd.eval_bop"] + args3, capture_output=True, text=True, cwd=ROOT, env=env2, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert _read_csv(csv_fn) == one
    # ... and they are the rows of the in-process run (same text after the float round trip)
    dump = B.build_dump(cfg, "ycbv", dets)
    with np.load(os.path.join(root, "inject.npz")) as z:
        inject = {k: z[k] for k in ("key", "inject1", "inject2")}
    rows = E.run(dict(cfg, path_to_output=None), "ycbv", dump, base_dir="/", batch_images=3, inject=inject)
    assert len(rows) == len(one)
    # the harness's per-step exchange THROUGH THE C ABI (p2p_est_pose_collect_gathered over RCCL: parallel.CabiPoseGather) with the one rank a
    # test box has: run_distributed on the "nccl" backend, two images per step -> every pooled batch is collected through the collective and
    # the rows are rebuilt from the gathered records (mask sums included) -- identical to the plain run
    import socket
    import torch.distributed as dist
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env_keep = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        rows_c = E.run_distributed(dict(cfg, path_to_output=None), "ycbv", dump, base_dir="/", backend="nccl", batch_images=2, inject=inject)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in env_keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert len(rows_c) == len(rows)
    for a, b in zip(rows_c, rows):
        assert [a["scene_id"], a["im_id"], a["obj_id"]] == [b["scene_id"], b["im_id"], b["obj_id"]] and a["score"] == b["score"]
        np.testing.assert_array_equal(np.asarray(a["R"]).reshape(-1), np.asarray(b["R"]).reshape(-1))
        np.testing.assert_array_equal(np.asarray(a["t"]).reshape(-1), np.asarray(b["t"]).reshape(-1))
    for a, f in zip(rows, one):
        assert [a["scene_id"], a["im_id"], a["obj_id"]] == [int(f[0]), int(f[1]), int(f[2])]
        assert str(a["score"]) == f[3] and " ".join(map(str, np.asarray(a["R"]).flatten().tolist())) == f[4]
    # only target objects of an image come back, at most inst_count (= 1) each under the ViVo task
    per_image = {}
    for f in one:
        per_image.setdefault((int(f[0]), int(f[1])), []).append(int(f[2]))
    for sid, iid, obj_t, inst in tl:
        got = per_image.get((sid, iid), [])
        assert set(got) <= set(obj_t) and len(got) == len(set(got))
