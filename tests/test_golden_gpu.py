"""GPU: the HIP path (through the C ABI) against the committed golden vectors -- no oracle run."""
import json
import os

import numpy as np
import pytest

from pix2pose_amd import synthetic, weights as W

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


@pytest.mark.parametrize("backbone", ["resnet50", "paper"])
def test_generator_vs_golden(backbone):
    from pix2pose_amd.runtime import Generator
    g = G["ae"][backbone]
    gen = Generator(W.synthetic_weights(backbone, g["weights_seed"]), backbone)
    x = (np.random.RandomState(g["input_seed"]).randint(0, 256, (g["n"], 128, 128, 3)).astype(np.float32) - 128) / 128
    d, p = gen.predict(x)
    idx = np.array(g["pixel_index"])
    assert np.abs(d.reshape(-1, 3)[idx] - np.array(g["decode"])).max() < 1e-4      # north_star bar: 1e-3 abs
    assert np.abs(p.reshape(-1)[idx] - np.array(g["prob"])).max() < 1e-4
    assert abs(np.abs(d).astype(np.float64).mean() - g["decode_abs_mean"]) < 1e-5


def test_pnp_vs_golden():
    from pix2pose_amd.runtime import default_context, pnp_ransac_batch
    cs = G["pnp"]
    ok, R, t, info, masks = pnp_ransac_batch(default_context(), [synthetic.LM_K] * len(cs), [np.array(c["obj"]) for c in cs],
                                             [np.array(c["img"]) for c in cs], want_mask=True)
    for i, c in enumerate(cs):
        assert bool(ok[i]) == c["ok"]
        if c["ok"]:
            assert np.nonzero(masks[i])[0].tolist() == c["inliers"]
            assert [int(v) for v in info[i]] == [c["meta"]["n_inliers"], c["meta"]["iterations"], c["meta"]["best_iter"]]
            dt, dr = synthetic.pose_error(np.array(c["R"]), np.array(c["t"]), R[i], t[i])
            assert dt < 1e-6 and dr < 1e-4


def test_est_pose_vs_golden():
    import torch
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch
    g = G["est_pose"]
    ctx = Context(0, max_batch=16)
    gen = Generator(W.synthetic_weights("paper", 1), "paper", ctx)
    spec = ObjectSpec(gen, synthetic.OBJ_PARAM, g["th_outlier"], g["th_inlier"])
    sc = synthetic.make_scene(g["n_det"], seed=g["seed"])
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    poses, ex = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=j1.data_ptr(), inject2=j2.data_ptr(),
                               inject_slots=3, debug=True)
    for i, gd in enumerate(g["dets"]):
        p = poses[i]
        assert p.status == 0 and p.n_init_mask == gd["n_init_mask"] and list(p.bbox_t) == gd["bbox_t"]
        assert ex["boxes2"][i].tolist() == gd["boxes2"] and p.best_slot == gd["best_slot"]
        for c in gd["cands"]:
            assert ex["cand"][i, c["slot"], :4].tolist() == [1, c["n_non_gray"], c["n_valid"], c["n_inliers"]]
        dt, dr = synthetic.pose_error(np.array(gd["R"]), np.array(gd["t"]), np.array(p.R).reshape(3, 3), np.array(p.t))
        assert dt < 1e-6 and dr < 1e-4
        assert abs(p.frac_inlier - gd["frac_inlier"]) < 1e-12
        assert abs(ex["x1"][i].astype(np.float64).sum() - gd["x1_sum"]) < 1e-3
