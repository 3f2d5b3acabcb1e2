"""BASELINE.json full sizes.  configs[1] (batch-64 generator): chunking invariance, device-pointer path.  configs[2]
(256 detections + 768 PnP solves): EVERY detection against the oracle (the injected pipeline and PnP are fast on the
CPU), plus determinism, batch-order and batch-composition invariance and agreement with the scene's ground truth.
configs[3] single-GPU share (30 resnet50 objects x 256 detections): the grouped generator pass == per-object passes bit
for bit, oracle on a 16-detection subset.  The asynchronous API returns the full reference tuple."""
import numpy as np
import pytest

from pix2pose_amd import synthetic as S
from pix2pose_amd import weights as W

pytestmark = pytest.mark.gpu
TH_O, TH_I = [0.2, 0.3, 0.35], 0.2


def _key(p):
    return (p.status, p.n_inliers, p.n_init_mask, p.best_slot, tuple(p.bbox_t), tuple(p.R), tuple(p.t), p.frac_inlier)


def test_generator_batch64_chunking_and_device_pointers():
    import torch
    from pix2pose_amd.runtime import Context, Generator
    w = W.synthetic_weights("resnet50", 1)
    x = (np.random.RandomState(1).randint(0, 256, (64, 128, 128, 3)).astype(np.float32) - 128) / 128
    big = Generator(w, "resnet50", Context(0, max_batch=64, winograd="off"))      # one form of the 5x5 layers at every pass size: bits do not depend on the chunking
    d64, p64 = big.predict(x)
    small = Generator(w, "resnet50", Context(0, max_batch=24, winograd="off"))       # 64 = 24 + 24 + 16
    d24, p24 = small.predict(x)
    np.testing.assert_array_equal(d64, d24)
    np.testing.assert_array_equal(p64, p24)
    # device-pointer path (interleaved output) == host path
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty(64, 128, 128, 4, device="cuda")
    torch.cuda.synchronize()
    big.forward_device(xd.data_ptr(), 64, yd.data_ptr())
    big.ctx.synchronize()
    y = yd.cpu().numpy()
    np.testing.assert_array_equal(y[..., :3], d64)
    np.testing.assert_array_equal(y[..., 3:], p64)
    assert np.isfinite(d64).all() and np.abs(d64).max() <= 1.0 and (p64 > 0).all() and (p64 < 1).all()


def test_est_pose_256_detections_invariants():
    import torch
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch
    ctx = Context(0, max_batch=256)
    gen = Generator(W.synthetic_weights("resnet50", 1), "resnet50", ctx)       # (decoder maps injected: the generator's form of the 5x5 layers does not reach the poses)
    spec = ObjectSpec(gen, S.OBJ_PARAM, TH_O, TH_I)
    sc = S.make_scene(256, seed=5)
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    run = lambda dets, a, b: est_pose_batch(ctx, [spec], list(sc["images"]), dets, inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3)[0]
    p1 = run(sc["dets"], j1, j2)
    p2 = run(sc["dets"], j1, j2)
    assert [_key(a) for a in p1] == [_key(b) for b in p2]                      # deterministic
    perm = np.random.RandomState(0).permutation(256)
    jp1, jp2 = j1[torch.from_numpy(perm).cuda()].contiguous(), j2[torch.from_numpy(perm).cuda()].contiguous()
    torch.cuda.synchronize()
    pp = run([sc["dets"][i] for i in perm], jp1, jp2)
    assert [_key(pp[k]) for k in range(256)] == [_key(p1[i]) for i in perm]    # order invariant
    for i in (0, 77, 255):                                                    # composition independent
        a, b = j1[i:i + 1].contiguous(), j2[i:i + 1].contiguous()
        torch.cuda.synchronize()
        assert _key(run([sc["dets"][i]], a, b)[0]) == _key(p1[i])
    ok = [p for p in p1 if p.status == 0]
    assert len(ok) >= 250
    errs = np.array([S.pose_error(sc["gt"][i][0], sc["gt"][i][1], np.array(p.R).reshape(3, 3), np.array(p.t))
                     for i, p in enumerate(p1) if p.status == 0])
    assert np.median(errs[:, 0]) < 12.0 and np.median(errs[:, 1]) < 1.0        # 8-bit XYZ + 10 % undetected outliers
    for p in ok:
        R = np.array(p.R).reshape(3, 3)
        assert abs(np.linalg.det(R) - 1) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9
        assert 0 < p.frac_inlier <= 4.0 and p.n_inliers >= 5
    # every one of the 256 detections against the oracle: status, counts, selected candidate, box exactly; pose to rounding
    _check_against_oracle(sc, p1, range(256))


def _check_against_oracle(sc, poses, idx, th_o=TH_O, th_i=TH_I, obj_param=None):
    from oracle import est_pose_oracle as E
    n_ok = 0
    for i in idx:
        def predict(x, stage, slots=None, i=i):
            m = sc["inject1"][i][None] if stage == 1 else sc["inject2"][i][slots]
            return [m[..., :3].copy(), m[..., 3:].copy()]
        img_i, _, bbox, K = sc["dets"][i]
        dbg = {}
        ref = E.est_pose(sc["images"][img_i], bbox, predict, K, sc["obj_param"] if obj_param is None else obj_param, th_o, th_i, debug=dbg)
        p = poses[i]
        ok_ref = not (isinstance(ref[4], int) and ref[4] == -1)
        assert (p.status == 0) == ok_ref, (i, p.status)
        np.testing.assert_array_equal(np.array(list(p.bbox_t)), ref[5])
        assert p.n_init_mask == dbg.get("n_init_mask", p.n_init_mask), i
        if ok_ref:
            n_ok += 1
            assert abs(p.frac_inlier - ref[4]) < 1e-12, (i, p.frac_inlier, ref[4])
            assert p.best_slot == dbg["slots"][dbg["best"]], i
            assert p.n_inliers == dbg["cands"][dbg["best"]]["n_inliers"], i
            dt, dr = S.pose_error(ref[2], ref[3], np.array(p.R).reshape(3, 3), np.array(p.t))
            assert dt < 1e-6 and dr < 1e-4, (i, dt, dr)
    return n_ok


def test_configs3_share_30_objects_256_detections():
    """BASELINE.json configs[3] on one GPU: 256 detections spread over 30 resnet50 object models (per-object weights,
    per-object obj_param), one grouped generator pass per stage.  (a) equals the 30 per-object passes bit for bit;
    (b) equals the oracle on a 16-detection subset; (c) the real (non-injected) generator output of the grouped pass
    equals the per-object generator output bit for bit (stage-2 inputs and candidate counts through the debug taps)."""
    import torch
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch
    n_obj, n_det = 30, 256
    # (c) compares a 256-input pass with 8-9-input passes bit for bit: the form of the 5x5 decoder layers must not depend on the pass size --
    # "always" = Winograd form throughout (objects with odd detection counts: the two-samples-per-workgroup layer pairs every object up on its own)
    ctx = Context(0, max_batch=1024, winograd="always")
    params = [S.OBJ_PARAM * (1.0 + 0.02 * k) for k in range(n_obj)]
    specs = [ObjectSpec(Generator(W.synthetic_weights("resnet50", 100 + k), "resnet50", ctx), params[k], TH_O, TH_I) for k in range(n_obj)]
    sc = S.make_scene(n_det, seed=31)
    obj_of = [(7 * i + 3) % n_obj for i in range(n_det)]                  # interleaved: the library sorts by object
    dets = [(d[0], obj_of[i], d[2], d[3]) for i, d in enumerate(sc["dets"])]
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    both = est_pose_batch(ctx, specs, list(sc["images"]), dets, inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3)[0]
    raw_both = est_pose_batch(ctx, specs, list(sc["images"]), dets, debug=True)[1]       # generator output drives everything
    for o in range(n_obj):
        idx = [i for i in range(n_det) if obj_of[i] == o]
        sub = [(dets[i][0], 0, dets[i][2], dets[i][3]) for i in idx]
        ii = torch.tensor(idx, device="cuda")
        a, b = j1[ii].contiguous(), j2[ii].contiguous()
        torch.cuda.synchronize()
        alone = est_pose_batch(ctx, [specs[o]], list(sc["images"]), sub, inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3)[0]
        assert [_key(both[i]) for i in idx] == [_key(x) for x in alone], o
        raw_alone = est_pose_batch(ctx, [specs[o]], list(sc["images"]), sub, debug=True)[1]
        for k, i in enumerate(idx):
            np.testing.assert_array_equal(raw_both["x2"][i], raw_alone["x2"][k])       # depends on the stage-1 network output
            np.testing.assert_array_equal(raw_both["cand"][i], raw_alone["cand"][k])   # depends on the stage-2 network output
    # oracle on a subset (per-object obj_param)
    for i in range(0, n_det, 16):
        _check_against_oracle(sc, both, [i], obj_param=params[obj_of[i]])
    assert sum(1 for p in both if p.status == 0) >= 245


@pytest.mark.parametrize("merge", [False, True])
def test_stream_of_unequal_batches_equals_blocking(merge):
    """The detection stream defers batch k's stage-2 generator pass to the next submit (or its collect); with merge_passes it
    merges it with batch k+1's stage-1 pass when the newcomer fits (same size or smaller); batches of different sizes, with two objects of different
    threshold counts, masks and detector masks: every result equals the blocking call's."""
    import torch
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch, est_pose_submit
    ctx = Context(0, max_batch=48)                       # smaller than the merged passes: chunking inside them
    specs = [ObjectSpec(Generator(W.synthetic_weights("paper", 1), "paper", ctx), S.OBJ_PARAM, TH_O, TH_I),
             ObjectSpec(Generator(W.synthetic_weights("paper", 2), "paper", ctx), S.OBJ_PARAM * 1.1, [0.25, 0.4], 0.25)]
    sizes = [24, 10, 30, 7, 7, 16]
    scenes, inj, dets, dmasks = [], [], [], []
    for k, n in enumerate(sizes):
        sc = S.make_scene(n, seed=60 + k, bbox_side=(70, 120) if k % 2 else (86, 86))
        scenes.append(sc)
        inj.append((torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()))
        dets.append([(d[0], (i + k) % 2, d[2], d[3]) for i, d in enumerate(sc["dets"])])
        H, Wd = sc["images"].shape[1:3]
        ms = []
        for d in sc["dets"]:
            m = np.zeros((H, Wd), bool)
            m[max(d[2][0], 0) + 6:d[2][2] - 4, max(d[2][1], 0) + 3:d[2][3] - 7] = True
            ms.append(m)
        dmasks.append(ms)
    torch.cuda.synchronize()
    kw = lambda k: dict(inject1=inj[k][0].data_ptr(), inject2=inj[k][1].data_ptr(), inject_slots=3, want_masks=True, det_masks=dmasks[k])
    ref = [est_pose_batch(ctx, specs, list(scenes[k]["images"]), dets[k], **kw(k)) for k in range(len(sizes))]
    pend, got = [], []
    for k in range(len(sizes)):
        pend.append(est_pose_submit(ctx, specs, list(scenes[k]["images"]), dets[k], merge_passes=merge, **kw(k)))
        if len(pend) == 2:
            pb = pend.pop(0)
            got.append((pb.collect(), pb.extras))
    while pend:
        pb = pend.pop(0)
        got.append((pb.collect(), pb.extras))
    assert sum(p.status == 0 for r in ref for p in r[0]) >= 70
    for k, ((rp, rex), (gp, gex)) in enumerate(zip(ref, got)):
        assert [_key(x) for x in rp] == [_key(x) for x in gp], k
        for name in ("valid_mask", "img_pred", "mask_stats"):
            np.testing.assert_array_equal(rex[name], gex[name])
    # a blocking call while the last submitted batch still waits for its stage-2 pass: both come out right
    p = est_pose_submit(ctx, specs, list(scenes[1]["images"]), dets[1], merge_passes=merge, **kw(1))
    q = est_pose_submit(ctx, specs, list(scenes[3]["images"]), dets[3], merge_passes=merge, **kw(3))
    assert [_key(x) for x in p.collect()] == [_key(x) for x in ref[1][0]]
    blk = est_pose_batch(ctx, specs, list(scenes[0]["images"]), dets[0], **kw(0))[0]       # slot 0 is free again, slot 1 pending
    assert [_key(x) for x in blk] == [_key(x) for x in ref[0][0]]
    assert [_key(x) for x in q.collect()] == [_key(x) for x in ref[3][0]]


def test_async_submit_collect_equals_blocking():
    """Stream mode (two batches in flight, PnP tail on a second HIP stream) returns exactly what the
    blocking call returns, batch after batch; misuse is reported, not hung."""
    import torch
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch, est_pose_submit
    ctx = Context(0, max_batch=64)
    gen = Generator(W.synthetic_weights("paper", 1), "paper", ctx)
    spec = ObjectSpec(gen, S.OBJ_PARAM, TH_O, TH_I)
    scenes = [S.make_scene(24, seed=40 + k) for k in range(4)]
    inj = [(torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()) for sc in scenes]
    torch.cuda.synchronize()
    # detector masks for the score_type-2 sums: a rectangle inside every box
    dmasks = []
    for sc in scenes:
        H, Wd = sc["images"].shape[1:3]
        ms = []
        for d in sc["dets"]:
            m = np.zeros((H, Wd), bool)
            m[d[2][0] + 10:d[2][2] - 5, d[2][1] + 8:d[2][3] - 12] = True
            ms.append(m)
        dmasks.append(ms)
    ref = [est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3,
                          want_masks=True, det_masks=dm)
           for sc, (a, b), dm in zip(scenes, inj, dmasks)]
    pend, got = [], []
    for sc, (a, b), dm in zip(scenes, inj, dmasks):
        pend.append(est_pose_submit(ctx, [spec], list(sc["images"]), sc["dets"], inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3,
                                    want_masks=True, det_masks=dm))
        if len(pend) == 2:
            pb = pend.pop(0)
            got.append((pb.collect(), pb.extras))
    while pend:
        pb = pend.pop(0)
        got.append((pb.collect(), pb.extras))
    for (rp, rex), (gp, gex) in zip(ref, got):
        assert [_key(x) for x in rp] == [_key(x) for x in gp]
        # the full reference tuple comes back from the asynchronous call too (recognition.py:189-193) ...
        np.testing.assert_array_equal(rex["valid_mask"], gex["valid_mask"])
        np.testing.assert_array_equal(rex["img_pred"], gex["img_pred"])
        assert rex["valid_mask"].any() and rex["img_pred"].any()
        # ... and so do the score_type-2 mask sums (tools/5_evaluation_bop_basic.py:307-316)
        np.testing.assert_array_equal(rex["mask_stats"], gex["mask_stats"])
    ref = [r[0] for r in ref]
    # three in flight -> capacity error; blocking call while one is in flight -> error; then recover
    a, b = inj[0]
    p1 = est_pose_submit(ctx, [spec], list(scenes[0]["images"]), scenes[0]["dets"], inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3)
    p2 = est_pose_submit(ctx, [spec], list(scenes[0]["images"]), scenes[0]["dets"], inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3)
    with pytest.raises(_lib.P2PError):
        est_pose_submit(ctx, [spec], list(scenes[0]["images"]), scenes[0]["dets"])
    with pytest.raises(_lib.P2PError):
        est_pose_batch(ctx, [spec], list(scenes[0]["images"]), scenes[0]["dets"])
    assert [_key(x) for x in p1.collect()] == [_key(x) for x in ref[0]]
    assert [_key(x) for x in p2.collect()] == [_key(x) for x in ref[0]]
    with pytest.raises(_lib.P2PError):
        p2.collect()                      # ticket no longer in flight
