"""BASELINE.json full-size properties (configs[1]: batch 64 generator; configs[2]: 256 detections
+ 768 PnP solves) through size-independent invariants -- the oracle is too slow at these sizes:
determinism, chunking invariance, batch-order (permutation) invariance, batch-composition
independence, and agreement with ground truth of the synthetic scene."""
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
    big = Generator(w, "resnet50", Context(0, max_batch=64))
    d64, p64 = big.predict(x)
    small = Generator(w, "resnet50", Context(0, max_batch=24))       # 64 = 24 + 24 + 16
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
    gen = Generator(W.synthetic_weights("resnet50", 1), "resnet50", ctx)
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
    ref = [est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3)[0]
           for sc, (a, b) in zip(scenes, inj)]
    pend, got = [], []
    for sc, (a, b) in zip(scenes, inj):
        pend.append(est_pose_submit(ctx, [spec], list(sc["images"]), sc["dets"], inject1=a.data_ptr(), inject2=b.data_ptr(), inject_slots=3))
        if len(pend) == 2:
            got.append(pend.pop(0).collect())
    while pend:
        got.append(pend.pop(0).collect())
    for r, g in zip(ref, got):
        assert [_key(x) for x in r] == [_key(x) for x in g]
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
