"""Operand-range guard of the split-f16 arithmetic (include/p2p_mi355.h: p2p_precision, P2P_ERR_RANGE, P2P_PREC_AUTO).
P2P_PREC_F16X3 splits every fp32 activation into two f16 halves, so |activation| must stay below 65504; the layer epilogues track the
largest magnitude they store.  The reference computes in fp32 (TensorFlow) and has no such limit -- the two linear Dense layers
(ae_model.py:199-200) have no BatchNorm behind them -- so the test scales the first Dense layer until its output leaves the range:
  f16x3 alone  -> P2PRangeError from predict() and from est_pose (blocking and submit / collect), nothing returned silently;
  auto         -> the object falls back to its strict-fp32 twin: same bits as an 'f32' generator, calls succeed;
  normal weights never trip the guard."""
import numpy as np
import pytest

from pix2pose_amd import synthetic as S
from pix2pose_amd import weights as W

pytestmark = pytest.mark.gpu


def _overflowing(backbone, seed=3, gain=3e5):
    w = dict(W.synthetic_weights(backbone, seed))
    w["dense_enc.kernel"] = (w["dense_enc.kernel"] * np.float32(gain)).astype(np.float32)       # |dense_enc output| ~ 1e5 .. 1e6
    w["dense_dec.kernel"] = (w["dense_dec.kernel"] / np.float32(gain)).astype(np.float32)       # the rest of the network sees normal values again
    return w


def _x(n, seed=0):
    return ((np.random.RandomState(seed).randint(0, 256, (n, 128, 128, 3)).astype(np.float32)) - 128) / 128


@pytest.mark.parametrize("backbone", ["paper", "resnet50"])
def test_predict_guard_and_fp32_fallback(backbone):
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Context, Generator
    ctx = Context(0, max_batch=4)
    w = _overflowing(backbone)
    x = _x(6)                                                   # two chunks of max_batch
    ref = Generator(w, backbone, ctx, precision="f32").predict(x)
    assert np.isfinite(ref[0]).all() and np.abs(ref[0]).max() > 0.05
    g = Generator(w, backbone, ctx, precision="f16x3")
    with pytest.raises(_lib.P2PRangeError) as e:
        g.predict(x)
    assert "operand range" in str(e.value)
    # the flag does not stick: a normal model on the same context is not blamed for it
    ok = Generator(W.synthetic_weights(backbone, 3), backbone, ctx, precision="f16x3")
    ok.predict(x)
    ga = Generator(w, backbone, ctx, precision="auto")
    assert ga.active_precision == "f16x3"
    out = ga.predict(x)
    assert ga.active_precision == "f32"
    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])       # the twin IS the fp32 model


def test_est_pose_guard_blocking_and_async():
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch, est_pose_submit
    ctx = Context(0, max_batch=16)
    w = _overflowing("paper")
    sc = S.make_scene(3, seed=77)
    imgs = list(sc["images"])
    ths = ([0.2, 0.3, 0.35], 0.2)
    normal = ObjectSpec(Generator(W.synthetic_weights("paper", 3), "paper", ctx), S.OBJ_PARAM, *ths)
    p_ok, _ = est_pose_batch(ctx, [normal], imgs, sc["dets"])          # real generator output (no injection): whatever it gives, no range error
    assert len(p_ok) == 3
    bad = ObjectSpec(Generator(w, "paper", ctx, precision="f16x3"), S.OBJ_PARAM, *ths)
    with pytest.raises(_lib.P2PRangeError):
        est_pose_batch(ctx, [bad], imgs, sc["dets"])
    pend = est_pose_submit(ctx, [bad], imgs, sc["dets"])
    with pytest.raises(_lib.P2PRangeError):
        pend.collect()
    # a following batch of a normal object is clean again (per-batch flags)
    p2, _ = est_pose_batch(ctx, [normal], imgs, sc["dets"])
    assert [p.status for p in p2] == [p.status for p in p_ok] and [tuple(p.t) for p in p2] == [tuple(p.t) for p in p_ok]
    # auto: the blocking call repeats the batch in fp32 by itself and equals an f32 object's result
    f32 = ObjectSpec(Generator(w, "paper", ctx, precision="f32"), S.OBJ_PARAM, *ths)
    want, _ = est_pose_batch(ctx, [f32], imgs, sc["dets"])
    ga = Generator(w, "paper", ctx, precision="auto")
    got, _ = est_pose_batch(ctx, [ObjectSpec(ga, S.OBJ_PARAM, *ths)], imgs, sc["dets"])
    assert ga.active_precision == "f32"
    key = lambda p: (p.status, p.n_inliers, p.n_init_mask, tuple(p.bbox_t), tuple(p.R), tuple(p.t))
    assert [key(p) for p in got] == [key(p) for p in want]
    # auto + async: one P2P_ERR_RANGE at collect, the re-submitted batch runs in fp32
    gb = Generator(w, "paper", ctx, precision="auto")
    spec_b = ObjectSpec(gb, S.OBJ_PARAM, *ths)
    pend = est_pose_submit(ctx, [spec_b], imgs, sc["dets"])
    with pytest.raises(_lib.P2PRangeError):
        pend.collect()
    assert gb.active_precision == "f32"
    again = est_pose_submit(ctx, [spec_b], imgs, sc["dets"]).collect()
    assert [key(p) for p in again] == [key(p) for p in want]


def test_mixed_batch_with_one_object_falling_back():
    """Two objects in one batch (a grouped generator pass); one of them overflows and is P2P_PREC_AUTO: it switches to its fp32 twin, the
    batch is repeated with the two objects on different arithmetics (per-object passes instead of the grouped one), and every detection
    equals what its object returns alone."""
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch
    ctx = Context(0, max_batch=16)
    ths = ([0.2, 0.3, 0.35], 0.2)
    w_bad, w_ok = _overflowing("paper"), W.synthetic_weights("paper", 5)
    sc = S.make_scene(6, seed=78)
    imgs = list(sc["images"])
    dets = [(d[0], i % 2, d[2], d[3]) for i, d in enumerate(sc["dets"])]
    key = lambda p: (p.status, p.n_inliers, p.n_init_mask, tuple(p.bbox_t), tuple(p.R), tuple(p.t))
    ga = Generator(w_bad, "paper", ctx, precision="auto")
    got, _ = est_pose_batch(ctx, [ObjectSpec(ga, S.OBJ_PARAM, *ths), ObjectSpec(Generator(w_ok, "paper", ctx), S.OBJ_PARAM, *ths)], imgs, dets)
    assert ga.active_precision == "f32"
    alone = [ObjectSpec(Generator(w_bad, "paper", ctx, precision="f32"), S.OBJ_PARAM, *ths), ObjectSpec(Generator(w_ok, "paper", ctx), S.OBJ_PARAM, *ths)]
    for i, d in enumerate(dets):
        want, _ = est_pose_batch(ctx, [alone[d[1]]], imgs, [(d[0], 0, d[2], d[3])])
        assert key(got[i]) == key(want[0]), i


def test_forward_async_reports_through_the_context():
    """p2p_forward_async cannot return a range error (it only enqueues): p2p_ctx_range_event tells, once, and clears the flag."""
    import torch
    from pix2pose_amd.runtime import Context, Generator
    ctx = Context(0, max_batch=4)
    x = torch.from_numpy(_x(3)).cuda()
    y = torch.empty(3, 128, 128, 4, device="cuda")
    torch.cuda.synchronize()
    ok = Generator(W.synthetic_weights("paper", 3), "paper", ctx)
    ok.forward_device(x.data_ptr(), 3, y.data_ptr())
    assert ctx.range_event() == 0.0
    bad = Generator(_overflowing("paper"), "paper", ctx)
    bad.forward_device(x.data_ptr(), 3, y.data_ptr())
    assert ctx.range_event() > 6e4
    assert ctx.range_event() == 0.0                 # cleared by the query
    f32 = Generator(_overflowing("paper"), "paper", ctx, precision="f32")
    f32.forward_device(x.data_ptr(), 3, y.data_ptr())
    assert ctx.range_event() == 0.0                 # the fp32 arithmetic has no such limit and is not guarded


def test_range_event_travels_as_pose_range_through_the_gather():
    """a rank whose batch left the operand range must not broadcast ordinary-looking records.  collect_gathered takes the verdict
    BEFORE packing: this rank gets P2P_ERR_RANGE, its records travel with status = P2P_POSE_RANGE, nothing is handed over (a pre-filled pose
    array and a pre-zeroed mask buffer stay untouched), and the communicator keeps working.  Also: a NaN weight is caught (the running
    maximum and the ReLUs propagate NaN), not turned into finite garbage."""
    import ctypes as C
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Comm, Context, Generator, ObjectSpec, est_pose_submit
    ctx = Context(0, max_batch=16)
    ths = ([0.2, 0.3, 0.35], 0.2)
    sc = S.make_scene(3, seed=79)
    imgs = list(sc["images"])
    comm = Comm(ctx, 0, 1, Comm.unique_id())
    bad = ObjectSpec(Generator(_overflowing("resnet50"), "resnet50", ctx, precision="f16x3"), S.OBJ_PARAM, *ths)
    pend = est_pose_submit(ctx, [bad], imgs, sc["dets"], want_masks=True)
    poses = (_lib.Pose * 3)()
    for p in poses:
        p.n_inliers = 424242
    allp = (_lib.Pose * 8)()
    rc = _lib.lib().p2p_est_pose_collect_gathered(ctx.handle, comm.handle, pend.ticket, poses, 8, allp)
    assert rc == _lib.ERR_RANGE
    assert [allp[i].status for i in range(8)] == [_lib.POSE_RANGE] * 3 + [_lib.POSE_ABSENT] * 5
    assert all(p.n_inliers == 424242 for p in poses)                       # nothing handed over
    assert not pend.extras["valid_mask"].any()
    ok = ObjectSpec(Generator(W.synthetic_weights("resnet50", 3), "resnet50", ctx), S.OBJ_PARAM, *ths)
    own, allp2 = est_pose_submit(ctx, [ok], imgs, sc["dets"]).collect_gathered(comm, 8)      # the next batch is clean
    assert [allp2[i].status for i in range(3)] == [p.status for p in own] and all(s >= 0 for s in [p.status for p in own])
    comm.close()
    # NaN: a NaN in a ResNet-front BatchNorm shift would be zeroed by a plain fmaxf ReLU and vanish from every later layer
    w = dict(W.synthetic_weights("resnet50", 3))
    k = [n for n in w if n.startswith("res2b_2a") and n.endswith(".beta")][0]
    beta = w[k].copy(); beta[5] = np.nan; w[k] = beta
    with pytest.raises(_lib.P2PRangeError):
        Generator(w, "resnet50", ctx, precision="f16x3").predict(_x(12))   # 12 inputs: res2b runs on the fused kernel
    with pytest.raises(_lib.P2PRangeError):
        Generator(w, "resnet50", ctx, precision="f16x3").predict(_x(2))    # 2 inputs: the streaming route
