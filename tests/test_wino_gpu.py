"""The 5x5 stride-1 decoder layers deconv1 / deconv2 / deconv3 (reference pix2pose_model/ae_model.py:207-211,217-220,227-230) in Winograd
F(4,5) form along the row axis (csrc/wino.hip: input transform + eight position GEMMs + inverse transform; 2.5x fewer MFMA products than
the direct kernels of csrc/igemm_halo.hip).

Unlike the other route pairs of the library the two forms do NOT compute the same bits (different products are formed), so the bar is the
ORACLE: network output within 1e-4 abs (north_star: 1e-3) on either route, both backbones, freshly-initialised and trained-like weight
statistics -- and the two routes within 6e-5 of each other (oracle/wino_study.py predicts 3.2e-5 from fp64 for this form).

The form is the caller's choice (p2p_ctx_set_winograd / Context(winograd=...)): "auto" (default) takes the Winograd route for passes of two
or more inputs (faster from 3 inputs up, 5 % slower at one); "always" / "off" pin one form at every size.  3 and 5 inputs (odd: the
two-samples-per-workgroup form of the 16x16 layer with a missing partner) reach the kernels here next to an
oracle that takes a second per input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

XYZ_TOL = 1e-4
ROUTE_TOL = 6e-5


def _run(backbone, fam, ns, winograd):
    from pix2pose_amd import weights as W
    from pix2pose_amd.runtime import Context, Generator
    x = _inputs(max(ns))
    w = W.trained_like_weights(backbone, 5) if fam == "trained_like" else W.synthetic_weights(backbone, 3)
    ctx = Context(0, max_batch=max(ns), winograd=winograd)
    g = Generator(w, backbone, ctx)
    r = {}
    for n in ns:
        ctx.profile(True)
        dec, prob = g.predict(x[:n])
        st = ctx.profile_read()
        ctx.profile(False)
        r["dec%d" % n], r["prob%d" % n] = dec, prob
        r["launches%d" % n] = np.array([s["launches"] for s in st])
    return r


def _inputs(n):
    return (np.random.RandomState(11).randint(0, 256, (n, 128, 128, 3)).astype(np.float32) - 128) / 128


@pytest.mark.parametrize("fam", ["synthetic", "trained_like"])
@pytest.mark.parametrize("backbone", ["resnet50", "paper"])
def test_winograd_route_matches_oracle_and_direct_route(backbone, fam):
    from oracle import ae_oracle as O
    from pix2pose_amd import weights as W
    ns = [3, 5]
    a = _run(backbone, fam, ns, "always")
    b = _run(backbone, fam, ns, "off")
    w = W.trained_like_weights(backbone, 5) if fam == "trained_like" else W.synthetic_weights(backbone, 3)
    d0, p0 = O.forward(w, _inputs(5), backbone)
    for n in ns:
        assert a["launches%d" % n][10] == 3 and a["launches%d" % n][11] == 3, a["launches%d" % n]       # the three layers, both kernels
        assert a["launches%d" % n][20] == 4 and a["launches%d" % n][21] == 4, a["launches%d" % n]       # conv4 / up1 / up2 / up3 in F(4,3) form ("always": at every size)
        assert b["launches%d" % n][10] == 0 and b["launches%d" % n][11] == 0 and b["launches%d" % n][20] == 0 and b["launches%d" % n][21] == 0
        for k, ref in (("dec", d0), ("prob", p0)):
            ya, yb = a["%s%d" % (k, n)], b["%s%d" % (k, n)]
            assert np.isfinite(ya).all()
            ea, eb, ed = np.abs(ya - ref[:n]).max(), np.abs(yb - ref[:n]).max(), np.abs(ya - yb).max()
            print("%s/%s n=%d %s: winograd-oracle %.2e  direct-oracle %.2e  winograd-direct %.2e" % (backbone, fam, n, k, ea, eb, ed))
            assert ea < XYZ_TOL and eb < XYZ_TOL and ed < ROUTE_TOL
    # a sample's bits do not depend on the batch it travels in as long as the route is the same
    np.testing.assert_array_equal(a["dec5"][:3], a["dec3"])
    np.testing.assert_array_equal(a["prob5"][:3], a["prob3"])


@pytest.mark.parametrize("backbone,n,precision", [("resnet50", 72, "f16x3"), ("resnet50", 24, "f32"), ("paper", 40, "f16x3"), ("paper", 20, "f32")])
def test_batched_routes_match_oracle(backbone, n, precision):
    """The kernels that carry the benchmark, held to the oracle DIRECTLY (not through the route-equivalence chain).  At 72 inputs EVERY
    convolution of a split-f16 resnet50 pass runs its batched / fused kernel -- asserted through the launch counts of p2p_profile_read: nothing
    but the split-K Dense layer on the small-launch route (igemm_stream_kernel), conv1 + pool, the seven fused bottleneck blocks, conv4 / up1 / up2 / up3 in
    Winograd F(4,3) form (wino3o.hip, wino3.hip), deconv1 / deconv2 / deconv3 in Winograd F(4,5) form, the merged heads.
    (Reference graph: pix2pose_model/ae_model.py:175-240.)"""
    from oracle import ae_oracle as O
    from pix2pose_amd import weights as W
    from pix2pose_amd.runtime import Context, Generator
    w = W.synthetic_weights(backbone, 4)
    ctx = Context(0, max_batch=n)
    g = Generator(w, backbone, ctx, precision=precision)
    x = _inputs(n)
    ctx.profile(True)
    dec, prob = g.predict(x)
    st = ctx.profile_read()
    ctx.profile(False)
    launches = [s["launches"] for s in st]
    print(backbone, precision, n, "launches per kernel family:", launches)
    if precision == "f16x3":
        assert launches[10] == 3 and launches[11] == 3, launches          # deconv1, deconv2, deconv3 in Winograd form
        assert launches[5] == 1, launches                                  # merged heads
    if precision == "f16x3" and backbone == "resnet50":
        assert launches[8] == 1 and launches[0] == 1, launches             # small-launch route: only dense_enc (its grid is the split-K factor x 2 up to 128 inputs); dense_dec batched
        assert launches[9] == 7, launches                                  # all seven bottleneck blocks fused
        # conv4, up1 (wino3o.hip), up2, up3 (wino3.hip) in Winograd F(4,3) form: none of them on the direct kernels
        assert launches[20] == 4 and launches[21] == 4 and launches[3] == 0 and launches[4] == 0 and launches[6] == 0, launches
    d0, p0 = O.forward(w, x, backbone)
    e = max(np.abs(dec - d0).max(), np.abs(prob - p0).max())
    print("%s/%s %d inputs: |d|max vs oracle %.2e" % (backbone, precision, n, e))
    assert e < XYZ_TOL


def test_winograd_route_in_a_mixed_object_pass():
    """Grouped generator passes (BASELINE.json configs[3]: detections of several objects in one batch, every workgroup of the Winograd
    kernels looks its sample's panel up; the two samples of a 16x16-layer workgroup belong to one object): 3 objects x 16 detections through
    est_pose_batch == the same detections object by object, bit for bit -- with the form pinned (winograd="always": all three layers in
    Winograd form whatever the batch size), a sample's bits do not depend on the batch it travels in."""
    from pix2pose_amd import synthetic as S
    from pix2pose_amd import weights as W
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch
    ctx = Context(0, max_batch=256, winograd="always")
    specs = [ObjectSpec(Generator(W.synthetic_weights("resnet50", 10 + k), "resnet50", ctx), S.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2) for k in range(3)]
    sc = S.make_scene(48, seed=3)
    dets = [(d[0], i % 3, d[2], d[3]) for i, d in enumerate(sc["dets"])]
    ctx.profile(True)
    mixed = est_pose_batch(ctx, specs, list(sc["images"]), dets)[0]
    st = ctx.profile_read()
    ctx.profile(False)
    assert st[10]["launches"] >= 3 and st[10]["launches"] == st[11]["launches"], [s["launches"] for s in st]
    assert st[20]["launches"] >= 4 and st[20]["launches"] == st[21]["launches"], [s["launches"] for s in st]      # conv4 and the transposed convolutions too (grouped wino3 / wino3o launches)
    for k in range(3):
        idx = [i for i in range(48) if i % 3 == k]
        alone = est_pose_batch(ctx, specs, list(sc["images"]), [dets[i] for i in idx])[0]
        for i, q in zip(idx, alone):
            p = mixed[i]
            assert (p.status, p.n_inliers, p.n_init_mask, tuple(p.bbox_t), tuple(p.R), tuple(p.t)) == \
                   (q.status, q.n_inliers, q.n_init_mask, tuple(q.bbox_t), tuple(q.R), tuple(q.t)), i
