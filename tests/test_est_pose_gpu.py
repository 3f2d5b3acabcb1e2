"""GPU parity of the batched est_pose pipeline (through p2p_est_pose_batch) against the numpy
oracle restatement of recognition.py:70-224.

Integer / index results (crop boxes, mask counts, correspondence counts, inlier counts, status,
selected candidate) must match exactly; poses within 1e-6 mm / 1e-4 deg of the oracle (north_star:
1 mm / 1 deg).  The generator output is either injected (synthetic NOCS scenes, meaningful PnP) or
taken from the GPU generator itself and fed to the oracle, so that the comparison isolates the
pipeline from fp32 summation-order noise (the generator has its own parity test)."""
import numpy as np
import pytest

from pix2pose_amd import weights as W
from pix2pose_amd import synthetic as synth

pytestmark = pytest.mark.gpu

TH_O = [0.2, 0.3, 0.35]     # cfg/cfg_bop2020.json:8-9
TH_I = 0.2


@pytest.fixture(scope="module")
def rig():
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec
    ctx = Context(0, max_batch=64, winograd="off")      # bit-identity tests across batch compositions below: one form of the 5x5 layers at every size
    gen = Generator(W.synthetic_weights("paper", 1), "paper", ctx)
    spec = ObjectSpec(gen, synth.OBJ_PARAM, TH_O, TH_I)
    return ctx, gen, spec


def _oracle(sc, i, predict, dbg, anti_aliasing=False):
    from oracle import est_pose_oracle as E
    img_i, _, bbox, K = sc["dets"][i]
    return E.est_pose(sc["images"][img_i], bbox, predict, K, sc["obj_param"], TH_O, TH_I, debug=dbg, anti_aliasing=anti_aliasing)


def _compare(p, ex, i, ref, dbg, injected):
    ok_ref = not (isinstance(ref[4], int) and ref[4] == -1)
    assert (p.status == 0) == ok_ref, (i, p.status, ref[1:5])
    np.testing.assert_array_equal(np.array(list(p.bbox_t)), ref[5])
    if "n_init_mask" in dbg:
        assert p.n_init_mask == dbg["n_init_mask"], i
    if "x1" in dbg:
        assert np.abs(ex["x1"][i] - dbg["x1"]).max() <= 1e-6, i
    if dbg.get("boxes2"):
        np.testing.assert_array_equal(ex["boxes2"][i], dbg["boxes2"][0])
        for c, slot in enumerate(dbg["slots"]):
            assert ex["cand"][i, slot, 0] == 1
            assert np.abs(ex["x2"][i, slot] - dbg["x2"][c]).max() <= 1e-6, (i, slot)
            cd = dbg["cands"][c]
            assert ex["cand"][i, slot, 1] == cd["n_non_gray"], (i, slot)
            if "n_valid" in cd and cd["n_valid"] >= 0:
                assert ex["cand"][i, slot, 2] == cd["n_valid"], (i, slot)
            if "n_inliers" in cd:
                assert ex["cand"][i, slot, 3] == cd["n_inliers"], (i, slot, cd.get("meta"), ex["cand"][i, slot])
    assert p.n_candidates == len(dbg.get("slots", []))
    if ok_ref:
        dt, dr = synth.pose_error(ref[2], ref[3], np.array(p.R).reshape(3, 3), np.array(p.t))
        assert dt < 1e-6 and dr < 1e-4, (i, dt, dr)
        assert abs(p.frac_inlier - ref[4]) < 1e-12
        assert p.best_slot == dbg["slots"][dbg["best"]]
        v1, v2, u1, u2 = ref[5]
        H, Wd = ref[1].shape
        np.testing.assert_array_equal(ex["valid_mask"][i][:H * Wd].reshape(H, Wd).astype(bool), ref[1])
        np.testing.assert_array_equal(ex["img_pred"][i][:(v2 - v1) * (u2 - u1) * 3].reshape(v2 - v1, u2 - u1, 3), ref[0])


def _run_injected(rig, sc, anti_aliasing=False):
    import torch
    from pix2pose_amd.runtime import est_pose_batch
    ctx, gen, spec = rig
    j1 = torch.from_numpy(sc["inject1"]).cuda()
    j2 = torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    poses, ex = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=j1.data_ptr(), inject2=j2.data_ptr(),
                               inject_slots=3, want_masks=True, debug=True, anti_aliasing=anti_aliasing)
    n_ok = 0
    for i in range(len(sc["dets"])):
        def predict(x, stage, slots=None, i=i):
            if stage == 1:
                m = sc["inject1"][i][None]
            else:
                m = sc["inject2"][i][slots]
            return [m[..., :3].copy(), m[..., 3:].copy()]
        dbg = {}
        ref = _oracle(sc, i, predict, dbg, anti_aliasing)
        _compare(poses[i], ex, i, ref, dbg, True)
        if poses[i].status == 0:
            n_ok += 1
            R, t = sc["gt"][i]
            bb = sc["dets"][i][2]
            if bb[2] - bb[0] >= 86:      # sanity vs ground truth where the object covers enough pixels
                dt, dr = synth.pose_error(R, t, np.array(poses[i].R).reshape(3, 3), np.array(poses[i].t))
                assert dt < 15.0 and dr < 3.0, (i, dt, dr)     # bounded by 8-bit XYZ quantisation + 10 % wrong coords
    return n_ok


def test_injected_scenes_128px_crops(rig):
    """BASELINE.json configs[2] shape: 128-px crops (every resize is the identity)."""
    sc = synth.make_scene(12, seed=3)
    assert _run_injected(rig, sc) >= 11


def test_injected_scenes_general_crop_sizes(rig):
    """Non-identity resizes (crop sides 74..300 px, up- and down-sampling), 'next' row f-3."""
    sc = synth.make_scene(10, seed=4, bbox_side=(50, 200))
    assert _run_injected(rig, sc) >= 8


def test_anti_aliased_resizes_small_crops(rig):
    """p2p_est_pose_opts.resize_anti_aliasing (scikit-image 0.17 - 0.18): crop sides 60..126 px -- the keep mask and the
    prob / img_pred / non_gray maps are Gaussian-filtered (scipy.ndimage.gaussian_filter in the oracle, csrc/resize_aa.hip on
    the device; float32 maps rounded per axis pass) before they shrink to the crop; network inputs are up-scaled (no filter).
    Same exactness as without the filter: masks, u8 images, counts, boxes bit for bit."""
    sc = synth.make_scene(8, seed=41, bbox_side=(40, 84))
    assert _run_injected(rig, sc, anti_aliasing=True) >= 6


def test_anti_aliased_resizes_large_crops(rig):
    """Crop sides 136..450 px: the stage-1 / stage-2 canvases cut from the frame are filtered ('mirror' border) before they
    shrink to 128x128; stage-2 re-crops below 128 px take the other branch in the same detection."""
    sc = synth.make_scene(8, seed=42, bbox_side=(91, 300))
    assert _run_injected(rig, sc, anti_aliasing=True) >= 6


def test_anti_aliased_resizes_sweep_with_border_clipping(rig):
    """A denser sweep of crop sides (66..390 px: Gaussian radii 0..5 on the way in, 0..4 on the way back) on small frames, so
    that many stage-1 / stage-2 squares hang over the frame border (zero canvas outside the paste window, 'mirror' filter
    border) -- against the oracle, whose filter is scipy's."""
    sc = synth.make_scene(20, seed=45, bbox_side=(44, 260), H=300, W=400, n_images=3)
    assert _run_injected(rig, sc, anti_aliasing=True) >= 12


def _exp_sensitive_sides():
    """Crop sides whose Gaussian weights depend on the exp() implementation: the library builds them with libm's exp (numpy <= 1.18, the
    reference's era, did the same), the scipy of this interpreter with numpy's SIMD exp, 1 ulp apart on some arguments."""
    import ctypes as C
    from pix2pose_amd import _lib
    L = _lib.lib()
    bad = set()
    for side in list(range(5, 128)) + list(range(129, 700)):
        w = (C.c_double * 256)()
        r = L.p2p_aa_weights(side, w)
        if r <= 0:
            continue
        n_in, n_out = (side, 128) if side > 128 else (128, side)
        sigma = (n_in / n_out - 1) / 2
        x = np.arange(-r, r + 1)
        phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        if not np.array_equal(np.array(w[:r + 1]), (phi / phi.sum())[r:]):
            bad.add(side)
    return bad


def test_generation_2_filters_the_bool_keep_mask(rig):
    """p2p_est_pose_opts.resize_anti_aliasing = 2 (scikit-image 0.15 / 0.16, the generation the reference's own python-3.5 image resolves to):
    every image is filtered as passed -- the BOOL keep mask of recognition.py:103 into a bool array (keep_filter_kernel), the float32 maps in
    float32 -- and warped in double.  Against the oracle (scipy's own gaussian_filter on the bool array), bit for bit, over crop sides 54 .. 390 px.
    A detection whose filters hit a crop side where libm's exp and this interpreter's numpy exp build different weights may legitimately differ
    (the bool filter turns one ulp into a whole mask): those are held to status and box only."""
    import torch
    from oracle import est_pose_oracle as E
    from pix2pose_amd.runtime import est_pose_batch
    ctx, gen, spec = rig
    bad = _exp_sensitive_sides()
    n_exact = n_sensitive = n_filtered_masks = 0
    for seed, sides in ((61, (36, 84)), (62, (40, 260)), (63, (70, 84))):
        sc = synth.make_scene(10, seed=seed, bbox_side=sides, H=300, W=400, n_images=3)
        j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
        torch.cuda.synchronize()
        poses, ex = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=j1.data_ptr(), inject2=j2.data_ptr(),
                                   inject_slots=3, want_masks=True, debug=True, anti_aliasing=2)
        poses1, _ = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3, anti_aliasing=1)
        for i in range(len(sc["dets"])):
            def predict(x, stage, slots=None, i=i):
                m = sc["inject1"][i][None] if stage == 1 else sc["inject2"][i][slots]
                return [m[..., :3].copy(), m[..., 3:].copy()]
            dbg = {}
            ref = _oracle(sc, i, predict, dbg, 2)
            b1 = E.get_boxes(sc["dets"][i][2], 300, 400)
            used = {b1.v2_ori - b1.v1_ori} | {b[1] - b[0] for b in dbg.get("boxes2", [])}
            if used & bad:
                n_sensitive += 1
                continue
            _compare(poses[i], ex, i, ref, dbg, True)
            n_exact += 1
            if b1.v2_ori - b1.v1_ori < 128 and (poses[i].n_inliers != poses1[i].n_inliers or poses[i].status != poses1[i].status):
                n_filtered_masks += 1
    assert n_exact >= 18, (n_exact, n_sensitive)
    assert n_filtered_masks >= 1          # the generation really is a different computation from 0.17 / 0.18
    with pytest.raises(Exception):
        est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"][:1], anti_aliasing=3)


@pytest.mark.parametrize("aa", [False, True])
def test_large_frames_large_crops(rig, aa):
    """1920x1080 frames, boxes of 300-420 px (crop sides 450-630 px, 70 000 correspondences per candidate: far beyond the 128-px
    configuration the bench runs) mixed with a 40-px box in the same batch: the correspondence storage, the RANSAC scoring and
    refit loops (no per-thread inlier mask above 16 384 points), the resize gathers and the anti-aliasing filter (radius up to 8)
    at sizes the reference meets on real images -- against the oracle, bit for bit as everywhere else."""
    a = synth.make_scene(4, seed=91, bbox_side=(300, 420), H=1080, W=1920, n_images=2)
    b = synth.make_scene(2, seed=92, bbox_side=(40, 48), H=1080, W=1920, n_images=2)
    sc = {"images": np.concatenate([a["images"], b["images"]]), "obj_param": a["obj_param"],
          "dets": a["dets"] + [(d[0] + 2, d[1], d[2], d[3]) for d in b["dets"]],
          "gt": a["gt"] + b["gt"], "inject1": np.concatenate([a["inject1"], b["inject1"]]), "inject2": np.concatenate([a["inject2"], b["inject2"]])}
    assert _run_injected(rig, sc, anti_aliasing=aa) >= 5


def test_huge_crop_takes_the_one_output_filter_path(rig):
    """A 640-700-px box in a 1920x1080 frame: crop side 960-1050 px, Gaussian radius 13-15 -- beyond the register window of the
    anti-aliasing filter (resize_aa.hip: AA_RMAX = 8), so the one-output-at-a-time path runs; 0.17 / 0.18 generation, against the oracle."""
    sc = synth.make_scene(1, seed=97, bbox_side=(640, 700), H=1080, W=1920, n_images=1)
    assert _run_injected(rig, sc, anti_aliasing=True) >= 1


def test_anti_aliasing_off_and_on_differ_and_identity_at_128(rig):
    import torch
    from pix2pose_amd.runtime import est_pose_batch
    ctx, gen, spec = rig
    sc = synth.make_scene(3, seed=43, bbox_side=(150, 200))
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    kw = dict(inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3, debug=True)
    off = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], **kw)[1]
    on = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], anti_aliasing=True, **kw)[1]
    assert np.abs(off["x1"] - on["x1"]).max() > 0.05          # random frames: the filter changes the network input a lot
    sc = synth.make_scene(3, seed=44)                          # 128-px crops: sigma = 0, the option changes nothing
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    kw = dict(inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3, debug=True, want_masks=True)
    p0, e0 = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], **kw)
    p1, e1 = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], anti_aliasing=True, **kw)
    for k in ("x1", "x2", "cand", "valid_mask", "img_pred"):
        np.testing.assert_array_equal(e0[k], e1[k])
    assert [tuple(a.R) for a in p0] == [tuple(b.R) for b in p1]


def test_real_generator_and_frame_border_crops(rig):
    """Boxes hanging over the frame border, tiny and degenerate boxes; the oracle consumes the
    GPU generator's own output so every downstream decision must agree exactly."""
    from pix2pose_amd.runtime import est_pose_batch
    ctx, gen, spec = rig
    rs = np.random.RandomState(0)
    H, Wd = 240, 320
    images = rs.randint(0, 256, (2, H, Wd, 3)).astype(np.uint8)
    boxes = [[-10, -20, 60, 70], [180, 250, 250, 330], [100, 100, 186, 186], [5, 5, 8, 8], [100, 100, 100, 100],
             [0, 0, 240, 320], [60, 200, 200, 290], [230, 310, 239, 319]]
    dets = [(i % 2, 0, b, synth.LM_K) for i, b in enumerate(boxes)]
    sc = {"images": images, "dets": dets, "obj_param": synth.OBJ_PARAM}
    poses, ex = est_pose_batch(ctx, [spec], list(images), dets, want_masks=True, debug=True)
    statuses = []
    for i in range(len(dets)):
        dbg = {}
        ref = _oracle(sc, i, lambda x, **kw: gen.predict(x), dbg)
        _compare(poses[i], ex, i, ref, dbg, False)
        statuses.append(poses[i].status)
    assert statuses[3] == 1 and statuses[4] == 1          # crop too small (recognition.py:78-79)


def test_shim_surface_matches_reference_tuple(rig):
    from pix2pose_amd.recognition import pix2pose
    ctx, gen, spec = rig
    p = pix2pose("synthetic:paper:1", synth.LM_K, 640, 480, synth.OBJ_PARAM, th_outlier=TH_O, th_inlier=TH_I,
                 backbone="paper", ctx=ctx)
    rgb = np.random.RandomState(1).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    out = p.est_pose(rgb, [100, 200, 186, 286])
    assert len(out) == 6
    img_pred, mask, R, t, frac, box = out
    assert box.shape == (4,) and box.dtype.kind == "i"
    if isinstance(frac, int) and frac == -1:            # failure sentinels (recognition.py:127,191)
        assert mask == -1 and R == -1 and t == -1
        # ... and the reference's first element there: the clipped (decode + 1) / 2 preview of stage 1 (:127) / of the last candidate (:191)
        assert img_pred.shape == (128, 128, 3) and img_pred.dtype == np.float32 and img_pred.min() >= 0 and img_pred.max() <= 1
        _, ex = __import__("pix2pose_amd.runtime", fromlist=["x"]).est_pose_batch(ctx, [p._spec()], [rgb], [(0, 0, [100, 200, 186, 286], synth.LM_K)], debug=True)
        d1 = p.generator_train.predict(ex["x1"])[0][0]
        st1 = np.clip((d1 + 1) / 2, 0, 1)
        if not ex["cand"][0, :, 0].any():               # no stage-2 input at all: the stage-1 preview, bit for bit
            np.testing.assert_array_equal(img_pred, st1)
    else:
        assert img_pred.dtype == np.uint8 and mask.dtype == bool and mask.shape == (480, 640)
        assert R.shape == (3, 3) and t.shape == (3,)
    out = p.est_pose(rgb, [5, 5, 7, 7])
    assert out[4] == -1 and out[0].shape == (1,)
    d, pr = p.generator_train.predict(np.zeros((2, 128, 128, 3)))
    assert d.shape == (2, 128, 128, 3) and pr.shape == (2, 128, 128, 1)
    with pytest.raises(ValueError):
        pix2pose("synthetic:paper:1", synth.LM_K, 640, 480, synth.OBJ_PARAM, backbone="vgg")


def test_mixed_objects_and_order_independence(rig):
    """Two objects (different backbones, different threshold counts) interleaved in one batch:
    results must equal the per-object runs, in the caller's order."""
    from pix2pose_amd.runtime import Generator, ObjectSpec, est_pose_batch
    ctx, gen, spec = rig
    gen2 = Generator(W.synthetic_weights("resnet50", 2), "resnet50", ctx)
    spec2 = ObjectSpec(gen2, synth.OBJ_PARAM * 1.3, [0.3, 0.5], 0.3)
    sc = synth.make_scene(6, seed=9, n_images=2)
    dets = [(d[0], i % 2, d[2], d[3]) for i, d in enumerate(sc["dets"])]
    both, _ = est_pose_batch(ctx, [spec, spec2], list(sc["images"]), dets)
    for o, sp in enumerate([spec, spec2]):
        sub = [(d[0], 0, d[2], d[3]) for d in dets if d[1] == o]
        alone, _ = est_pose_batch(ctx, [sp], list(sc["images"]), sub)
        k = 0
        for i, d in enumerate(dets):
            if d[1] != o:
                continue
            a, b = both[i], alone[k]
            k += 1
            assert (a.status, a.n_inliers, a.n_init_mask, a.best_slot) == (b.status, b.n_inliers, b.n_init_mask, b.best_slot)
            np.testing.assert_array_equal(np.array(a.R), np.array(b.R))
            np.testing.assert_array_equal(np.array(a.t), np.array(b.t))


def test_grouped_pass_same_backbone_objects(rig):
    """Several objects with the SAME backbone in one batch take the grouped path (one launch per layer,
    per-tile weight panels); results must equal the per-object runs exactly, for group sizes that are
    not multiples of anything (1, 2, 4 detections -> partial tiles on the 8x8 and dense layers)."""
    from pix2pose_amd.runtime import Generator, ObjectSpec, est_pose_batch
    ctx, gen, spec = rig
    gens = [gen] + [Generator(W.synthetic_weights("paper", 10 + k), "paper", ctx) for k in range(2)]
    specs = [ObjectSpec(g, synth.OBJ_PARAM * (1 + 0.1 * k), TH_O, TH_I) for k, g in enumerate(gens)]
    sc = synth.make_scene(7, seed=13, n_images=2)
    obj_of = [2, 0, 1, 1, 2, 2, 2]                       # interleaved; counts 1 / 2 / 4
    dets = [(d[0], obj_of[i], d[2], d[3]) for i, d in enumerate(sc["dets"])]
    both, _ = est_pose_batch(ctx, specs, list(sc["images"]), dets)
    for o, sp in enumerate(specs):
        idx = [i for i in range(7) if obj_of[i] == o]
        alone, _ = est_pose_batch(ctx, [sp], list(sc["images"]), [(dets[i][0], 0, dets[i][2], dets[i][3]) for i in idx])
        for k, i in enumerate(idx):
            a, b = both[i], alone[k]
            assert (a.status, a.n_inliers, a.n_init_mask, a.best_slot, tuple(a.bbox_t)) == (b.status, b.n_inliers, b.n_init_mask, b.best_slot, tuple(b.bbox_t)), (o, i)
            np.testing.assert_array_equal(np.array(a.R), np.array(b.R))
            np.testing.assert_array_equal(np.array(a.t), np.array(b.t))
    # the generator itself: grouped == per-object, bit for bit (through the debug taps of stage 2)
    _, ex_both = est_pose_batch(ctx, specs, list(sc["images"]), dets, debug=True)
    for o, sp in enumerate(specs):
        idx = [i for i in range(7) if obj_of[i] == o]
        _, ex_alone = est_pose_batch(ctx, [sp], list(sc["images"]), [(dets[i][0], 0, dets[i][2], dets[i][3]) for i in idx], debug=True)
        for k, i in enumerate(idx):
            np.testing.assert_array_equal(ex_both["x2"][i], ex_alone["x2"][k])      # depends on the stage-1 network output
            np.testing.assert_array_equal(ex_both["cand"][i], ex_alone["cand"][k])
