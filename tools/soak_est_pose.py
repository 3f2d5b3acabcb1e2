"""Soak: the HIP est_pose pipeline against the oracle on many random detections (development aid; the committed tests hold fixed scenes).
    python tools/soak_est_pose.py [n_scenes] [resize generation 0|1|2] [seed0] [detections per scene: 8 = 24 PnP problems per call (the team form of the
    EPnP solvers, K splits of the small generator launches); 40 = 120 problems (the quad form, batched kernels)]
Every detection: status, returned box, valid mask, uint8 image identical; pose within 1e-6 mm / 1e-4 deg.
Generation 2 (scikit-image 0.15 / 0.16) filters the BOOL keep mask: a detection whose filters use a crop side where libm's exp (the library,
numpy <= 1.18) and this interpreter's numpy exp build different Gaussian weights may legitimately differ -- counted apart."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import est_pose_oracle as E
from pix2pose_amd import synthetic as S
from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
aa = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
n_per = int(sys.argv[4]) if len(sys.argv) > 4 else 8
TH_O, TH_I = [0.2, 0.3, 0.35], 0.2
ctx = Context(0, max_batch=max(64, 3 * n_per))
spec = ObjectSpec(Generator(W.synthetic_weights("paper", 1), "paper", ctx), S.OBJ_PARAM, TH_O, TH_I)
n_det = n_bad = n_ok = n_sens = n_sens_bad = 0
t0 = time.time()


def exp_sensitive_sides():
    import ctypes as C
    from pix2pose_amd import _lib
    L = _lib.lib()
    bad = set()
    for side in list(range(5, 128)) + list(range(129, 900)):
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


BAD = exp_sensitive_sides() if aa == 2 else set()
for k in range(n_scenes):
    rs = np.random.RandomState(seed0 + k)
    lo = int(rs.randint(24, 200))
    sc = S.make_scene(n_per, seed=seed0 + k, bbox_side=(lo, lo + int(rs.randint(1, 200))), outlier_frac=float(rs.uniform(0.1, 0.5)))
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    poses, ex = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3,
                               want_masks=True, anti_aliasing=aa)
    for i, p in enumerate(poses):
        def predict(x, stage, slots=None, i=i):
            m = sc["inject1"][i][None] if stage == 1 else sc["inject2"][i][slots]
            return [m[..., :3].copy(), m[..., 3:].copy()]
        img_i, _, bbox, K = sc["dets"][i]
        dbg = {}
        ref = E.est_pose(sc["images"][img_i], bbox, predict, K, sc["obj_param"], TH_O, TH_I, anti_aliasing=aa, debug=dbg)
        b1 = E.get_boxes(bbox, sc["images"].shape[1], sc["images"].shape[2])
        sens = bool(({b1.v2_ori - b1.v1_ori} | {b[1] - b[0] for b in dbg.get("boxes2", [])}) & BAD)
        n_sens += sens
        ok_ref = not (isinstance(ref[4], int) and ref[4] == -1)
        n_det += 1
        why = None
        if (p.status == 0) != ok_ref:
            why = "status %d vs %s" % (p.status, ok_ref)
        elif list(p.bbox_t) != [int(v) for v in ref[5]]:
            why = "box"
        elif ok_ref:
            n_ok += 1
            v1, v2, u1, u2 = ref[5]
            H, Wd = ref[1].shape
            dt, dr = S.pose_error(ref[2], ref[3], np.array(p.R).reshape(3, 3), np.array(p.t))
            if not np.array_equal(ex["valid_mask"][i][:H * Wd].reshape(H, Wd).astype(bool), ref[1]):
                why = "mask (%d px)" % int((ex["valid_mask"][i][:H * Wd].reshape(H, Wd).astype(bool) != ref[1]).sum())
            elif not np.array_equal(ex["img_pred"][i][:(v2 - v1) * (u2 - u1) * 3].reshape(v2 - v1, u2 - u1, 3), ref[0]):
                why = "img_pred"
            elif dt > 1e-6 or dr > 1e-4 or abs(p.frac_inlier - ref[4]) > 1e-12:
                why = "pose %.3g mm %.3g deg" % (dt, dr)
        if why and sens:
            n_sens_bad += 1
        elif why:
            n_bad += 1
            print("MISMATCH scene seed %d det %d bbox %s: %s" % (seed0 + k, i, list(bbox), why))
print("soak: %d detections (%d with a pose, %d per call), resize generation %d, %d mismatches; %d detections on exp-sensitive crop sides, %d of them differ; %.0f s"
      % (n_det, n_ok, n_per, aa, n_bad, n_sens, n_sens_bad, time.time() - t0))
sys.exit(1 if n_bad else 0)
