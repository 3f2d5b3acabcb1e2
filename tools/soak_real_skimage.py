"""Soak (build container only: needs /root/reference and /opt/conda/bin/python3.9): the oracle's anti_aliasing=True mode against the reference's est_pose run with the REAL scikit-image 0.18.3 (exact affine
matrix patch, cv2 / keras stood in) on many random detections.  Run under /opt/conda/bin/python3.9."""
import sys, warnings, zlib
warnings.filterwarnings("ignore")
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import make_reference_vectors as G
G.install_shims(real_skimage=True)
sys.path.insert(0, "/root/reference")
from pix2pose_model import recognition as ref
from oracle import est_pose_oracle as O
from pix2pose_amd import synthetic
_warps, Exact = G._exact_affine_patch()
_warps.AffineTransform = Exact
n_scenes = int(sys.argv[1]); seed0 = int(sys.argv[2])
crc = lambda a: int(zlib.crc32(np.ascontiguousarray(a).tobytes()))
n = bad = 0
for k in range(n_scenes):
    rs = np.random.RandomState(seed0 + k)
    lo = int(rs.randint(24, 200))
    sc = synthetic.make_scene(6, seed=seed0 + k, bbox_side=(lo, lo + int(rs.randint(1, 200))), outlier_frac=float(rs.uniform(0.1, 0.5)))
    for i, (img_i, _, bbox, K) in enumerate(sc["dets"]):
        p = object.__new__(ref.pix2pose)
        p.camK, p.res_x, p.res_y = np.asarray(K, float), 640, 480
        p.th_ransac, p.th_o, p.th_i = 3.0, G.TH_O, G.TH_I
        p.obj_scale, p.obj_ct = sc["obj_param"][:3], sc["obj_param"][3:]
        p.box_size, p.dist_coeff = 1.5, None
        p.generator_train = G._Predict(sc["inject1"][i], sc["inject2"][i])
        try:
            r = p.est_pose(sc["images"][img_i], np.asarray(bbox))
        except AssertionError:
            continue
        def predict(x, stage, slots=None, i=i):
            m = sc["inject1"][i][None] if stage == 1 else sc["inject2"][i][slots]
            return [m[..., :3].copy(), m[..., 3:].copy()]
        o = O.est_pose(sc["images"][img_i], bbox, predict, K, sc["obj_param"], G.TH_O, G.TH_I, anti_aliasing=True)
        n += 1
        ok_r = not (isinstance(r[1], int) and r[1] == -1); ok_o = not (isinstance(o[1], (int, np.integer)) and o[1] == -1)
        same = ok_r == ok_o and [int(v) for v in r[5]] == [int(v) for v in o[5]]
        if same and ok_r:
            same = crc(np.packbits(r[1])) == crc(np.packbits(o[1])) and crc(r[0]) == crc(o[0]) and np.array_equal(np.asarray(r[2]), np.asarray(o[2])) and np.array_equal(np.asarray(r[3]), np.asarray(o[3])) and float(r[4]) == float(o[4])
        if not same:
            bad += 1
            print("MISMATCH seed", seed0 + k, "det", i, list(bbox))
print("oracle vs reference est_pose with the real scikit-image 0.18.3 (exact affine matrix): %d detections, %d mismatches" % (n, bad))
