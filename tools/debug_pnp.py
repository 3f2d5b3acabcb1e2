import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from pix2pose_amd import synthetic as S
from pix2pose_amd.runtime import default_context, pnp_ransac_batch
from oracle import est_pose_oracle as E, pnp_oracle as P

sc = S.make_scene(12, seed=3)
i, slot = 1, 1
TH_O, TH_I = [0.2, 0.3, 0.35], 0.2
def predict(x, stage, slots=None):
    m = sc["inject1"][i][None] if stage == 1 else sc["inject2"][i][slots]
    return [m[..., :3].copy(), m[..., 3:].copy()]
# replicate the oracle path to get correspondences of the candidate
img_i, _, bbox, K = sc["dets"][i]
dbg = {}
ref = E.est_pose(sc["images"][img_i], bbox, predict, K, sc["obj_param"], TH_O, TH_I, debug=dbg)
b = dbg["boxes2"][0]
v1o, v2o, u1o, u2o, v1, v2, u1, u2, vv1, vv2, uu1, uu2 = b
m = sc["inject2"][i][slot]
dec = m[..., :3].copy(); prob = m[..., 3]
gray = np.linalg.norm(dec, axis=2) < 0.3
dec[gray] = 0
pred = np.clip((dec + 1) / 2, 0, 1)
S2 = v2o - v1o
prob_ori = E.resize_bilinear(prob, (S2, S2), "constant", 1)[vv1:vv2, uu1:uu2]
pred_ori = (E.resize_bilinear(pred, (S2, S2), "constant", 0.5) * 255)[vv1:vv2, uu1:uu2]
ng = (E.resize_bilinear((~gray).astype(float), (S2, S2), "constant", 0) > 0.9)[vv1:vv2, uu1:uu2]
canvas = np.zeros(sc["images"][0].shape, np.uint8)
canvas[v1:v2, u1:u2] = pred_ori
obj, img, valid = E.correspondences(canvas, prob_ori, ng, v1, v2, u1, u2, sc["obj_param"][:3], sc["obj_param"][3:], TH_I)
print("n", len(obj))
for iters in (9,):
    ok, R, t, info, masks = pnp_ransac_batch(default_context(), [K], [obj], [img], iterations=iters, want_mask=True)
    ok0, R0, t0, inl0, meta = P.solve_pnp_ransac(obj, img, K, iterations=iters)
    print(iters, info, meta)
    gm = np.nonzero(masks[0])[0]
    diff = np.setxor1d(gm, inl0)
    print("diff idx", diff)
    lib = P.lib()
    rvec = np.zeros(3); tvec = np.zeros(3); idx = np.zeros(5, np.int32)
    dp = C.POINTER(C.c_double)
    Kf = np.ascontiguousarray(K, np.float64).reshape(9)
    lib.p2po_debug_hypothesis(Kf.ctypes.data_as(dp), np.ascontiguousarray(obj).ctypes.data_as(dp), np.ascontiguousarray(img).ctypes.data_as(dp),
                              len(obj), meta["best_iter"], rvec.ctypes.data_as(dp), tvec.ctypes.data_as(dp), idx.ctypes.data_as(C.POINTER(C.c_int)))
    Rh = P.rodrigues(rvec)
    print("hyp idx", idx, "rvec", rvec, "tvec", tvec)
    for d in diff:
        X = obj[d].astype(np.float32).astype(np.float64)
        x = Rh @ X + tvec
        pu = np.float32(x[0] / x[2] * Kf[0] + Kf[2]); pv = np.float32(x[1] / x[2] * Kf[4] + Kf[5])
        du = np.float32(img[d, 0]) - pu; dv = np.float32(img[d, 1]) - pv
        print("point", d, "err", np.float32(du * du + dv * dv), repr(float(np.float32(du*du+dv*dv))))
