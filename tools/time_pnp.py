import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pix2pose_amd import synthetic as S
from pix2pose_amd.runtime import default_context, pnp_ransac_batch
rs = np.random.RandomState(0)
Ks, objs, imgs = [], [], []
for p in range(768):
    n = 2500
    R = S.random_rotation(rs); t = np.array([rs.uniform(-60, 60), rs.uniform(-60, 60), rs.uniform(400, 1200)])
    P = rs.uniform(-1, 1, (n, 3)) * S.OBJ_PARAM[:3]
    uv = np.round(S.project(S.LM_K, R, t, P) + 0.3 * rs.randn(n, 2))
    k = int(0.1 * n); uv[:k] += rs.uniform(20, 60, (k, 2))
    Ks.append(S.LM_K); objs.append(P); imgs.append(uv)
for it in (100, 16, 1):
    ok, R, t, info, _ = pnp_ransac_batch(default_context(), Ks, objs, imgs, iterations=it)
    print(it, ok.sum(), info[:, 1].mean())
