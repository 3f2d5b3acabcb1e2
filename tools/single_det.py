"""One detection at a time through the drop-in shim (the reference's caller: tools/5_evaluation_bop_basic.py:289-304).
Prints the median latency; under rocprofv3 --kernel-trace, tools/trace_small.py-style listings come from the database.
Usage: python tools/single_det.py [calls]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from pix2pose_amd import synthetic, weights as W
from pix2pose_amd.recognition import pix2pose
from pix2pose_amd.runtime import Context

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
TH_O, TH_I = [0.2, 0.3, 0.35], 0.2
ctx = Context(0, max_batch=16)
wts = W.synthetic_weights("resnet50", 1)
sc = synthetic.make_scene(32, seed=1000)
inj1 = torch.from_numpy(sc["inject1"]).cuda()
inj2 = torch.from_numpy(sc["inject2"]).cuda()
torch.cuda.synchronize()
shim = pix2pose(dict(wts), synthetic.LM_K, 640, 480, synthetic.OBJ_PARAM, th_outlier=TH_O, th_inlier=TH_I, backbone="resnet50", ctx=ctx)
lat = []
for i in range(calls + 5):
    j = i % 32
    shim._inject = (inj1[j:j + 1].data_ptr(), inj2[j:j + 1].data_ptr(), 3)
    img_i, _, bbox, K = sc["dets"][j]
    shim.camK = K
    t1 = time.perf_counter()
    r = shim.est_pose(sc["images"][img_i], bbox)
    lat.append(time.perf_counter() - t1)
lat = np.array(lat[5:]) * 1e3
# where the host time goes: the C call alone (same arguments, marshalled once) against the whole shim call
from pix2pose_amd import runtime, _lib
import ctypes as C
spec = shim._spec()
img_i, _, bbox, K = sc["dets"][0]
rgb = sc["images"][img_i]
inj = dict(inject1=shim._inject[0], inject2=shim._inject[1], inject_slots=3)
objs, imgs, dets, opts, extras, keep = runtime._marshal([spec], [rgb], [(0, 0, [int(b) for b in bbox], K)], inj["inject1"], inj["inject2"], 3, True, None, 0, 0.0, 0.0)
poses = (_lib.Pose * 1)()
tc = []
for i in range(60):
    t1 = time.perf_counter()
    _lib.lib().p2p_est_pose_batch(ctx.handle, objs, 1, imgs, 1, dets, 1, poses, C.byref(opts))
    tc.append(time.perf_counter() - t1)
tm = []
for i in range(60):
    t1 = time.perf_counter()
    runtime._marshal([spec], [rgb], [(0, 0, [int(b) for b in bbox], K)], inj["inject1"], inj["inject2"], 3, True, None, 0, 0.0, 0.0)
    tm.append(time.perf_counter() - t1)
print("C call alone: median %.3f ms; marshalling alone: %.3f ms" % (np.median(tc[10:]) * 1e3, np.median(tm[10:]) * 1e3))
print("single est_pose: median %.3f ms  mean %.3f  p10 %.3f  p90 %.3f  (%d calls)" % (np.median(lat), lat.mean(), np.percentile(lat, 10), np.percentile(lat, 90), calls))
