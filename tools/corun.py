"""Timing experiment (A/B build with -DP2P_TIMING_SWITCHES): does the memory-bound ResNet front of one pass run 'for free' on a slice of the
CUs while the MFMA-bound (power-limited) rest of another pass runs on the others?
    P2P_LIB=tools/ab/libp2p_ab.so python tools/corun.py [n_inputs] [front_cus]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fcus = int(sys.argv[2]) if len(sys.argv) > 2 else 64
from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Context, Generator

w = W.synthetic_weights("resnet50", 1)


def ctx(part, cus=None):
    os.environ["P2P_DEV_PART"] = str(part)
    if cus:
        os.environ["P2P_DEV_CUS"] = cus
    else:
        os.environ.pop("P2P_DEV_CUS", None)
    c = Context(0, max_batch=n)
    return c, Generator(w, "resnet50", c)


x = torch.randn(n, 128, 128, 3, device="cuda")
y = [torch.empty(n, 128, 128, 4, device="cuda") for _ in range(2)]


def timeit(jobs, reps=6):
    for c, g, yy in jobs:
        g.forward_device(x.data_ptr(), n, yy.data_ptr())
    for c, g, yy in jobs:
        c.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for c, g, yy in jobs:
            g.forward_device(x.data_ptr(), n, yy.data_ptr())
    for c, g, yy in jobs:
        c.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


full = ctx(0)
front = ctx(1)
rest = ctx(2)
t_full = timeit([(full[0], full[1], y[0])])
t_front = timeit([(front[0], front[1], y[0])])
t_rest = timeit([(rest[0], rest[1], y[0])])
print("n = %d: whole pass %.3f ms; front alone (all CUs) %.3f ms; rest alone (all CUs) %.3f ms; sum %.3f" % (n, t_full, t_front, t_rest, t_front + t_rest))
t_both = timeit([(front[0], front[1], y[0]), (rest[0], rest[1], y[1])])
print("front || rest on two unmasked streams: %.3f ms" % t_both)
for fc in (fcus, 32, 96, 128):
    fm = ctx(1, "0:%d" % fc)
    rm = ctx(2, "%d:256" % fc)
    tf, tr = timeit([(fm[0], fm[1], y[0])]), timeit([(rm[0], rm[1], y[1])])
    tb = timeit([(fm[0], fm[1], y[0]), (rm[0], rm[1], y[1])])
    print("front on %3d CUs alone %.3f ms, rest on %3d CUs alone %.3f ms, together %.3f ms  (serial on all CUs: %.3f)" % (fc, tf, 256 - fc, tr, tb, t_front + t_rest))
