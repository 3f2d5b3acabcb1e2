"""Generator forward passes only: wall time per pass and the per-family kernel times of p2p_profile_read:
    python tools/time_pass.py [n_inputs] [reps] [backbone] [winograd: auto|off|always]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from pix2pose_amd import _lib, weights as W
from pix2pose_amd.runtime import Context, Generator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bb = sys.argv[3] if len(sys.argv) > 3 else "resnet50"
ctx = Context(0, max_batch=n, winograd=sys.argv[4] if len(sys.argv) > 4 else "auto")
g = Generator(W.synthetic_weights(bb, 1), bb, ctx)
x = (torch.randint(0, 256, (n, 128, 128, 3), device="cuda").float() - 128) / 128
y = torch.empty(n, 128, 128, 4, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    g.forward_device(x.data_ptr(), n, y.data_ptr())
ctx.synchronize()
t = time.perf_counter()
for _ in range(reps):
    g.forward_device(x.data_ptr(), n, y.data_ptr())
ctx.synchronize()
wall = (time.perf_counter() - t) / reps
ctx.profile(True)
for _ in range(reps):
    g.forward_device(x.data_ptr(), n, y.data_ptr())
st = ctx.profile_read()
fam = "  ".join("%s %.0f us" % (_lib.PROFILE_KERNELS[i][1].split("<")[0].replace("_kernel", ""), s["total_ms"] * 1e3 / reps) for i, s in enumerate(st) if s["launches"])
print("%d inputs: %.1f us per pass (%.0f inputs/s) | %s" % (n, wall * 1e6, n / wall, fam))
