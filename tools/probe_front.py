"""Generator forward passes only (p2p_forward_async on device buffers), for rocprofv3 kernel traces / PMC passes of single kernels:
    python tools/probe_front.py [n_inputs] [passes] [backbone]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Context, Generator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bb = sys.argv[3] if len(sys.argv) > 3 else "resnet50"
ctx = Context(0, max_batch=n)
g = Generator(W.synthetic_weights(bb, 1), bb, ctx)
x = (torch.randint(0, 256, (n, 128, 128, 3), device="cuda").float() - 128) / 128
y = torch.empty(n, 128, 128, 4, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    g.forward_device(x.data_ptr(), n, y.data_ptr())
ctx.synchronize()
print("ok", n, reps)
