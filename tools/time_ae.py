"""Quick timing of the generator forward pass on one GPU (development aid, not bench.py)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import numpy as np
import torch

from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Context, Generator

bb = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = Context(0, max_batch=chunk)
g = Generator(W.synthetic_weights(bb, 1), bb, ctx)
x = torch.rand(n, 128, 128, 3, device="cuda") * 2 - 1
y = torch.empty(n, 128, 128, 4, device="cuda")
torch.cuda.synchronize()
st = torch.cuda.ExternalStream(ctx.stream)
for _ in range(2):
    g.forward_device(x.data_ptr(), n, y.data_ptr())
ctx.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(st):
    e0.record(st)
    for _ in range(reps):
        g.forward_device(x.data_ptr(), n, y.data_ptr())
    e1.record(st)
ctx.synchronize()
ms = e0.elapsed_time(e1) / reps
gf = {"resnet50": 10.70, "paper": 12.58}[bb]
print("%s n=%d chunk=%d: %.3f ms/forward  %.1f crops/s  %.1f TFLOP/s" % (bb, n, chunk, ms, n / ms * 1e3, n * gf / ms))
