"""Forward-pass latency at small batch sizes (the reference's one-roi-at-a-time caller): ms per pass for n = 1, 2, 3, ...
Usage: python tools/time_small.py [backbone] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Context, Generator

bb = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ns = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64]
ctx = Context(0, max_batch=max(ns))
g = Generator(W.synthetic_weights(bb, 1), bb, ctx)
st = torch.cuda.ExternalStream(ctx.stream)
for n in ns:
    x = torch.rand(n, 128, 128, 3, device="cuda") * 2 - 1
    y = torch.empty(n, 128, 128, 4, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
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
    print("%s n=%-3d %.3f ms/pass  %.3f ms/input  (P2P_STREAM_WGS=%s)" % (bb, n, ms, ms / n, os.environ.get("P2P_STREAM_WGS", "default")), flush=True)
