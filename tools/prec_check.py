"""Accuracy and speed of the two generator precisions against the oracle (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ae_oracle as O
from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Context, Generator

ctx = Context(0, max_batch=256)
for bb in ("resnet50", "paper"):
    w = W.synthetic_weights(bb, 1)
    x = (np.random.RandomState(0).randint(0, 256, (3, 128, 128, 3)).astype(np.float32) - 128) / 128
    d0, p0 = O.forward(w, x, bb)
    for prec in ("f32", "f16x3"):
        g = Generator(w, bb, ctx, precision=prec)
        d, p = g.predict(x)
        xx = torch.rand(256, 128, 128, 3, device="cuda") * 2 - 1
        yy = torch.empty(256, 128, 128, 4, device="cuda")
        torch.cuda.synchronize()
        st = torch.cuda.ExternalStream(ctx.stream)
        g.forward_device(xx.data_ptr(), 256, yy.data_ptr()); ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(st)
            for _ in range(3):
                g.forward_device(xx.data_ptr(), 256, yy.data_ptr())
            e1.record(st)
        ctx.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("%-9s %-6s max|d-oracle| %.2e  max|p-oracle| %.2e  mean|d-oracle| %.2e   %.2f ms / 256 inputs = %.0f inputs/s"
              % (bb, prec, np.abs(d - d0).max(), np.abs(p - p0).max(), np.abs(d - d0).mean(), ms, 256 / ms * 1e3))
        g.close()
