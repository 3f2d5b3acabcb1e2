"""Phase time stamps of wino3_gemm_kernel's tiles (A/B build -DP2P_W3_STAMPS, loaded through P2P_LIB): where a tile's time goes.
    P2P_LIB=tools/ab/libp2p_stamps.so python tools/w3_stamps.py [n_inputs]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pix2pose_amd import _lib, weights as W
from pix2pose_amd.runtime import Context, Generator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = Context(0, max_batch=n)
g = Generator(W.synthetic_weights("resnet50", 1), "resnet50", ctx)
x = (torch.randint(0, 256, (n, 128, 128, 3), device="cuda").float() - 128) / 128
y = torch.empty(n, 128, 128, 4, device="cuda")
for _ in range(3):
    g.forward_device(x.data_ptr(), n, y.data_ptr())
ctx.synchronize()
L = C.CDLL(os.environ["P2P_LIB"])
buf = np.zeros(8 * 16 * 16, np.uint64)
assert L.p2p_dbg_w3_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
s = buf.reshape(8, 16, 16).astype(np.float64) / 100.0        # us (100 MHz); the LAST wino3 launch of the pass = up3
for b in range(3):
    print("workgroup %d: per tile, us after the tile's start: K loop start | end of the K loop of waves 0..11 (wave = 2 position + unit) | barrier released | next tile" % (b * 32))
    for t in range(15):
        a = s[b, t]
        nxt = s[b, t + 1, 12]
        if a[12] == 0 or nxt == 0:
            break
        print("   tile %2d: %5.2f | %s | %6.2f | %6.2f" % (t, a[13] - a[12], " ".join("%5.1f" % (a[w] - a[12]) for w in range(12)), a[14] - a[12], nxt - a[12]))
