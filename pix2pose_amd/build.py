"""Builds libp2p_mi355.so (hipcc, gfx950 only) in-tree next to the sources.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libp2p_mi355.so")
SOURCES = ["igemm.hip", "igemm_halo.hip", "heads.hip", "conv1.hip", "misc_kernels.hip", "model.hip", "pipeline.hip", "pnp.hip"]
HEADERS = ["kernels.h", "model.h", "pipeline.h", os.path.join("..", "..", "include", "p2p_mi355.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _newer(a: str, b: str) -> bool:
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    objs = []
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    for s in srcs:
        o = os.path.splitext(s)[0] + ".o"
        objs.append(o)
        if force or _newer(s, o) or os.path.getmtime(o) < hdr_time:
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
