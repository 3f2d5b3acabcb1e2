"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Compiles the oracle's C restatements (gcc, OpenMP) into oracle/_build/libp2p_oracle.so.
Building the checker is not using it: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg load the result.

No reference build (oracle/_ref) exists for this project: the reference is pure Python on
top of un-vendored TensorFlow/Keras/OpenCV/scikit-image, none of which is installed, so
there is nothing to compile from /root/reference ("Oracle").
"""
from __future__ import annotations

import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libp2p_oracle.so")
STAMP = os.path.join(OUT_DIR, "host.stamp")
SOURCES = ["ae_layers.c", "pnp_oracle.c"]


def _host_tag() -> str:
    """-march=native output must not travel to a host with a different ISA."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha1(line.encode()).hexdigest()
    except OSError:
        pass
    return "unknown"


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    tag = _host_tag()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        newest = max(os.path.getmtime(s) for s in srcs)
        with open(STAMP) as f:
            same_host = f.read().strip() == tag
        if same_host and os.path.getmtime(LIB) >= newest:
            return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    # -ffp-contract=off: keep a*b+c as two roundings so the oracle's double arithmetic does
    # not depend on whether the host CPU has FMA.
    cmd = ["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
           "-std=c11", "-o", LIB] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(tag)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
