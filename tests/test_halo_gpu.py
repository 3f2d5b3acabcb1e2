"""The specialised F16X3 kernels (halo-tiled igemm, heads, first layer) against the generic implicit-GEMM
path they replace: same layers, same split-f16 arithmetic, different tiling and summation order, so the
network outputs agree to fp32 summation noise.  The generic path is selected per process with
P2P_NO_HALO=1 (a route switch that exists in the development twin of the library only, read once per process: pix2pose_amd/build.py
dev_switches), hence the subprocesses; the other side of every comparison is the SHIPPED library."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pix2pose_amd.build import dev_switches

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Generator
backbone, precision, out = sys.argv[1], sys.argv[2], sys.argv[3]
import os
x = (np.random.RandomState(11).randint(0, 256, (int(os.environ.get("P2P_TEST_N", "5")), 128, 128, 3)).astype(np.float32) - 128) / 128
x[3] *= 40.0            # activations far outside [-1, 1]
g = Generator(W.synthetic_weights(backbone, 4), backbone, precision=precision)
dec, prob = g.predict(x)
np.savez(out, dec=dec, prob=prob)
""" % ROOT


def _run(tmp_path, backbone, precision, tag, env_extra):
    out = str(tmp_path / ("%s_%s_%s.npz" % (backbone, precision, tag)))
    env = dict(os.environ)
    env.update(env_extra)
    subprocess.run([sys.executable, "-c", _SCRIPT, backbone, precision, out], check=True, env=env, cwd=ROOT, timeout=600)
    return np.load(out)


@pytest.mark.parametrize("backbone", ["resnet50", "paper"])
def test_specialised_kernels_match_generic_path(tmp_path, backbone):
    a = _run(tmp_path, backbone, "f16x3", "halo", {})
    b = _run(tmp_path, backbone, "f16x3", "generic", dev_switches(P2P_NO_HALO=1))
    assert np.isfinite(a["dec"]).all() and np.isfinite(a["prob"]).all()
    big = 3                                     # the x40 sample: every pre-activation, and so every rounding difference, is 40x larger
    rest = [0, 1, 2, 4]
    assert np.abs(a["dec"][rest] - b["dec"][rest]).max() < 5e-5
    assert np.abs(a["prob"][rest] - b["prob"][rest]).max() < 5e-5
    assert np.abs(a["dec"][big] - b["dec"][big]).max() < 40 * 5e-5
    assert np.abs(a["prob"][big] - b["prob"][big]).max() < 40 * 5e-5


def test_specialised_kernels_track_fp32_mode(tmp_path):
    a = _run(tmp_path, "resnet50", "f16x3", "halo", {})
    c = _run(tmp_path, "resnet50", "f32", "f32", {})
    rest = [0, 1, 2, 4]
    assert np.abs(a["dec"][rest] - c["dec"][rest]).max() < 1e-4
    assert np.abs(a["prob"][rest] - c["prob"][rest]).max() < 1e-4
    assert np.abs(a["dec"][3] - c["dec"][3]).max() < 40 * 1e-4


def test_heads_two_ahead_schedule_is_bit_identical(tmp_path):
    """heads_halo_kernel<2> (P2P_HEADS_TWO_AHEAD=1: 8-row workgroups fetching two stages ahead) computes an output pixel with the same chain
    of MFMAs as the default 16-row schedule: identical bits."""
    a = _run(tmp_path, "resnet50", "f16x3", "default", {"P2P_TEST_N": "20"})           # >= 16 inputs: the large-launch variants of the kernel
    b = _run(tmp_path, "resnet50", "f16x3", "two_ahead", dev_switches(P2P_TEST_N=20, P2P_HEADS_TWO_AHEAD=1))
    assert np.array_equal(a["dec"], b["dec"]) and np.array_equal(a["prob"], b["prob"])
