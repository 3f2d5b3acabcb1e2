"""The fused bottleneck blocks -- identity blocks res2b/c, res3b/c/d and the projection blocks res2a / res3a (last convolution = [2c | shortcut]
over [t_b ; x], stride 2 in res3a) -- (csrc/resblock.hip: 1x1 -> 3x3 -> 1x1 + residual in ONE launch, both intermediates in LDS; reference
resnet50_mod.py:40-73) against the three launches it replaces (igemm.hip / igemm_halo.hip / igemm.hip).  The claim is bit identity: every
output element is the same chain of MFMAs over the same K-step order, the intermediates are split into f16 halves by the same conversion, and
the epilogues evaluate the same expressions.  The three-launch route is selected with P2P_NO_FUSED_BLOCK=1, a route switch of the library's
development twin (pix2pose_amd/build.py dev_switches), read once per process -- hence the subprocesses; the fused side is the SHIPPED library."""
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
from pix2pose_amd.runtime import Context, Generator
out, ns, trained_like = sys.argv[1], [int(v) for v in sys.argv[2].split(",")], sys.argv[3] == "1"
x = (np.random.RandomState(5).randint(0, 256, (max(ns), 128, 128, 3)).astype(np.float32) - 128) / 128
x[1] *= 30.0                        # one sample far outside [-1, 1]
w = W.trained_like_weights("resnet50", 3) if trained_like else W.synthetic_weights("resnet50", 3)
g = Generator(w, "resnet50", Context(0, max_batch=max(ns), winograd="off"))
r = {}
for n in ns:
    dec, prob = g.predict(x[:n])
    r["dec%%d" %% n], r["prob%%d" %% n] = dec, prob
np.savez(out, **r)
""" % ROOT


def _run(tmp_path, tag, ns, env_extra, trained_like=False):
    out = str(tmp_path / ("%s.npz" % tag))
    env = dict(os.environ)
    env.update(env_extra)
    subprocess.run([sys.executable, "-c", _SCRIPT, out, ",".join(map(str, ns)), "1" if trained_like else "0"], check=True, env=env, cwd=ROOT, timeout=900)
    return np.load(out)


@pytest.mark.parametrize("trained_like", [False, True])
def test_fused_block_is_bit_identical_to_three_launches(tmp_path, trained_like):
    """48 inputs: every identity block of both stages on the fused kernel (res2: 384 workgroups, res3: 192); 12 inputs: res2 fused (96),
    res3 still on three launches (48 workgroups: below the small-launch threshold); 1 and 3 inputs: the streaming route throughout.
    Border patches (zero-padded t_a rows / columns) and interior ones are all in every image."""
    ns = [1, 3, 12, 48]
    a = _run(tmp_path, "fused%d" % trained_like, ns, {}, trained_like)
    b = _run(tmp_path, "three%d" % trained_like, ns, dev_switches(P2P_NO_FUSED_BLOCK=1, P2P_NO_FUSED_PROJ=1), trained_like)
    for k in a.files:
        assert np.isfinite(a[k]).all(), k
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    # and a sample's bits do not depend on the batch it travels in (fused at 48, partly fused at 12, streaming at 3 and 1)
    for n in (1, 3, 12):
        np.testing.assert_array_equal(a["dec48"][:n], a["dec%d" % n])
        np.testing.assert_array_equal(a["prob48"][:n], a["prob%d" % n])


def test_fused_block_in_a_mixed_object_pass():
    """Grouped generator passes (BASELINE.json configs[3]: detections of several objects in one batch, every workgroup looks its sample's
    weight panels up): 3 objects x 16 detections through est_pose_batch == the same detections object by object, bit for bit."""
    from pix2pose_amd import synthetic as S
    from pix2pose_amd import weights as W
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch
    ctx = Context(0, max_batch=256, winograd="off")
    specs = [ObjectSpec(Generator(W.synthetic_weights("resnet50", 10 + k), "resnet50", ctx), S.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2) for k in range(3)]
    sc = S.make_scene(48, seed=3)
    dets = [(d[0], i % 3, d[2], d[3]) for i, d in enumerate(sc["dets"])]
    mixed = est_pose_batch(ctx, specs, list(sc["images"]), dets)[0]
    for k in range(3):
        idx = [i for i in range(48) if i % 3 == k]
        alone = est_pose_batch(ctx, specs, list(sc["images"]), [dets[i] for i in idx])[0]
        for i, q in zip(idx, alone):
            p = mixed[i]
            assert (p.status, p.n_inliers, p.n_init_mask, tuple(p.bbox_t), tuple(p.R), tuple(p.t)) == \
                   (q.status, q.n_inliers, q.n_init_mask, tuple(q.bbox_t), tuple(q.R), tuple(q.t)), i
