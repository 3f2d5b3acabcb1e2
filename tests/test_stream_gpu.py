"""The small-launch route (igemm_stream.hip: one wave per 32x32 output tile, operands streamed to registers) against the batched
kernels it stands in for.  The claim is stronger than a tolerance: every output element is the same chain of MFMAs over the same
K-step order whichever kernel computes it, so the generator output of a crop is BIT-IDENTICAL alone (streaming route), inside a
large batch (batched kernels), and with the streaming route switched off (P2P_STREAM_WGS=0: a route switch of the library's development
twin, read once per process -- hence the subprocess; pix2pose_amd/build.py dev_switches).  This is what lets the reference's one-roi-at-a-time caller (tools/5_evaluation_bop_basic.py:289-304) and a pooled
batch agree to the last bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pix2pose_amd.build import dev_switches
from pix2pose_amd import weights as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from pix2pose_amd import weights as W
from pix2pose_amd.runtime import Context, Generator
backbone, out = sys.argv[1], sys.argv[2]
x = (np.random.RandomState(31).randint(0, 256, (4, 128, 128, 3)).astype(np.float32) - 128) / 128
r = {}
for mode in ("off", "always"):      # the form of the 5x5 layers pinned: under "auto" the small route also splits K (bits depend on the pass size)
    g = Generator(W.synthetic_weights(backbone, 3), backbone, Context(0, max_batch=8, winograd=mode))
    for n in (1, 3, 4):
        dec, prob = g.predict(x[:n])
        r["dec%%d_%%s" %% (n, mode)], r["prob%%d_%%s" %% (n, mode)] = dec, prob
np.savez(out, **r)
""" % ROOT


def _run(tmp_path, backbone, tag, env_extra):
    out = str(tmp_path / ("%s_%s.npz" % (backbone, tag)))
    env = dict(os.environ)
    env.update(env_extra)
    subprocess.run([sys.executable, "-c", _SCRIPT, backbone, out], check=True, env=env, cwd=ROOT, timeout=600)
    return np.load(out)


@pytest.mark.parametrize("backbone", ["resnet50", "paper"])
def test_streaming_route_is_bit_identical_to_batched_kernels(tmp_path, backbone):
    a = _run(tmp_path, backbone, "stream", {})
    b = _run(tmp_path, backbone, "batched", dev_switches(P2P_STREAM_WGS=0))
    for k in a.files:
        assert np.isfinite(a[k]).all()
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("winograd", ["off", "always"])
@pytest.mark.parametrize("backbone", ["resnet50", "paper"])
def test_crop_alone_equals_crop_inside_a_large_batch(backbone, winograd):
    """40 inputs put every big layer on the batched kernels (more than 64 workgroups); one input runs the streaming route.  The form of the
    5x5 decoder layers is pinned (direct / Winograd at every size): with the default "auto" a sample's bits depend on the batch SIZE."""
    from pix2pose_amd.runtime import Context, Generator
    ctx = Context(0, max_batch=48, winograd=winograd)
    g = Generator(W.synthetic_weights(backbone, 7), backbone, ctx)
    x = (np.random.RandomState(8).randint(0, 256, (40, 128, 128, 3)).astype(np.float32) - 128) / 128
    dec, prob = g.predict(x)
    for i in (0, 17, 39):
        d1, p1 = g.predict(x[i:i + 1])
        np.testing.assert_array_equal(dec[i:i + 1], d1)
        np.testing.assert_array_equal(prob[i:i + 1], p1)
    d3, p3 = g.predict(x[20:23])            # the stage-2 pass of one detection: K = 3 inputs
    np.testing.assert_array_equal(dec[20:23], d3)
    np.testing.assert_array_equal(prob[20:23], p3)


@pytest.mark.parametrize("inject", [True, False])
def test_est_pose_alone_equals_est_pose_inside_a_batch(inject):
    """The whole call, not only the generator: a detection handed over alone (the reference's loop: small-launch generator kernels, segmented
    glue kernels, two-launch correspondence build) returns the very same record -- R, t, counts, box, selected candidate -- as inside a batch
    of 40 (batched kernels throughout).  With injected decoder maps (a pose worth comparing) and with the random-weight generator's own
    output driving the masks (whatever it yields must not depend on the batch)."""
    import torch
    from pix2pose_amd import synthetic as S
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch
    ctx = Context(0, max_batch=160, winograd="off")
    gen = Generator(W.synthetic_weights("resnet50", 11), "resnet50", ctx)
    spec = ObjectSpec(gen, S.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2)
    sc = S.make_scene(40, seed=77, bbox_side=(60, 140))
    j1 = torch.from_numpy(sc["inject1"]).cuda()
    j2 = torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()

    def key(p):
        return (p.status, p.n_inliers, p.n_init_mask, p.best_slot, p.n_candidates, p.ransac_iters, tuple(p.bbox_t), tuple(p.R), tuple(p.t), p.frac_inlier)

    kw = dict(inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3) if inject else {}
    batch, _ = est_pose_batch(ctx, [spec], list(sc["images"]), sc["dets"], **kw)
    if inject:
        assert sum(p.status == 0 for p in batch) >= 36
    for i in (0, 7, 23, 39):
        kw1 = dict(inject1=j1[i:i + 1].data_ptr(), inject2=j2[i:i + 1].data_ptr(), inject_slots=3) if inject else {}
        one, _ = est_pose_batch(ctx, [spec], list(sc["images"]), [sc["dets"][i]], **kw1)
        assert key(one[0]) == key(batch[i]), i


@pytest.mark.parametrize("backbone", ["resnet50", "paper"])
def test_small_pass_with_split_k_stays_at_the_single_chain(backbone):
    """Under "auto" the long-K layers of a pass over a few inputs split K over the chip and a reduction applies the epilogue: conv4 on the
    streaming kernel (model.hip run_conv_small), the stride-1 5x5 layers in Winograd form with their channel slices cut into ranges (try_wino:
    a launch of 2-24 workgroups otherwise).  Another summation order, not another result: the one- and three-input passes (the two passes of a
    single est_pose call) stay within 1e-4 of the direct single-chain form ("off"; tests/test_ae_gpu.py and tests/test_wino_gpu.py hold
    every form to the oracle at 1e-4), and repeat bit for bit (the partial sums are added in a fixed order)."""
    from pix2pose_amd.runtime import Context, Generator
    w = W.synthetic_weights(backbone, 5)
    x = (np.random.RandomState(12).randint(0, 256, (3, 128, 128, 3)).astype(np.float32) - 128) / 128
    ga = Generator(w, backbone, Context(0, max_batch=4, winograd="auto"))
    go = Generator(w, backbone, Context(0, max_batch=4, winograd="off"))
    for n in (1, 3):
        d, p = ga.predict(x[:n])
        d2, p2 = ga.predict(x[:n])
        np.testing.assert_array_equal(d, d2)
        np.testing.assert_array_equal(p, p2)
        r, q = go.predict(x[:n])
        assert np.abs(d - r).max() < 1e-4 and np.abs(p - q).max() < 1e-4
