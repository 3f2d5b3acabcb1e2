"""Remaining boundary cases of the C ABI on the GPU: float32 frames, frames of different sizes in
one batch, 1 and 4 outlier thresholds (cfg_tless_paper / ros_config), device-pointer predict,
argument validation."""
import ctypes as C
import os

import numpy as np
import pytest

from pix2pose_amd import synthetic as S
from pix2pose_amd import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rig():
    from pix2pose_amd.runtime import Context, Generator
    ctx = Context(0, max_batch=32)
    return ctx, Generator(W.synthetic_weights("paper", 1), "paper", ctx)


def _key(p):
    return (p.status, p.n_inliers, p.n_init_mask, p.best_slot, tuple(p.bbox_t), tuple(p.R), tuple(p.t))


def test_float32_frames_equal_uint8_frames(rig):
    from pix2pose_amd.runtime import ObjectSpec, est_pose_batch
    ctx, gen = rig
    spec = ObjectSpec(gen, S.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2)
    rs = np.random.RandomState(2)
    a = rs.randint(0, 256, (200, 260, 3)).astype(np.uint8)
    b = rs.randint(0, 256, (300, 220, 3)).astype(np.uint8)          # different frame sizes in one batch
    dets = [(0, 0, [20, 30, 120, 140], S.LM_K), (1, 0, [150, 60, 290, 210], S.LM_K), (0, 0, [100, 180, 199, 259], S.LM_K)]
    pu, _ = est_pose_batch(ctx, [spec], [a, b], dets)
    pf, _ = est_pose_batch(ctx, [spec], [a.astype(np.float32), b.astype(np.float32)], dets)
    assert [_key(x) for x in pu] == [_key(x) for x in pf]


@pytest.mark.parametrize("ths", [[0.25], [0.1, 0.2, 0.3, 0.4]])
def test_one_and_four_thresholds_match_oracle(rig, ths):
    from oracle import est_pose_oracle as E
    from pix2pose_amd.runtime import ObjectSpec, est_pose_batch
    ctx, gen = rig
    spec = ObjectSpec(gen, S.OBJ_PARAM, ths, 0.15)
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (240, 320, 3)).astype(np.uint8)
    dets = [(0, 0, [40, 50, 160, 170], S.LM_K), (0, 0, [100, 150, 230, 300], S.LM_K)]
    poses, ex = est_pose_batch(ctx, [spec], [img], dets, debug=True)
    for i, d in enumerate(dets):
        dbg = {}
        ref = E.est_pose(img, d[2], lambda x, **kw: gen.predict(x), S.LM_K, S.OBJ_PARAM, ths, 0.15, debug=dbg)
        ok_ref = not (isinstance(ref[4], int) and ref[4] == -1)
        assert (poses[i].status == 0) == ok_ref
        assert poses[i].n_candidates == len(dbg.get("slots", []))
        assert list(poses[i].bbox_t) == ref[5].tolist()
        for c, slot in enumerate(dbg.get("slots", [])):
            assert ex["cand"][i, slot, 1] == dbg["cands"][c]["n_non_gray"]


def test_predict_device_pointers_and_argument_validation(rig):
    import torch
    from pix2pose_amd import _lib
    ctx, gen = rig
    x = (np.random.RandomState(4).randint(0, 256, (3, 128, 128, 3)).astype(np.float32) - 128) / 128
    d0, p0 = gen.predict(x)
    xd = torch.from_numpy(x).cuda()
    dd = torch.empty(3, 128, 128, 3, device="cuda")
    pd = torch.empty(3, 128, 128, 1, device="cuda")
    torch.cuda.synchronize()
    L = _lib.lib()
    _lib.check(L.p2p_predict(ctx.handle, gen.handle, xd.data_ptr(), 3, dd.data_ptr(), pd.data_ptr(), _lib.MEM_DEVICE), "p2p_predict")
    np.testing.assert_array_equal(dd.cpu().numpy(), d0)
    np.testing.assert_array_equal(pd.cpu().numpy(), p0)
    assert L.p2p_predict(ctx.handle, gen.handle, xd.data_ptr(), 3, dd.data_ptr(), pd.data_ptr(), 7) == -1     # bad mem flag
    assert L.p2p_predict(None, gen.handle, xd.data_ptr(), 3, dd.data_ptr(), pd.data_ptr(), 0) == -1
    assert b"bad" in L.p2p_last_error() or b"null" in L.p2p_last_error()
    h = C.c_void_p()
    assert L.p2p_ctx_create(99, 8, C.byref(h)) == -1 and b"out of range" in L.p2p_last_error()
    t = (_lib.Tensor * 1)()
    t[0].name, t[0].data, t[0].numel = b"conv1_1.kernel", x.ctypes.data_as(C.POINTER(C.c_float)), 10
    assert L.p2p_model_create(ctx.handle, t, 1, 0, C.byref(h)) == -3                                          # P2P_ERR_WEIGHTS
    assert b"conv1_1.kernel" in L.p2p_last_error()
    assert L.p2p_model_create(ctx.handle, t, 1, 5, C.byref(h)) == -1                                          # unknown backbone


@pytest.mark.gpu
def test_library_first_then_torch_in_one_process():
    """The library links the system HIP runtime, a PyTorch-ROCm wheel brings its own under the same soname: whichever is mapped
    first serves both, and torch sees no device through the system one.  The binding maps torch's runtime first when torch is
    installed, so a caller may create a context BEFORE touching torch.cuda (the harness does, eval_bop.run)."""
    import subprocess
    import sys
    code = ("from pix2pose_amd import runtime\n"
            "ctx = runtime.Context(0, max_batch=4)\n"
            "import torch\n"
            "x = torch.arange(8, dtype=torch.float32).cuda()\n"
            "print('sum', float((x * 2).sum()))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "sum 56.0" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
