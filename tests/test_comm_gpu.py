"""RCCL gather of the C ABI (include/p2p_mi355.h: p2p_comm_*, p2p_est_pose_collect_gathered; SURVEY.md section 8e -- no reference call
site) on the one GPU a test box has: a one-rank communicator goes through ncclGetUniqueId, ncclCommInitRank, the device-side packing of
the records into the caller's order, ncclAllGather on the tail stream and the single D2H -- everything but a second GPU."""
import ctypes as C

import numpy as np
import pytest

from pix2pose_amd import synthetic as S
from pix2pose_amd import weights as W

pytestmark = pytest.mark.gpu


def _key(p):
    return (p.status, p.n_inliers, p.n_init_mask, p.best_slot, tuple(p.bbox_t), tuple(p.R), tuple(p.t), p.frac_inlier)


def test_collect_gathered_equals_collect_in_caller_order():
    import torch
    from pix2pose_amd import _lib
    from pix2pose_amd.parallel import gathered_to_records, poses_to_records
    from pix2pose_amd.runtime import Comm, Context, Generator, ObjectSpec, est_pose_submit
    ctx = Context(0, max_batch=64)
    specs = [ObjectSpec(Generator(W.synthetic_weights("paper", 1 + k), "paper", ctx), S.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2) for k in range(3)]
    sc = S.make_scene(10, seed=31)
    dets = [(d[0], (7 * i) % 3, d[2], d[3]) for i, d in enumerate(sc["dets"])]        # objects interleaved: the batch is processed in sorted order
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    kw = dict(inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3)
    want = est_pose_submit(ctx, specs, list(sc["images"]), dets, **kw).collect()
    comm = Comm(ctx, 0, 1, Comm.unique_id())
    assert "librccl" in Comm.library()
    n_max = 16
    for _ in range(2):                                                                # twice: the communicator and its buffers are reused
        own, allp = est_pose_submit(ctx, specs, list(sc["images"]), dets, **kw).collect_gathered(comm, n_max)
        assert [_key(p) for p in own] == [_key(p) for p in want]
        assert [_key(allp[i]) for i in range(10)] == [_key(p) for p in want]          # caller order, bit for bit
        assert all(allp[i].status == _lib.POSE_ABSENT for i in range(10, n_max))      # padding
        rec = gathered_to_records(allp, 1, n_max)
        assert np.array_equal(rec, poses_to_records(want))
    # a batch larger than n_max is refused, not truncated
    pend = est_pose_submit(ctx, specs, list(sc["images"]), dets, **kw)
    with pytest.raises(_lib.P2PError):
        pend.collect_gathered(comm, 4)
    assert [_key(p) for p in pend.collect()] == [_key(p) for p in want]
    comm.close()


def test_empty_shard_and_errors_still_enter_the_collective():
    """A rank without a batch joins with P2P_TICKET_NONE (n_max padding records); an unknown ticket is reported AFTER the collective (its
    peers are not left blocked in ncclAllGather -- with one rank: the call returns instead of hanging and the communicator stays usable);
    the score_type-2 mask sums travel inside the gathered records."""
    import torch
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Comm, Context, Generator, ObjectSpec, collect_gathered_empty, est_pose_submit
    ctx = Context(0, max_batch=64)
    comm = Comm(ctx, 0, 1, Comm.unique_id())
    n_max = 8
    allp = collect_gathered_empty(ctx, comm, n_max)                    # before any batch was ever submitted on this context
    assert all(allp[i].status == _lib.POSE_ABSENT for i in range(n_max))
    spec = ObjectSpec(Generator(W.synthetic_weights("paper", 2), "paper", ctx), S.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2)
    sc = S.make_scene(5, seed=77)
    H, Wd = sc["images"].shape[1:3]
    masks = np.zeros((5, H, Wd), np.uint8)
    for i, d in enumerate(sc["dets"]):
        b = d[2]
        masks[i, b[0] + 8:b[2] - 6, b[1] + 5:b[3] - 9] = 1
    j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    kw = dict(inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3, det_masks=list(masks))
    pend = est_pose_submit(ctx, [spec], list(sc["images"]), sc["dets"], **kw)
    own, allp = pend.collect_gathered(comm, n_max)
    assert [p.status for p in own] == [0] * 5 and all(allp[i].status == _lib.POSE_ABSENT for i in range(5, n_max))
    for i in range(5):
        assert list(allp[i].mask_stats) == pend.extras["mask_stats"][i].tolist() == list(own[i].mask_stats)
        assert allp[i].mask_stats[0] > 0 and allp[i].mask_stats[1] >= allp[i].mask_stats[0]
    L = _lib.lib()
    poses = (_lib.Pose * n_max)()
    out = (_lib.Pose * n_max)()
    assert L.p2p_est_pose_collect_gathered(ctx.handle, comm.handle, 12345, poses, n_max, out) == -1      # unknown ticket: error after the gather
    assert all(out[i].status == _lib.POSE_ABSENT for i in range(n_max))                                  # ... which ran, with padding only
    assert b"12345" in L.p2p_last_error()
    allp = collect_gathered_empty(ctx, comm, n_max)                    # an empty step between two real ones
    assert all(allp[i].status == _lib.POSE_ABSENT for i in range(n_max))
    own2, allp2 = est_pose_submit(ctx, [spec], list(sc["images"]), sc["dets"], **kw).collect_gathered(comm, n_max)
    assert [_key(p) for p in own2] == [_key(p) for p in own]
    comm.close()


_INJECT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from pix2pose_amd import _lib, synthetic as S, weights as W
from pix2pose_amd.runtime import Comm, Context, Generator, ObjectSpec, collect_gathered_empty, est_pose_submit
ctx = Context(0, max_batch=16)
comm = Comm(ctx, 0, 1, Comm.unique_id())
spec = ObjectSpec(Generator(W.synthetic_weights("paper", 2), "paper", ctx), S.OBJ_PARAM, [0.2, 0.3, 0.35], 0.2)
sc = S.make_scene(3, seed=78)
j1, j2 = torch.from_numpy(sc["inject1"]).cuda(), torch.from_numpy(sc["inject2"]).cuda()
torch.cuda.synchronize()
pend = est_pose_submit(ctx, [spec], list(sc["images"]), sc["dets"], inject1=j1.data_ptr(), inject2=j2.data_ptr(), inject_slots=3)
L = _lib.lib()
poses, out = (_lib.Pose * 8)(), (_lib.Pose * 8)()
rc = L.p2p_est_pose_collect_gathered(ctx.handle, comm.handle, pend.ticket, poses, 8, out)
assert rc == -2, rc                                                               # P2P_ERR_HIP, reported AFTER the collective ...
assert b"joined the collective" in L.p2p_last_error(), L.p2p_last_error()
assert all(out[i].status == _lib.POSE_ABSENT for i in range(8))                   # ... which ran with this rank's padding records
allp = collect_gathered_empty(ctx, comm, 8)                                       # and the communicator is still usable
assert all(allp[i].status == _lib.POSE_ABSENT for i in range(8))
print("joined")
"""


def test_local_failure_before_the_collective_still_joins_it():
    """A failure between the buffers and ncclAllGather (memset / pack-kernel launch; injected here through a switch of the development
    twin) must not strand the peers: the rank joins with padding records and reports the error after the collective."""
    import os, subprocess, sys
    from pix2pose_amd.build import dev_switches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(dev_switches(P2P_COMM_INJECT_PACK_FAILURE=1))
    r = subprocess.run([sys.executable, "-c", _INJECT % root], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "joined" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_comm_argument_validation():
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Comm, Context
    ctx = Context(0, max_batch=8)
    with pytest.raises(ValueError):
        Comm(ctx, 0, 1, b"short")
    L = _lib.lib()
    h = C.c_void_p()
    assert L.p2p_comm_create(ctx.handle, 2, 2, b"\0" * 128, C.byref(h)) == -1        # rank out of range
    assert L.p2p_comm_create(None, 0, 1, b"\0" * 128, C.byref(h)) == -1
    assert L.p2p_comm_unique_id(None) == -1
