"""Data-parallel sharding of detections over the GPUs of one node + the final pose gather.

The path shards perfectly: every detection is independent (the reference handles them one by
one, tools/5_evaluation_bop_basic.py:289-304), so each rank runs the whole pipeline on its shard
with no data-path collective.  The only exchange is one all-gather of fixed-size pose records
(R, t, score, ...) at the end -- RCCL over xGMI when the backend is "nccl", gloo on CPU in tests.
The reference has no distributed code at all; there is no call site to mirror.
"""
from __future__ import annotations

import numpy as np

REC = 20      # floats per record: id, status, score(frac_inlier), n_inliers, n_init_mask, best_slot, R[9], t[3], pad[2]


def shard_detections(detections, world: int):
    """Split detections into `world` contiguous shards, grouped by object first (weights
    locality), balanced by count (sizes differ by at most one).  -> (order, bounds): `order` is
    the permutation (indices into `detections`), rank r owns order[bounds[r]:bounds[r+1]]."""
    n = len(detections)
    order = sorted(range(n), key=lambda i: (detections[i][1], i))
    base, extra = divmod(n, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return order, bounds


def poses_to_records(poses, base_id=0, ids=None) -> np.ndarray:
    """p2p_pose structs -> [n, REC] float64 records (float64 keeps R, t bit-exact).  `poses` is a list of _lib.Pose or the
    ctypes array the binding fills (one vectorised view instead of a Python loop: this runs once per step on every rank)."""
    from . import _lib
    n = len(poses)
    out = np.zeros((n, REC), np.float64)
    if n == 0:
        return out
    if isinstance(poses, (list, tuple)):
        arr = (_lib.Pose * n)(*poses)
    else:
        arr = poses
    v = np.frombuffer(arr, dtype=_lib.POSE_DTYPE, count=n)
    out[:, 0] = np.asarray(ids, np.float64) if ids is not None else base_id + np.arange(n)
    out[:, 1] = v["status"]
    out[:, 2] = v["frac_inlier"]
    out[:, 3] = v["n_inliers"]
    out[:, 4] = v["n_init_mask"]
    out[:, 5] = v["best_slot"]
    out[:, 6:15] = v["R"]
    out[:, 15:18] = v["t"]
    return out


def gather_poses(records: np.ndarray, device=None, pad_to: int | None = None) -> np.ndarray:
    """All-gather the per-rank records.  Shards may differ in length by padding to the largest
    (``pad_to`` or an all-reduce MAX); padded rows carry id = -1 and are dropped.  Returns the
    concatenation over ranks, sorted by record id."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n = records.shape[0]
    dev = device if device is not None else torch.device("cpu")
    if pad_to is None:
        m = torch.tensor([n], dtype=torch.int64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        pad_to = int(m.item())
    buf = torch.full((pad_to, REC), -1.0, dtype=torch.float64, device=dev)
    if n:
        buf[:n] = torch.from_numpy(np.ascontiguousarray(records)).to(dev)
    out = torch.empty((world * pad_to, REC), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf)
    res = out.cpu().numpy()
    res = res[res[:, 0] >= 0]
    return res[np.argsort(res[:, 0], kind="stable")]
