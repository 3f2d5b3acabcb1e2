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
    """p2p_pose structs -> [n, REC] float64 records (float64 keeps R, t bit-exact)."""
    out = np.zeros((len(poses), REC), np.float64)
    for i, p in enumerate(poses):
        out[i, 0] = (ids[i] if ids is not None else base_id + i)
        out[i, 1] = p.status
        out[i, 2] = p.frac_inlier
        out[i, 3] = p.n_inliers
        out[i, 4] = p.n_init_mask
        out[i, 5] = p.best_slot
        out[i, 6:15] = list(p.R)
        out[i, 15:18] = list(p.t)
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
