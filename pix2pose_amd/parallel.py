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


class PendingGather:
    """An all-gather of pose records in flight (gather_poses_async).  result() waits for it."""

    def __init__(self, work, out, keep):
        self._work, self._out, self._keep = work, out, keep

    def result(self) -> np.ndarray:
        if self._work is not None:
            self._work.wait()
        res = self._out.cpu().numpy()
        self._keep = None
        res = res[res[:, 0] >= 0]
        return res[np.argsort(res[:, 0], kind="stable")]


def gather_poses_async(records: np.ndarray, device=None, pad_to: int | None = None) -> PendingGather:
    """Start the all-gather of the per-rank records and return at once: the caller enqueues its next batch and picks the result up
    one step later (PendingGather.result()), so that a rank that is late for step i does not hold the other ranks' step i + 1 back.
    Shards may differ in length by padding to the largest (``pad_to``, or an all-reduce MAX -- which does block); padded rows carry
    id = -1 and are dropped.  On the "nccl" backend (= RCCL over xGMI) the records travel from device memory."""
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
        buf[:n] = torch.from_numpy(np.ascontiguousarray(records)).to(dev, non_blocking=True)
    out = torch.empty((world * pad_to, REC), dtype=torch.float64, device=dev)
    work = dist.all_gather_into_tensor(out, buf, async_op=True)
    return PendingGather(work, out, buf)


def gather_poses(records: np.ndarray, device=None, pad_to: int | None = None) -> np.ndarray:
    """All-gather the per-rank records (blocking).  Returns the concatenation over ranks, sorted by record id."""
    return gather_poses_async(records, device, pad_to).result()


def create_comm(ctx):
    """The C ABI's RCCL communicator for this rank (runtime.Comm), its 128-byte id drawn on rank 0 and handed round through the
    torch.distributed group the launcher set up -- the only thing torch.distributed is used for on this path; the gather itself is
    ncclAllGather inside libp2p_mi355.so (p2p_est_pose_collect_gathered), on device-resident records."""
    import torch.distributed as dist
    from .runtime import Comm
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [Comm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return Comm(ctx, rank, world, box[0])


class CabiPoseGather:
    """Per-step exchange of a detection stream's pose records through the C ABI: p2p_est_pose_collect_gathered (RCCL all-gather of the
    device-resident p2p_pose records over xGMI).  ``gather(pending, n_max)`` is collective: every rank calls it once per step, a rank
    without a batch passes ``pending = None`` (P2P_TICKET_NONE).  -> (own poses or [], numpy view [world * n_max] of _lib.POSE_DTYPE)."""

    def __init__(self, ctx):
        self.ctx, self.comm = ctx, create_comm(ctx)
        self.world, self.impl = self.comm.world, "C ABI ncclAllGather (%s)" % self.comm.library()

    def __call__(self, pending, n_max: int):
        from . import _lib
        from .runtime import collect_gathered_empty
        if pending is None:
            own, allp = [], collect_gathered_empty(self.ctx, self.comm, n_max)
        else:
            own, allp = pending.collect_gathered(self.comm, n_max)
        return own, np.frombuffer(allp, dtype=_lib.POSE_DTYPE, count=self.world * n_max)

    def close(self):
        self.comm.close()


class TorchPoseGather:
    """The same exchange through torch.distributed (gloo dry runs, several ranks on one device -- RCCL refuses that --, hosts without
    RCCL): the raw record bytes of each rank's collected batch, padded to n_max with status = POSE_ABSENT, in one all_gather."""

    def __init__(self, device=None):
        import torch.distributed as dist
        self.device, self.world = device, dist.get_world_size()
        self.impl = "torch.distributed all_gather_into_tensor (%s)" % dist.get_backend()

    def __call__(self, pending, n_max: int):
        import torch
        import torch.distributed as dist
        from . import _lib
        # like the C path, a local failure never strands the peers: the collective is entered with padding (or POSE_RANGE records) first
        own, err = [], None
        try:
            own = pending.collect() if pending is not None else []
            if len(own) > n_max:
                raise ValueError("the batch holds %d detections, n_max is %d" % (len(own), n_max))
        except Exception as e:      # noqa: BLE001 -- re-raised after the gather
            own, err = [], e
        buf = np.zeros(n_max, _lib.POSE_DTYPE)
        buf["status"] = _lib.POSE_ABSENT
        if isinstance(err, _lib.P2PRangeError) and pending is not None:
            buf["status"][:min(pending.n, n_max)] = _lib.POSE_RANGE
        if own:
            buf[:len(own)] = np.frombuffer(pending.pose_array, dtype=_lib.POSE_DTYPE, count=len(own))
        dev = self.device if self.device is not None else torch.device("cpu")
        send = torch.from_numpy(buf.view(np.uint8).copy()).to(dev)
        out = torch.empty(self.world * send.numel(), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out, send)
        if err is not None:
            raise err
        return own, np.frombuffer(out.cpu().numpy().tobytes(), dtype=_lib.POSE_DTYPE, count=self.world * n_max)

    def close(self):
        pass


def gathered_to_records(all_poses, world: int, n_max: int, ids_per_rank: int | None = None) -> np.ndarray:
    """The world * n_max p2p_pose records p2p_est_pose_collect_gathered returns -> [n, REC] records sorted by id
    (id = rank * ids_per_rank + position in the rank's batch), padding records dropped."""
    from . import _lib
    v = np.frombuffer(all_poses, dtype=_lib.POSE_DTYPE, count=world * n_max)
    keep = v["status"] != _lib.POSE_ABSENT
    stride = n_max if ids_per_rank is None else ids_per_rank
    ids = (np.arange(world * n_max) // n_max) * stride + (np.arange(world * n_max) % n_max)
    out = np.zeros((int(keep.sum()), REC), np.float64)
    vk = v[keep]
    out[:, 0] = ids[keep]
    out[:, 1] = vk["status"]
    out[:, 2] = vk["frac_inlier"]
    out[:, 3] = vk["n_inliers"]
    out[:, 4] = vk["n_init_mask"]
    out[:, 5] = vk["best_slot"]
    out[:, 6:15] = vk["R"]
    out[:, 15:18] = vk["t"]
    return out


def pin_rank_to_cpus(rank: int, world: int, usable: int | None = None):
    """Give every rank of a node its own slice of the CPUs this container may use (affinity mask, capped by the cgroup quota ``usable``):
    the GPU boxes show 256 CPUs but own a quota of 16, and eight Python ranks plus their HIP runtime threads otherwise migrate over
    all of them and throttle each other.  -> the CPU list the rank now runs on (or None where affinity cannot be set)."""
    import os
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if world <= 1 or not cpus:
        return cpus
    usable = min(len(cpus), usable or len(cpus))
    per = max(1, usable // world)
    stride = max(per, len(cpus) // world)              # spread the slices over the mask (distinct cores / CCDs where there are many)
    mine = cpus[rank * stride:rank * stride + per] or cpus[rank % len(cpus):rank % len(cpus) + 1]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine
