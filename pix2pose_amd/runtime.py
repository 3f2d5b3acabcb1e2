"""Python handles over the C ABI: a per-GPU context and per-object generator networks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import weights as W


class Context:
    """One GPU's pipeline context (stream + workspaces).  Not thread-safe (p2p_mi355.h)."""

    WINOGRAD = {"off": 0, "auto": 1, "always": 2}

    def __init__(self, device: int = 0, max_batch: int = 256, winograd: str = "auto"):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().p2p_ctx_create(device, max_batch, C.byref(self._h)), "p2p_ctx_create")
        self.device = device
        self.max_batch = max_batch
        if winograd != "auto":
            self.set_winograd(winograd)

    def set_winograd(self, mode: str):
        """Form of the 5x5 layers (deconv1-3, up1-3, conv4) in split-f16 passes (p2p_ctx_set_winograd): "auto" (default) = the fastest form at
        every pass size (Winograd F(4,5) at every size, F(4,3) from 5 / 9 inputs up, K splits for launches that would fill a fraction of the chip) -- a sample's bits depend on the pass SIZE;
        "off" / "always" = one form at every size."""
        _lib.check(_lib.lib().p2p_ctx_set_winograd(self._h, self.WINOGRAD[mode]), "p2p_ctx_set_winograd")

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return _lib.lib().p2p_ctx_stream(self._h) or 0

    def synchronize(self):
        _lib.check(_lib.lib().p2p_ctx_synchronize(self._h), "p2p_ctx_synchronize")

    def range_event(self) -> float:
        """Largest activation magnitude beyond the split-f16 operand range that direct forward calls (Generator.forward_device) stored since
        the last query; 0.0 = none.  Synchronises the context stream (p2p_ctx_range_event)."""
        v = C.c_float(0.0)
        _lib.check(_lib.lib().p2p_ctx_range_event(self._h, C.byref(v)), "p2p_ctx_range_event")
        return float(v.value)

    def profile(self, on: bool):
        """Bracket every implicit-GEMM launch with HIP events on the context stream."""
        _lib.check(_lib.lib().p2p_profile_enable(self._h, 1 if on else 0), "p2p_profile_enable")

    def profile_read(self, reset=True):
        """-> list of PROFILE_SLOTS dicts (kernel families, see PROFILE_KERNELS): launches, total_ms, algo_flops, algo_bytes."""
        st = (_lib.KernelStats * _lib.PROFILE_SLOTS)()
        _lib.check(_lib.lib().p2p_profile_read(self._h, st, 1 if reset else 0), "p2p_profile_read")
        return [{"launches": int(s.launches), "total_ms": float(s.total_ms), "algo_flops": float(s.algo_flops), "algo_bytes": float(s.algo_bytes)} for s in st]

    def close(self):
        if self._h:
            _lib.lib().p2p_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Generator:
    """Drop-in for the Keras model held in ``pix2pose.generator_train``
    (reference recognition.py:21-26): ``predict(x) -> [decode, prob]`` (recognition.py:84,129)."""

    def __init__(self, weights: dict, backbone: str, ctx: Context | None = None, precision: str = "f16x3"):
        """precision: 'f32' (fp32 matrix instructions), 'f16x3' (fp32 emulated with three split-f16
        MFMAs per product block, fp32 accumulate; see include/p2p_mi355.h) or 'auto' (f16x3 with an fp32 twin the object
        falls back to when an activation leaves the f16 operand range; without a twin such a pass raises P2PRangeError)."""
        if backbone not in _lib.BACKBONE:
            raise ValueError("unknown backbone %r" % (backbone,))
        if precision not in _lib.PRECISION:
            raise ValueError("unknown precision %r" % (precision,))
        self.precision = precision
        W.check_weights(backbone, weights)
        self.ctx = ctx or default_context()
        self.backbone = backbone
        specs = W.tensor_specs(backbone)
        arr = (_lib.Tensor * len(specs))()
        keep = []
        for i, (name, _) in enumerate(specs):
            a = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(a)
            arr[i].name = name.encode()
            arr[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
            arr[i].numel = a.size
        self._h = C.c_void_p()
        _lib.check(_lib.lib().p2p_model_create_ex(self.ctx.handle, arr, len(specs), _lib.BACKBONE[backbone],
                                                  _lib.PRECISION[precision], C.byref(self._h)), "p2p_model_create_ex")

    @property
    def handle(self):
        return self._h

    def predict(self, x):
        """x [N,128,128,3] float (float64 accepted, cast to float32 like Keras does)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 4 or x.shape[1:] != (128, 128, 3):
            raise ValueError("expected input of shape [N,128,128,3], got %r" % (x.shape,))
        n = x.shape[0]
        xyz = np.empty((n, 128, 128, 3), np.float32)
        prob = np.empty((n, 128, 128, 1), np.float32)
        _lib.check(_lib.lib().p2p_predict(self.ctx.handle, self._h, x.ctypes.data, n, xyz.ctypes.data,
                                          prob.ctypes.data, _lib.MEM_HOST), "p2p_predict")
        return [xyz, prob]

    @property
    def active_precision(self) -> str:
        """'f32' or 'f16x3': what the next pass computes in (an 'auto' generator reports 'f16x3' until a range event)."""
        return "f32" if _lib.lib().p2p_model_precision(self._h) == 0 else "f16x3"

    def forward_device(self, x_ptr: int, n: int, xyzp_ptr: int):
        """Asynchronous forward on device pointers (ints), interleaved [n,128,128,4] output."""
        _lib.check(_lib.lib().p2p_forward_async(self.ctx.handle, self._h, x_ptr, n, xyzp_ptr), "p2p_forward_async")

    def close(self):
        if self._h:
            _lib.lib().p2p_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------------------------
# batched pose estimation (C ABI: p2p_est_pose_batch / p2p_pnp_ransac_batch)
# --------------------------------------------------------------------------------------------
class ObjectSpec:
    """What one reference ``pix2pose`` instance holds per object (recognition.py:10-26)."""

    def __init__(self, generator: Generator, obj_param, th_outlier=(0.1, 0.2, 0.3), th_inlier=0.1, box_size=1.5):
        th_outlier = list(th_outlier)
        if not 1 <= len(th_outlier) <= _lib.MAX_TH:
            raise ValueError("th_outlier must hold 1..%d thresholds" % _lib.MAX_TH)
        self.generator = generator
        self.obj_param = np.asarray(obj_param, np.float64).reshape(6)
        self.th_outlier = [float(t) for t in th_outlier]
        self.th_inlier = float(th_inlier)
        self.box_size = float(box_size)

    def as_struct(self) -> "_lib.Object":
        o = _lib.Object()
        o.model = self.generator.handle
        for k in range(3):
            o.obj_scale[k] = self.obj_param[k]
            o.obj_ct[k] = self.obj_param[3 + k]
        o.n_outlier_th = len(self.th_outlier)
        for k, t in enumerate(self.th_outlier):
            o.outlier_th[k] = t
        o.inlier_th = self.th_inlier
        o.box_size = self.box_size
        return o


def _image_struct(img):
    """numpy HxWx3 uint8/float32 array, or (device_ptr, H, W, 'u8'|'f32') for a frame already in HBM."""
    s = _lib.Image()
    if isinstance(img, tuple):
        ptr, h, w, dt = img
        s.data, s.height, s.width = ptr, h, w
        s.dtype = 1 if dt == "f32" else 0
        s.mem = _lib.MEM_DEVICE
        return s, None
    a = np.asarray(img)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("image must be HxWx3, got %r" % (a.shape,))
    if a.dtype == np.uint8:
        a = np.ascontiguousarray(a)
        s.dtype = 0
    else:
        a = np.ascontiguousarray(a, dtype=np.float32)
        s.dtype = 1
    s.data, s.height, s.width, s.mem = a.ctypes.data, a.shape[0], a.shape[1], _lib.MEM_HOST
    return s, a


# skimage.transform.resize GENERATION (p2p_est_pose_opts.resize_anti_aliasing; include/p2p_mi355.h has the full account).  The reference calls
# resize six times per detection and does not pin scikit-image, so the caller names the library its reference environment resolves to.
RESIZE_GENERATIONS = {"0.14": 0, "0.15": 2, "0.16": 2, "0.17": 1, "0.18": 1}
_warned_unpinned = False


def resize_generation(spec) -> int:
    """None / False / 0 / "0.14" -> 0 (scikit-image <= 0.14: no anti-aliasing, every image warped in double; pinned to the real 0.18.3 float64 warp);
    True / 1 / "0.17" / "0.18" -> 1 (pinned bit for bit to the real 0.18.3); 2 / "0.15" / "0.16" -> 2 (anti-aliasing on for every input
    including the bool keep mask; its Gaussian filter pinned to the real scipy, its warp restated)."""
    if spec is None or spec is False:
        return 0
    if spec is True:
        return 1
    if isinstance(spec, str):
        if spec not in RESIZE_GENERATIONS:
            raise ValueError("skimage must be one of %s (>= 0.19 rejects the reference's bool mask resize), got %r" % (sorted(RESIZE_GENERATIONS), spec))
        return RESIZE_GENERATIONS[spec]
    g = int(spec)
    if g not in (0, 1, 2):
        raise ValueError("resize generation must be 0, 1 or 2, got %r" % (spec,))
    return g


def warn_unpinned_generation(who: str):
    """The reference does not pin scikit-image and its six resize calls differ between the library's generations: a caller that names none gets
    the <= 0.14 semantics (pinned to the real library's float64 warp like the others) -- say so, once."""
    global _warned_unpinned
    if not _warned_unpinned:
        _warned_unpinned = True
        import warnings
        warnings.warn("%s: no scikit-image generation was named -- using the <= 0.14 resize semantics (no anti-aliasing filter).  The reference "
                      "does not pin scikit-image and the generations give different masks: name the version the reference environment "
                      "resolves to (skimage='0.14', '0.15' / '0.16' -- what the reference's python-3.5 image installs -- or '0.17' / '0.18') "
                      "to silence this" % who, stacklevel=3)


def _marshal(objects, images, detections, inject1, inject2, inject_slots, want_masks, det_masks, ransac_iterations,
             reprojection_error, confidence, anti_aliasing=False):
    """Build the C arrays of one batch call.  -> (objs, imgs, dets, opts, extras, keep-alive list)"""
    n = len(detections)
    if ransac_iterations > _lib.MAX_RANSAC_ITERATIONS:
        raise ValueError("ransac_iterations %d exceeds the library's limit of %d" % (ransac_iterations, _lib.MAX_RANSAC_ITERATIONS))
    objs = (_lib.Object * max(len(objects), 1))(*[o.as_struct() for o in objects])
    keep = [objects, images]
    imgs = (_lib.Image * max(len(images), 1))()
    for i, im in enumerate(images):
        imgs[i], a = _image_struct(im)
        keep.append(a)
    dets = (_lib.Detection * max(n, 1))()
    if n:      # one vectorised fill instead of 15 ctypes assignments per detection
        dv = np.frombuffer(dets, dtype=_lib.DETECTION_DTYPE, count=n)
        dv["image"] = [d[0] for d in detections]
        dv["object"] = [d[1] for d in detections]
        dv["bbox"] = np.array([[int(b) for b in d[2]] for d in detections], np.int32).reshape(n, 4)     # int() truncation like the reference's roi.astype(np.int)
        dv["camK"] = np.array([np.asarray(d[3], np.float64).reshape(9) for d in detections], np.float64)
    opts = _lib.EstPoseOpts()
    opts.ransac_iterations, opts.reprojection_error, opts.confidence = ransac_iterations, reprojection_error, confidence
    opts.inject1, opts.inject2, opts.inject_slots = inject1, inject2, inject_slots
    opts.resize_anti_aliasing = resize_generation(anti_aliasing)
    extras = {}
    if want_masks and n:
        def hw(i):
            im = images[detections[i][0]]
            return (im[1], im[2]) if isinstance(im, tuple) else np.asarray(im).shape[:2]
        mstride = max(hw(i)[0] * hw(i)[1] for i in range(n))

        def side(i):      # stage-1 square of the detection (get_boxes, recognition.py:28-43): the stage-2 crop never exceeds it
            _, oi, b, _ = detections[i]
            bs = objects[oi].box_size if 0 <= oi < len(objects) else 1.5
            return 2 * int(min(9999, max((b[3] - b[1]) * bs, (b[2] - b[0]) * bs)) / 2)
        pstride = 3 * min(mstride, max(max(side(i), 1) ** 2 for i in range(n)))
        extras["valid_mask"] = np.zeros((n, mstride), np.uint8)
        extras["img_pred"] = np.zeros((n, pstride), np.uint8)
        opts.valid_mask, opts.mask_stride = extras["valid_mask"].ctypes.data, mstride
        opts.mask_prezeroed = 1          # np.zeros above: fresh zero pages, the library writes the crop rows only
        opts.img_pred, opts.pred_stride = extras["img_pred"].ctypes.data, pstride
    if det_masks is not None and n:
        # score_type 2: IoU of each detector mask with valid_mask_full, computed on the device
        dms = max(int(np.asarray(m).size) for m in det_masks)
        dm = np.zeros((n, dms), np.uint8)
        for i, m in enumerate(det_masks):
            im = images[detections[i][0]]
            hw = (im[1], im[2]) if isinstance(im, tuple) else tuple(np.asarray(im).shape[:2])
            if tuple(np.asarray(m).shape[:2]) != tuple(hw):
                # the device reads the mask with the frame's row stride; the reference resizes such masks to the frame first
                # (tools/5_evaluation_bop_basic.py:309-310) -- that resize is left to the caller; a mismatch is an error here, not a silent misread
                raise ValueError("detector mask %d has shape %r, its frame %r" % (i, tuple(np.asarray(m).shape[:2]), tuple(hw)))
            mm = np.ascontiguousarray(np.asarray(m) != 0, dtype=np.uint8).reshape(-1)
            dm[i, :mm.size] = mm
        extras["mask_stats"] = np.zeros((n, 3), np.int64)
        extras["_dm"] = dm
        opts.det_mask, opts.det_mask_stride = dm.ctypes.data, dms
        opts.mask_stats = extras["mask_stats"].ctypes.data
    return objs, imgs, dets, opts, extras, keep


def est_pose_batch(ctx: Context, objects, images, detections, *, inject1=None, inject2=None, inject_slots=0,
                   want_masks=False, debug=False, ransac_iterations=0, reprojection_error=0.0, confidence=0.0,
                   det_masks=None, anti_aliasing=False):
    """detections: list of (image_idx, object_idx, bbox[v1,u1,v2,u2], camK 3x3).
    anti_aliasing: scikit-image 0.17-0.18 resize semantics (Gaussian pre-filter when down-scaling); default = <= 0.14.
    Returns (poses: list[_lib.Pose], extras: dict)."""
    n = len(detections)
    objs, imgs, dets, opts, extras, keep = _marshal(objects, images, detections, inject1, inject2, inject_slots, want_masks,
                                                    det_masks, ransac_iterations, reprojection_error, confidence, anti_aliasing)
    poses = (_lib.Pose * max(n, 1))()
    K = max([len(o.th_outlier) for o in objects], default=0)
    if debug and n:
        extras["x1"] = np.zeros((n, 128, 128, 3), np.float32)
        extras["x2"] = np.zeros((n, K, 128, 128, 3), np.float32)
        extras["boxes2"] = np.zeros((n, 12), np.int32)
        extras["cand"] = np.zeros((n, K, 6), np.int32)
        opts.dbg_x1, opts.dbg_x2 = extras["x1"].ctypes.data, extras["x2"].ctypes.data
        opts.dbg_boxes2, opts.dbg_cand = extras["boxes2"].ctypes.data, extras["cand"].ctypes.data
        extras["y1"] = np.zeros((n, 128, 128, 4), np.float32)
        extras["y2"] = np.zeros((n, K, 128, 128, 4), np.float32)
        opts.dbg_y1, opts.dbg_y2 = extras["y1"].ctypes.data, extras["y2"].ctypes.data
    _lib.check(_lib.lib().p2p_est_pose_batch(ctx.handle, objs, len(objects), imgs, len(images), dets, n, poses,
                                             C.byref(opts)), "p2p_est_pose_batch")
    return [poses[i] for i in range(n)], extras


class Comm:
    """RCCL communicator of the C ABI (p2p_comm_*): one per rank, on the rank's context.  ``Comm.unique_id()`` on rank 0, the 128 bytes
    handed to every rank by the host program, ``Comm(ctx, rank, world, id)`` on all of them (collective)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().p2p_comm_unique_id(buf), "p2p_comm_unique_id")
        return buf.raw

    def __init__(self, ctx: Context, rank: int, world: int, uid: bytes):
        if len(uid) != _lib.COMM_ID_BYTES:
            raise ValueError("the communicator id has %d bytes" % _lib.COMM_ID_BYTES)
        self.ctx, self.rank, self.world = ctx, rank, world
        self._h = C.c_void_p()
        _lib.check(_lib.lib().p2p_comm_create(ctx.handle, rank, world, uid, C.byref(self._h)), "p2p_comm_create")

    @property
    def handle(self):
        return self._h

    @staticmethod
    def library() -> str:
        return (_lib.lib().p2p_comm_library() or b"").decode()

    def close(self):
        if self._h:
            _lib.lib().p2p_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PendingBatch:
    """Handle of a batch enqueued with est_pose_submit (keeps the argument and output buffers alive)."""

    def __init__(self, ctx, ticket, n, keep, extras):
        self.ctx, self.ticket, self.n, self._keep, self.extras = ctx, ticket, n, keep, extras

    def collect(self):
        """Wait for the batch; -> list of p2p_pose records in the caller's detection order.  The optional outputs
        requested at submit time (valid_mask / img_pred / mask_stats) are in ``self.extras`` afterwards."""
        poses = (_lib.Pose * max(self.n, 1))()
        _lib.check(_lib.lib().p2p_est_pose_collect(self.ctx.handle, self.ticket, poses), "p2p_est_pose_collect")
        self._keep = None
        self.pose_array = poses          # the ctypes array itself (parallel.poses_to_records takes it without a Python loop)
        return [poses[i] for i in range(self.n)]

    def collect_gathered(self, comm: "Comm", n_max: int):
        """collect() + the RCCL all-gather of every rank's records (p2p_est_pose_collect_gathered: device to device on the batch's tail
        stream, one D2H afterwards).  Collective.  -> (this rank's poses, ctypes array of world * n_max records in rank order, each
        rank's block in its caller's detection order, padded with status = POSE_ABSENT)."""
        poses = (_lib.Pose * max(self.n, 1))()
        allp = (_lib.Pose * (comm.world * n_max))()
        _lib.check(_lib.lib().p2p_est_pose_collect_gathered(self.ctx.handle, comm.handle, self.ticket, poses, n_max, allp),
                   "p2p_est_pose_collect_gathered")
        self._keep = None
        self.pose_array = poses
        return [poses[i] for i in range(self.n)], allp


def collect_gathered_empty(ctx: Context, comm: "Comm", n_max: int):
    """This rank has NO batch this step (an empty shard: its images ran out, or none of its detections survived the candidate limits)
    but its peers do: join their p2p_est_pose_collect_gathered with ticket = P2P_TICKET_NONE, contributing n_max padding records.
    Collective.  -> ctypes array of world * n_max records as from PendingBatch.collect_gathered."""
    allp = (_lib.Pose * (comm.world * n_max))()
    _lib.check(_lib.lib().p2p_est_pose_collect_gathered(ctx.handle, comm.handle, _lib.TICKET_NONE, None, n_max, allp),
               "p2p_est_pose_collect_gathered(empty shard)")
    return allp


def est_pose_submit(ctx: Context, objects, images, detections, *, inject1=None, inject2=None, inject_slots=0,
                    ransac_iterations=0, reprojection_error=0.0, confidence=0.0, want_masks=False, det_masks=None,
                    anti_aliasing=False, merge_passes=False) -> PendingBatch:
    """Asynchronous est_pose_batch for detection streams: enqueue and return; at most two batches in
    flight per context.  The PnP-RANSAC tail of this batch overlaps the generator passes of the next.
    ``want_masks`` / ``det_masks`` as in est_pose_batch: the arrays in ``PendingBatch.extras`` are filled by collect()."""
    n = len(detections)
    objs, imgs, dets, opts, extras, keep = _marshal(objects, images, detections, inject1, inject2, inject_slots, want_masks,
                                                    det_masks, ransac_iterations, reprojection_error, confidence, anti_aliasing)
    opts.merge_stream_passes = 1 if merge_passes else 0       # the next submit may run this batch's stage-2 pass merged with its stage-1 pass
    ticket = C.c_int(-1)
    _lib.check(_lib.lib().p2p_est_pose_submit(ctx.handle, objs, len(objects), imgs, len(images), dets, n, C.byref(opts),
                                              C.byref(ticket)), "p2p_est_pose_submit")
    keep += [objs, imgs, dets, opts]
    return PendingBatch(ctx, ticket.value, n, keep, extras)


def pnp_ransac_batch(ctx: Context, Ks, objs, imgs, iterations=100, reproj_err=5.0, confidence=0.99, want_mask=False):
    """Batch of independent cv2.solvePnPRansac(EPNP) problems on the GPU.
    -> ok [P] bool, R [P,3,3], t [P,3], info [P,3] (n_inliers, iterations, best_iter), masks list|None"""
    n_prob = len(objs)
    offsets = np.zeros(n_prob + 1, np.int32)
    offsets[1:] = np.cumsum([len(o) for o in objs])
    obj = np.ascontiguousarray(np.concatenate(objs) if n_prob else np.zeros((0, 3)), np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(np.concatenate(imgs) if n_prob else np.zeros((0, 2)), np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(Ks, np.float64).reshape(n_prob, 9)
    R = np.zeros((n_prob, 9)); t = np.zeros((n_prob, 3))
    info = np.zeros((n_prob, 3), np.int32); ok = np.zeros(n_prob, np.int32)
    mask = np.zeros(max(int(offsets[-1]), 1), np.uint8) if want_mask else None
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    _lib.check(_lib.lib().p2p_pnp_ransac_batch(ctx.handle, K.ctypes.data_as(dp), obj.ctypes.data_as(dp),
                                               img.ctypes.data_as(dp), offsets.ctypes.data_as(ip), n_prob, iterations,
                                               reproj_err, confidence, R.ctypes.data_as(dp), t.ctypes.data_as(dp),
                                               info.ctypes.data_as(ip), ok.ctypes.data_as(ip),
                                               mask.ctypes.data if want_mask else None), "p2p_pnp_ransac_batch")
    masks = [mask[offsets[i]:offsets[i + 1]] for i in range(n_prob)] if want_mask else None
    return ok.astype(bool), R.reshape(-1, 3, 3), t, info, masks
