"""ctypes binding of libp2p_mi355.so (C ABI: include/p2p_mi355.h).

There is no CPU fallback: if the HIP library cannot be loaded, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P2P_LIB", os.path.join(_HERE, "libp2p_mi355.so"))     # P2P_LIB: development override

P2P_OK = 0
ABI_VERSION = 9            # P2P_ABI_VERSION of include/p2p_mi355.h these ctypes declarations follow
MAX_RANSAC_ITERATIONS = 128
BACKBONE = {"paper": 0, "resnet50": 1}
PRECISION = {"f32": 0, "f16x3": 1, "auto": 2}     # p2p_precision; "auto" = split-f16 with an fp32 twin it falls back to on a range event
ERR_RANGE = -5
MEM_HOST, MEM_DEVICE = 0, 1
COMM_ID_BYTES = 128
POSE_ABSENT = -1
POSE_RANGE = -2          # gathered record of a rank whose batch left the split-f16 operand range
TICKET_NONE = -1         # p2p_est_pose_collect_gathered: this rank has no batch this step (empty shard)


class P2PError(RuntimeError):
    pass


class P2PRangeError(P2PError):
    """P2P_ERR_RANGE: a split-f16 generator pass stored an activation beyond the f16 operand range (include/p2p_mi355.h)."""


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


MAX_TH = 8


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("height", C.c_int), ("width", C.c_int), ("dtype", C.c_int), ("mem", C.c_int)]


class Object(C.Structure):
    _fields_ = [("model", C.c_void_p), ("obj_scale", C.c_double * 3), ("obj_ct", C.c_double * 3),
                ("n_outlier_th", C.c_int), ("outlier_th", C.c_double * MAX_TH), ("inlier_th", C.c_double),
                ("box_size", C.c_double)]


class Detection(C.Structure):
    _fields_ = [("image", C.c_int), ("object", C.c_int), ("bbox", C.c_int * 4), ("camK", C.c_double * 9)]


class Pose(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3), ("frac_inlier", C.c_double), ("n_inliers", C.c_int),
                ("n_init_mask", C.c_int), ("status", C.c_int), ("best_slot", C.c_int), ("bbox_t", C.c_int * 4),
                ("n_candidates", C.c_int), ("ransac_iters", C.c_int), ("mask_stats", C.c_int64 * 3)]


import numpy as _np

# numpy views of the struct arrays (same field order / alignment as the ctypes definitions; sizes asserted below)
POSE_DTYPE = _np.dtype([("R", "<f8", (9,)), ("t", "<f8", (3,)), ("frac_inlier", "<f8"), ("n_inliers", "<i4"), ("n_init_mask", "<i4"),
                        ("status", "<i4"), ("best_slot", "<i4"), ("bbox_t", "<i4", (4,)), ("n_candidates", "<i4"), ("ransac_iters", "<i4"),
                        ("mask_stats", "<i8", (3,))], align=True)
DETECTION_DTYPE = _np.dtype([("image", "<i4"), ("object", "<i4"), ("bbox", "<i4", (4,)), ("camK", "<f8", (9,))], align=True)
assert POSE_DTYPE.itemsize == C.sizeof(Pose) and DETECTION_DTYPE.itemsize == C.sizeof(Detection)


class EstPoseOpts(C.Structure):
    _fields_ = [("ransac_iterations", C.c_int), ("reprojection_error", C.c_double), ("confidence", C.c_double),
                ("inject1", C.c_void_p), ("inject2", C.c_void_p), ("inject_slots", C.c_int),
                ("valid_mask", C.c_void_p), ("mask_stride", C.c_int64), ("img_pred", C.c_void_p),
                ("pred_stride", C.c_int64), ("dbg_x1", C.c_void_p), ("dbg_x2", C.c_void_p),
                ("dbg_boxes2", C.c_void_p), ("dbg_cand", C.c_void_p), ("dbg_y1", C.c_void_p), ("dbg_y2", C.c_void_p),
                ("det_mask", C.c_void_p), ("det_mask_stride", C.c_int64), ("mask_stats", C.c_void_p),
                ("resize_anti_aliasing", C.c_int), ("merge_stream_passes", C.c_int), ("mask_prezeroed", C.c_int)]


PROFILE_SLOTS = 22    # P2P_PROFILE_SLOTS
# kernel family of each slot: (label, substring of the rocprofv3 kernel name; %d = precision template argument)
PROFILE_KERNELS = [("igemm_kernel 128x128 tiles", "igemm_kernel<2, 2, 2, 2, %d>"),
                   ("igemm_kernel 128x64 tiles", "igemm_kernel<2, 2, 2, 1, %d>"),
                   ("igemm_kernel 128x32 tiles", "igemm_kernel<4, 1, 1, 1, %d>"),
                   ("igemm_halo_kernel 128x128 tiles (halo-tiled stride-1 multi-tap layers)", "igemm_halo_kernel<2, 2>"),
                   ("igemm_halo_kernel 256x64 tiles (Cout = 64 layers)", "igemm_halo_kernel<4, 2>"),
                   ("heads_halo_kernel (merged output heads)", "heads_halo_kernel"),
                   ("igemm_halo8_kernel 128x128 tiles (8x8-grid layers: conv4 through parity planes, first transposed conv)", "igemm_halo8_kernel"),
                   ("igemm_halo_s2_kernel (5x5 stride-2 convolutions on 16x16 and larger grids: the paper encoder)", "igemm_halo_s2_kernel"),
                   ("igemm_stream_kernel (small launches: one wave per 32x32 output tile, operands streamed to registers)", "igemm_stream_kernel"),
                   ("resblock_kernel (ResNet identity bottleneck block in one launch: 1x1 -> 3x3 -> 1x1 + residual, intermediates in LDS)", "resblock_kernel"),
                   ("wino_gemm_kernel (5x5 stride-1 decoder layers, Winograd F(4,5) along the row axis: eight position GEMMs + inverse transform)", "wino_gemm_kernel"),
                   ("wino_input_kernel (input transform of the Winograd layers: x -> split-f16 V)", "wino_input_kernel"),
                   ("pnp_hypotheses_kernel (5-point EPnP models, fp64)", "pnp_hypotheses_kernel"),
                   ("pnp_count_kernel (inlier counts of 8 models per pass over the correspondences)", "pnp_count_kernel"),
                   ("pnp_score_kernel (OpenCV's rule on the counts + Gram sums of the best model's inliers)", "pnp_score_kernel"),
                   ("pnp_fit_solve_kernel + pnp_fit_select_kernel (refit on the inliers, final inlier mask)", "pnp_fit_"),
                   ("aa_filter_kernel<0> (Gaussian pre-filter, first axis)", "aa_filter_kernel<0>"),
                   ("aa_filter_kernel<1> (Gaussian pre-filter, second axis)", "aa_filter_kernel<1>"),
                   ("cand_eval_kernel + cand_compact_kernel / cand_corr_kernel (stage-2 decode: back-resizes, masks, correspondences)", "cand_"),
                   ("stage2_input_kernel (stage-2 crops: masked re-crop + resize to 128x128)", "stage2_input_kernel"),
                   ("wino3_gemm_kernel (transposed convolutions up2 / up3, Winograd F(4,3) along the row axis: six position GEMMs for the four phases + inverse transform)", "wino3_gemm_kernel"),
                   ("wino3_input_kernel (input transform of the Winograd transposed convolutions: x -> split-f16 V)", "wino3_input_kernel")]


class KernelStats(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("algo_flops", C.c_double), ("algo_bytes", C.c_double)]


_lib = None


_torch_hip_loaded = False


def _preload_torch_hip():
    """One process, one HIP runtime.  A PyTorch-ROCm wheel carries its own libamdhip64.so; the library links the system one under
    the same soname.  Whichever is mapped first serves both, and torch cannot see a device through the system runtime when the
    library came first ("No HIP GPUs are available").  So, when torch is installed, its runtime is mapped before the library is
    -- the order in which the caller imports things no longer matters.  (Found without importing torch: that costs seconds.)"""
    global _torch_hip_loaded
    if _torch_hip_loaded:
        return
    _torch_hip_loaded = True
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        fn = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(fn):
            C.CDLL(fn, mode=C.RTLD_GLOBAL)
    except Exception:  # pragma: no cover -- best effort; torch-first import orders never needed it
        pass


def _stale_reason(path):
    """Why the library at `path` must not be used with this tree (None = fine).  Checked through a private handle that is
    closed again: a refused library is never left typed in `_lib`, and after a rebuild the loader maps the NEW file instead of
    handing back the image it still had open under the same path."""
    _preload_torch_hip()
    try:
        L = C.CDLL(path)
    except OSError as e:
        return "cannot load %s: %s" % (path, e)
    try:
        return _stale_reason_of(L)
    finally:
        try:
            import _ctypes
            _ctypes.dlclose(L._handle)
        except Exception:  # pragma: no cover
            pass


def _stale_reason_of(L):
    L.p2p_abi_version.restype = C.c_int
    if L.p2p_abi_version() != ABI_VERSION:
        return "ABI version %d, binding expects %d" % (L.p2p_abi_version(), ABI_VERSION)
    if not hasattr(L, "p2p_build_id"):
        return "no build id"
    L.p2p_build_id.restype = C.c_char_p
    from . import build as _build
    # (an explicitly named A/B build is taken as is; a deployment that ships the .so without its sources has nothing to compare
    # with -- the ABI version and struct sizes below still guard the binding)
    if "P2P_LIB" not in os.environ and _build.have_sources():
        want = _build.source_hash()
        got = (L.p2p_build_id() or b"").decode()
        if got != want:
            return "built from other sources (build id %s, tree %s)" % (got, want)
    L.p2p_abi_sizeof.restype = C.c_int
    L.p2p_abi_sizeof.argtypes = [C.c_int]
    for which, typ in enumerate((Tensor, Image, Object, Detection, Pose, EstPoseOpts, KernelStats)):
        if L.p2p_abi_sizeof(which) != C.sizeof(typ):
            return "sizeof(%s) = %d in the library, %d in the binding" % (typ.__name__, L.p2p_abi_sizeof(which), C.sizeof(typ))
    return None


def lib():
    """Load and type the library.  A missing or stale .so (other ABI version, other sources, other struct sizes) is
    rebuilt when hipcc is present and refused otherwise -- never loaded silently."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build
    reason = "not built" if not os.path.exists(LIB_PATH) else _stale_reason(LIB_PATH)
    if reason is not None:
        if "P2P_LIB" in os.environ or not _build.have_compiler():
            raise P2PError("%s: %s (the HIP path is mandatory, there is no fallback)" % (LIB_PATH, reason))
        try:
            _build.build()
        except Exception as e:  # pragma: no cover
            raise P2PError("libp2p_mi355.so is %s and could not be rebuilt: %s" % (reason, e))
        reason = _stale_reason(LIB_PATH)
        if reason is not None:
            raise P2PError("%s: %s after a rebuild" % (LIB_PATH, reason))
    _preload_torch_hip()
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise P2PError("cannot load %s: %s (the HIP path is mandatory, there is no fallback)" % (LIB_PATH, e))
    vp, fp, ci = C.c_void_p, C.POINTER(C.c_float), C.c_int
    L.p2p_abi_version.restype = ci
    L.p2p_abi_sizeof.argtypes = [ci]
    L.p2p_build_id.restype = C.c_char_p
    L.p2p_aa_weights.argtypes = [ci, C.POINTER(C.c_double)]
    L.p2p_last_error.restype = C.c_char_p
    L.p2p_device_count.argtypes = [C.POINTER(ci)]
    L.p2p_ctx_create.argtypes = [ci, ci, C.POINTER(vp)]
    L.p2p_ctx_destroy.argtypes = [vp]
    L.p2p_ctx_destroy.restype = None
    L.p2p_ctx_synchronize.argtypes = [vp]
    L.p2p_ctx_stream.argtypes = [vp]
    L.p2p_ctx_stream.restype = vp
    L.p2p_model_create.argtypes = [vp, C.POINTER(Tensor), ci, ci, C.POINTER(vp)]
    L.p2p_model_create_ex.argtypes = [vp, C.POINTER(Tensor), ci, ci, ci, C.POINTER(vp)]
    L.p2p_model_destroy.argtypes = [vp]
    L.p2p_model_destroy.restype = None
    L.p2p_model_precision.argtypes = [vp]
    L.p2p_ctx_range_event.argtypes = [vp, C.POINTER(C.c_float)]
    L.p2p_ctx_set_winograd.argtypes = [vp, ci]
    L.p2p_predict.argtypes = [vp, vp, vp, ci, vp, vp, ci]
    L.p2p_forward_async.argtypes = [vp, vp, vp, ci, vp]
    L.p2p_est_pose_batch.argtypes = [vp, C.POINTER(Object), ci, C.POINTER(Image), ci, C.POINTER(Detection), ci,
                                     C.POINTER(Pose), C.POINTER(EstPoseOpts)]
    L.p2p_est_pose_submit.argtypes = [vp, C.POINTER(Object), ci, C.POINTER(Image), ci, C.POINTER(Detection), ci,
                                      C.POINTER(EstPoseOpts), C.POINTER(ci)]
    L.p2p_est_pose_collect.argtypes = [vp, ci, C.POINTER(Pose)]
    L.p2p_debug_back_resize.argtypes = [vp, vp, vp, vp, ci, ci, C.c_double, ci, vp, vp, vp]
    L.p2p_comm_unique_id.argtypes = [C.c_char_p]
    L.p2p_comm_create.argtypes = [vp, ci, ci, C.c_char_p, C.POINTER(vp)]
    L.p2p_comm_destroy.argtypes = [vp]
    L.p2p_comm_destroy.restype = None
    L.p2p_comm_library.restype = C.c_char_p
    L.p2p_est_pose_collect_gathered.argtypes = [vp, vp, ci, C.POINTER(Pose), ci, C.POINTER(Pose)]
    L.p2p_profile_enable.argtypes = [vp, ci]
    L.p2p_profile_read.argtypes = [vp, C.POINTER(KernelStats), ci]
    dp = C.POINTER(C.c_double)
    L.p2p_pnp_ransac_batch.argtypes = [vp, dp, dp, dp, C.POINTER(ci), ci, ci, C.c_double, C.c_double, dp, dp,
                                       C.POINTER(ci), C.POINTER(ci), vp]
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != P2P_OK:
        msg = lib().p2p_last_error()
        raise (P2PRangeError if rc == ERR_RANGE else P2PError)("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))
