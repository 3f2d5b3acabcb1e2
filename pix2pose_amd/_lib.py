"""ctypes binding of libp2p_mi355.so (C ABI: include/p2p_mi355.h).

There is no CPU fallback: if the HIP library cannot be loaded, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libp2p_mi355.so")

P2P_OK = 0
BACKBONE = {"paper": 0, "resnet50": 1}
MEM_HOST, MEM_DEVICE = 0, 1


class P2PError(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


_lib = None


def lib():
    """Load (building first if the .so is absent and hipcc is present) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        try:
            _build.build()
        except Exception as e:  # pragma: no cover
            raise P2PError("libp2p_mi355.so is not built and could not be built: %s" % (e,))
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise P2PError("cannot load %s: %s (the HIP path is mandatory, there is no fallback)" % (LIB_PATH, e))
    vp, fp, ci = C.c_void_p, C.POINTER(C.c_float), C.c_int
    L.p2p_abi_version.restype = ci
    L.p2p_last_error.restype = C.c_char_p
    L.p2p_device_count.argtypes = [C.POINTER(ci)]
    L.p2p_ctx_create.argtypes = [ci, ci, C.POINTER(vp)]
    L.p2p_ctx_destroy.argtypes = [vp]
    L.p2p_ctx_destroy.restype = None
    L.p2p_ctx_synchronize.argtypes = [vp]
    L.p2p_ctx_stream.argtypes = [vp]
    L.p2p_ctx_stream.restype = vp
    L.p2p_model_create.argtypes = [vp, C.POINTER(Tensor), ci, ci, C.POINTER(vp)]
    L.p2p_model_destroy.argtypes = [vp]
    L.p2p_model_destroy.restype = None
    L.p2p_predict.argtypes = [vp, vp, vp, ci, vp, vp, ci]
    L.p2p_forward_async.argtypes = [vp, vp, vp, ci, vp]
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != P2P_OK:
        msg = lib().p2p_last_error()
        raise P2PError("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))
