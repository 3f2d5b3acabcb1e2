"""Python handles over the C ABI: a per-GPU context and per-object generator networks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import weights as W


class Context:
    """One GPU's pipeline context (stream + workspaces).  Not thread-safe (p2p_mi355.h)."""

    def __init__(self, device: int = 0, max_batch: int = 256):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().p2p_ctx_create(device, max_batch, C.byref(self._h)), "p2p_ctx_create")
        self.device = device
        self.max_batch = max_batch

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return _lib.lib().p2p_ctx_stream(self._h) or 0

    def synchronize(self):
        _lib.check(_lib.lib().p2p_ctx_synchronize(self._h), "p2p_ctx_synchronize")

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

    def __init__(self, weights: dict, backbone: str, ctx: Context | None = None):
        if backbone not in _lib.BACKBONE:
            raise ValueError("unknown backbone %r" % (backbone,))
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
        _lib.check(_lib.lib().p2p_model_create(self.ctx.handle, arr, len(specs), _lib.BACKBONE[backbone],
                                               C.byref(self._h)), "p2p_model_create")

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
