"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement of the Pix2Pose generator forward pass (what
``self.generator_train.predict(x)`` computes, reference pix2pose_model/recognition.py:84,129):

  * ``forward(w, x, 'resnet50')`` follows pix2pose_model/ae_model.py:175-240 with the front
    of pix2pose_model/resnet50_mod.py:200-213 (blocks :40-73 identity, :76-118 conv).
  * ``forward(w, x, 'paper')``    follows pix2pose_model/ae_model.py:70-150.

The graph wiring is here; the arithmetic is in oracle/ae_layers.c (double accumulation).

PINNING: the graph WIRING is pinned to the reference -- tests/golden/reference_graph.json holds outputs of the
graphs built by the reference's own ae_model.py / resnet50_mod.py (executed against a stand-in Keras API,
tests/golden/make_reference_graph_vectors.py) and forward() reproduces them to 1e-6.  The layer SEMANTICS
(oracle/ae_layers.c: TF "SAME" padding, Conv2DTranspose, BatchNormalization, LeakyReLU defaults) stay UNPINNED:
Keras/TensorFlow are not installable here and the reference ships no tests; they are
checked against an independent torch-CPU formulation and micro-fixtures of TF "SAME" conv/deconv semantics.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build as _build

BN_EPS = 1e-3
LEAKY = 0.3
ACT = {"none": 0, "relu": 1, "leaky": 2, "tanh": 3, "sigmoid": 4}

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        fp = C.POINTER(C.c_float)
        _lib.p2po_conv2d.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        _lib.p2po_conv2d_transpose.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int,
                                               C.c_int, C.c_int, C.c_int, fp]
        _lib.p2po_bn_act.argtypes = [fp, C.c_size_t, C.c_int, fp, fp, fp, fp, C.c_double, C.c_int, C.c_double]
        _lib.p2po_add_relu.argtypes = [fp, fp, C.c_size_t, fp]
        _lib.p2po_maxpool.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, fp]
        _lib.p2po_dense.argtypes = [fp, C.c_int, C.c_int, fp, fp, C.c_int, fp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def same_pad(n, k, s):
    """TF 'SAME': (out, pad_before)."""
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2


def conv2d(x, kernel, bias, stride=1, padding="same"):
    x = _f32(x); kernel = _f32(kernel)
    n, h, w_, cin = x.shape
    kh, kw, kcin, cout = kernel.shape
    assert kcin == cin, (kernel.shape, x.shape)
    if padding == "same":
        ho, pt = same_pad(h, kh, stride)
        wo, pl = same_pad(w_, kw, stride)
    else:  # valid
        ho, pt = (h - kh) // stride + 1, 0
        wo, pl = (w_ - kw) // stride + 1, 0
    y = np.empty((n, ho, wo, cout), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().p2po_conv2d(_p(x), n, h, w_, cin, _p(kernel), _p(b), kh, kw, cout, stride, pt, pl, ho, wo, _p(y))
    return y


def conv2d_transpose(x, kernel, bias, stride=2):
    x = _f32(x); kernel = _f32(kernel)
    n, h, w_, cin = x.shape
    kh, kw, cout, kcin = kernel.shape
    assert kcin == cin, (kernel.shape, x.shape)
    y = np.empty((n, h * stride, w_ * stride, cout), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().p2po_conv2d_transpose(_p(x), n, h, w_, cin, _p(kernel), _p(b), kh, kw, cout, stride, _p(y))
    return y


def bn_act(x, w, name, act):
    """In-place BatchNorm (if ``name`` has BN tensors) + activation."""
    c = x.shape[-1]
    g = w.get(name + ".gamma")
    if g is not None:
        lib().p2po_bn_act(_p(x), x.size // c, c, _p(_f32(g)), _p(_f32(w[name + ".beta"])),
                          _p(_f32(w[name + ".mean"])), _p(_f32(w[name + ".var"])), BN_EPS, ACT[act], LEAKY)
    else:
        lib().p2po_bn_act(_p(x), x.size // c, c, None, None, None, None, BN_EPS, ACT[act], LEAKY)
    return x


def add_relu(a, b):
    y = np.empty_like(a)
    lib().p2po_add_relu(_p(a), _p(b), a.size, _p(y))
    return y


def maxpool_3x3_s2_same(x):
    n, h, w_, c = x.shape
    ho, pt = same_pad(h, 3, 2)
    wo, pl = same_pad(w_, 3, 2)
    y = np.empty((n, ho, wo, c), np.float32)
    lib().p2po_maxpool(_p(x), n, h, w_, c, 3, 2, pt, pl, ho, wo, _p(y))
    return y


def dense(x, kernel, bias):
    x = _f32(x); kernel = _f32(kernel)
    n, cin = x.shape
    y = np.empty((n, kernel.shape[1]), np.float32)
    lib().p2po_dense(_p(x), n, cin, _p(kernel), _p(_f32(bias)), kernel.shape[1], _p(y))
    return y


def _cba(x, w, name, stride, padding, act):
    """Conv2D -> BatchNormalization -> activation."""
    y = conv2d(x, w[name + ".kernel"], w[name + ".bias"], stride, padding)
    return bn_act(y, w, name, act)


def _identity_block(x, w, base):
    """resnet50_mod.py:40-73."""
    y = _cba(x, w, base + "_2a", 1, "valid", "relu")       # :59-61
    y = _cba(y, w, base + "_2b", 1, "same", "relu")        # :63-66
    y = _cba(y, w, base + "_2c", 1, "valid", "none")       # :68-69
    return add_relu(y, x)                                  # :71-72


def _conv_block(x, w, base, stride):
    """resnet50_mod.py:76-118 (stride on the first 1x1 and on the shortcut)."""
    y = _cba(x, w, base + "_2a", stride, "valid", "relu")  # :99-102
    y = _cba(y, w, base + "_2b", 1, "same", "relu")        # :104-107
    y = _cba(y, w, base + "_2c", 1, "valid", "none")       # :109-110
    sc = _cba(x, w, base + "_1", stride, "valid", "none")  # :112-114
    return add_relu(y, sc)                                 # :116-117


def _resnet_front(w, x):
    """resnet50_mod.py:200-213 up to act3d_branch; returns (f1, f2, f3) (ae_model.py:179-184)."""
    xp = np.pad(x, ((0, 0), (3, 3), (3, 3), (0, 0)))       # ZeroPadding2D(3)  :200
    f1 = _cba(xp, w, "conv1", 2, "valid", "relu")          # :201-203 -> act_conv1
    p = maxpool_3x3_s2_same(f1)                            # :204 (padding='same': a modification)
    y = _conv_block(p, w, "res2a", 1)                      # :206
    y = _identity_block(y, w, "res2b")
    f2 = _identity_block(y, w, "res2c")                    # act2c_branch
    y = _conv_block(f2, w, "res3a", 2)                     # :210
    y = _identity_block(y, w, "res3b")
    y = _identity_block(y, w, "res3c")
    f3 = _identity_block(y, w, "res3d")                    # act3d_branch
    return f1, f2, f3


def _decoder(w, f4, s3, s2, s1):
    """Bottleneck + decoder (ae_model.py:198-236 == :108-146)."""
    n = f4.shape[0]
    enc = dense(f4.reshape(n, -1), w["dense_enc.kernel"], w["dense_enc.bias"])        # Flatten + Dense(256)
    d = dense(enc, w["dense_dec.kernel"], w["dense_dec.bias"]).reshape(n, 8, 8, 256)   # Dense + Reshape

    def up(x, name):
        return bn_act(conv2d_transpose(x, w[name + ".kernel"], w[name + ".bias"]), w, name, "leaky")

    d1 = up(d, "up1")                                                                   # :202-205
    d1 = _cba(np.concatenate([d1, s3], -1), w, "deconv1", 1, "same", "leaky")           # :207-211
    d2 = up(d1, "up2")                                                                  # :213-216
    d2 = _cba(np.concatenate([d2, s2], -1), w, "deconv2", 1, "same", "leaky")           # :217-220
    d3 = up(d2, "up3")                                                                  # :223-226
    d3 = _cba(np.concatenate([d3, s1], -1), w, "deconv3", 1, "same", "leaky")           # :227-230
    dec = conv2d_transpose(d3, w["head_xyz.kernel"], w["head_xyz.bias"])                # :233
    dec = bn_act(dec, {}, "", "tanh")                                                   # :234
    prob = conv2d_transpose(d3, w["head_prob.kernel"], w["head_prob.bias"])             # :235
    prob = bn_act(prob, {}, "", "sigmoid")                                              # :236
    return dec, prob


def forward(w: dict, x: np.ndarray, backbone: str, taps: dict | None = None):
    """x [N,128,128,3] float -> (decode [N,128,128,3] f32, prob [N,128,128,1] f32).

    ``taps`` (optional dict) receives named intermediate activations for per-layer tests.
    """
    x = _f32(x)
    assert x.ndim == 4 and x.shape[1:] == (128, 128, 3), x.shape
    if backbone == "resnet50":
        f1, f2, f3 = _resnet_front(w, x)
        s1, s2, s3 = f1[..., :32], f2[..., :128], f3[..., :128]                         # ae_model.py:186-188
        f4_1 = _cba(f3, w, "conv4_1", 2, "same", "leaky")                               # :190-192
        f4_2 = _cba(f3, w, "conv4_2", 2, "same", "leaky")                               # :193-195
        f4 = np.concatenate([f4_1, f4_2], -1)                                           # :196
    elif backbone == "paper":
        f = x
        skips = []
        for lvl in (1, 2, 3, 4):                                                        # ae_model.py:74-106
            a = _cba(f, w, "conv%d_1" % lvl, 2, "same", "leaky")
            b = _cba(f, w, "conv%d_2" % lvl, 2, "same", "leaky")
            skips.append(b)
            f = np.concatenate([a, b], -1)
        f4 = f
        s1, s2, s3 = skips[0], skips[1], skips[2]                                       # f1_2, f2_2, f3_2
        f1 = f2 = f3 = None
    else:
        raise ValueError(backbone)
    dec, prob = _decoder(w, f4, s3, s2, s1)
    if taps is not None:
        taps.update(f1=f1, f2=f2, f3=f3, f4=f4)
    return dec, prob
