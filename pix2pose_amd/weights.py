"""Weight artefact for the Pix2Pose autoencoder hot path.

Tensor inventory (names, Keras-native layouts) for the two generator graphs the
reference builds:

  * ``resnet50``  -> ``aemodel_unet_resnet50``  (reference pix2pose_model/ae_model.py:175-240,
                     front = pix2pose_model/resnet50_mod.py:200-213, blocks :40-118)
  * ``paper``     -> ``aemodel_unet_prob``      (reference pix2pose_model/ae_model.py:70-150)

Layouts are exactly what Keras 2.2 stores (so a converter from the reference's
``inference*.hdf5`` only has to rename):

  Conv2D          kernel (kh, kw, Cin, Cout)   bias (Cout,)
  Conv2DTranspose kernel (kh, kw, Cout, Cin)   bias (Cout,)
  Dense           kernel (in, out)             bias (out,)
  BatchNorm       gamma, beta, mean, var       (C,)   eps = 1e-3 (Keras default)

The on-disk artefact is a plain ``.npz`` holding these tensors under the names
below plus a ``__backbone__`` string.  ``synthetic:<backbone>:<seed>`` is accepted
everywhere a weight file name is (there are no trained weights offline).

The synthetic generator is a counter-based integer hash so that every value is
bit-reproducible on any host (no libm involved): see ``_hash_normal``.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-3          # Keras BatchNormalization default (ae_model.py:75 etc. use defaults)
LEAKY_ALPHA = 0.3      # keras.layers.LeakyReLU() default (ae_model.py:16,192,...)

BACKBONES = ("paper", "resnet50")


def _conv(name, kh, kw, cin, cout, bn=True):
    t = [(name + ".kernel", (kh, kw, cin, cout)), (name + ".bias", (cout,))]
    if bn:
        t += _bn(name, cout)
    return t


def _deconv(name, cin, cout, bn=True):
    t = [(name + ".kernel", (5, 5, cout, cin)), (name + ".bias", (cout,))]
    if bn:
        t += _bn(name, cout)
    return t


def _bn(name, c):
    return [(name + ".gamma", (c,)), (name + ".beta", (c,)),
            (name + ".mean", (c,)), (name + ".var", (c,))]


def _dense(name, cin, cout):
    return [(name + ".kernel", (cin, cout)), (name + ".bias", (cout,))]


def _res_block(stage, block, cin, f1, f2, f3, shortcut):
    base = "res%d%s" % (stage, block)
    t = _conv(base + "_2a", 1, 1, cin, f1)
    t += _conv(base + "_2b", 3, 3, f1, f2)
    t += _conv(base + "_2c", 1, 1, f2, f3)
    if shortcut:
        t += _conv(base + "_1", 1, 1, cin, f3)
    return t


def _decoder(skip3, skip2, skip1):
    """Bottleneck + decoder shared by both graphs (ae_model.py:108-146 == :198-236)."""
    t = _dense("dense_enc", 8 * 8 * 512, 256)
    t += _dense("dense_dec", 256, 8 * 8 * 256)
    t += _deconv("up1", 256, 256)
    t += _conv("deconv1", 5, 5, 256 + skip3, 256)
    t += _deconv("up2", 256, 128)
    t += _conv("deconv2", 5, 5, 128 + skip2, 256)
    t += _deconv("up3", 256, 64)
    t += _conv("deconv3", 5, 5, 64 + skip1, 128)
    t += _deconv("head_xyz", 128, 3, bn=False)
    t += _deconv("head_prob", 128, 1, bn=False)
    return t


def tensor_specs(backbone: str):
    """Ordered [(name, shape)] for one generator graph."""
    if backbone == "resnet50":
        t = _conv("conv1", 7, 7, 3, 64)                                   # resnet50_mod.py:200-203
        t += _res_block(2, "a", 64, 64, 64, 256, True)                    # :206
        t += _res_block(2, "b", 256, 64, 64, 256, False)                  # :207
        t += _res_block(2, "c", 256, 64, 64, 256, False)                  # :208
        t += _res_block(3, "a", 256, 128, 128, 512, True)                 # :210
        for b in "bcd":                                                   # :211-213
            t += _res_block(3, b, 512, 128, 128, 512, False)
        t += _conv("conv4_1", 5, 5, 512, 256)                             # ae_model.py:190-192
        t += _conv("conv4_2", 5, 5, 512, 256)                             # ae_model.py:193-195
        t += _decoder(128, 128, 32)                                       # ae_model.py:186-188
        return t
    if backbone == "paper":
        t = []
        for lvl, (cin, cout) in enumerate([(3, 64), (128, 128), (256, 128), (256, 256)], 1):
            t += _conv("conv%d_1" % lvl, 5, 5, cin, cout)                 # ae_model.py:74-106
            t += _conv("conv%d_2" % lvl, 5, 5, cin, cout)
        t += _decoder(128, 128, 64)
        return t
    raise ValueError("unknown backbone %r (expected 'paper' or 'resnet50')" % (backbone,))


def n_params(backbone: str) -> int:
    return int(sum(int(np.prod(s)) for _, s in tensor_specs(backbone)))


# --------------------------------------------------------------------------------------
# portable counter-based generator
# --------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _sm64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def hash_u64(seed: int, stream: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        key = _sm64(np.array([(seed * 0x100000001B3 + stream) & 0xFFFFFFFFFFFFFFFF], np.uint64))
        return _sm64(key + np.arange(n, dtype=np.uint64))


def hash_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """U[0,1) float64, exactly reproducible."""
    return (hash_u64(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _hash_normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Approximately N(0,1): sum of four 16-bit uniforms, rescaled to unit variance.

    Integer sum then one multiply by a constant -> bit-identical everywhere.
    Support is +-3.46 sigma, which is all a synthetic He-initialisation needs.
    """
    u = hash_u64(seed, stream, n)
    m = np.uint64(0xFFFF)
    s = ((u & m) + ((u >> np.uint64(16)) & m) + ((u >> np.uint64(32)) & m) + (u >> np.uint64(48))).astype(np.float64)
    return (s - 131070.0) * (1.7320508075688772 / 65536.0)


def synthetic_weights(backbone: str, seed: int = 1) -> dict:
    """Deterministic synthetic weights: He-normal(fan_in) kernels,
    bias~N(0,0.05), BN gamma=1 (0.4 on residual-branch outputs), beta~N(0,0.1), mean~N(0,0.1), var~U(0.5,1.5)."""
    out = {}
    for stream, (name, shape) in enumerate(tensor_specs(backbone)):
        n = int(np.prod(shape))
        kind = name.rsplit(".", 1)[1]
        if kind == "kernel":
            if len(shape) == 4:
                kh, kw, a, b = shape
                # Conv2D (kh,kw,Cin,Cout): fan_in = kh*kw*Cin.  Conv2DTranspose (kh,kw,Cout,Cin):
                # each output sees on average kh*kw/4 taps of Cin channels at stride 2.
                is_t = name.startswith("up") or name.startswith("head_")
                fan_in = (kh * kw * b / 4.0) if is_t else (kh * kw * a)
            else:
                fan_in = shape[0]
            # He-normal, except: linear layers (Dense, heads) use gain 1 / 0.5 so the
            # bottleneck does not inflate and tanh/sigmoid are not saturated.
            gain = 2.0
            if name.startswith("dense_"):
                gain = 1.0
            elif name.startswith("head_"):
                gain = 0.5
            v = _hash_normal(seed, stream, n) * np.sqrt(gain / fan_in)
        elif kind == "bias":
            v = 0.05 * _hash_normal(seed, stream, n)      # non-zero so a dropped bias is visible
        elif kind == "gamma":
            # residual-branch output BNs are damped so seven stacked blocks keep O(1) activations
            v = np.full(n, 0.4 if name.split(".")[0].endswith("_2c") else 1.0)
        elif kind in ("beta", "mean"):
            v = 0.1 * _hash_normal(seed, stream, n)
        elif kind == "var":
            v = 0.5 + hash_uniform(seed, stream, n)
        else:  # pragma: no cover
            raise AssertionError(name)
        out[name] = np.ascontiguousarray(v.reshape(shape).astype(np.float32))
    return out


def trained_like_weights(backbone: str, seed: int = 1) -> dict:
    """Synthetic weights with the statistics of a TRAINED network instead of a freshly initialised one: BatchNorm moving
    variances spread over four decades (log-uniform 1e-3 .. 1e1), gammas log-uniform 0.05 .. 2, and every layer's kernel /
    bias / moving mean rescaled per output channel so that its pre-BN activations really have that variance (which is what
    training produces: BN statistics follow the layer, not the other way round).  The network function stays O(1) -- the
    per-channel factors cancel inside BN up to the eps term -- but the operands the kernels see do not: weight rows differ by
    two decades within a layer and the folded BN scales by three.  Used by the precision tests of the split-f16 arithmetic."""
    w = synthetic_weights(backbone, seed)
    for stream, (name, shape) in enumerate(tensor_specs(backbone)):
        if not name.endswith(".var"):
            continue
        base = name[:-4]
        c = shape[0]
        u1 = hash_uniform(seed + 7919, stream, c)
        u2 = hash_uniform(seed + 104729, stream, c)
        var = np.exp(np.log(1e-3) + u1 * (np.log(1e1) - np.log(1e-3)))
        gamma_hi = 0.8 if base.endswith("_2c") else 2.0          # residual-branch outputs stay damped (seven stacked blocks)
        gamma = np.exp(np.log(0.05) + u2 * (np.log(gamma_hi) - np.log(0.05)))
        f = np.sqrt(var / w[name].astype(np.float64))             # per-channel factor on the pre-BN activation
        k = w[base + ".kernel"].astype(np.float64)
        is_t = k.ndim == 4 and (base.startswith("up") or base.startswith("head_"))
        k = k * (f[None, None, :, None] if is_t else f)           # Conv2DTranspose kernels are (kh,kw,Cout,Cin)
        w[base + ".kernel"] = k.astype(np.float32)
        w[base + ".bias"] = (w[base + ".bias"].astype(np.float64) * f).astype(np.float32)
        w[base + ".mean"] = (w[base + ".mean"].astype(np.float64) * f).astype(np.float32)
        w[name] = var.astype(np.float32)
        w[base + ".gamma"] = gamma.astype(np.float32)
    return w


def check_weights(backbone: str, w: dict) -> None:
    """Raise ValueError on a missing / mis-shaped tensor."""
    for name, shape in tensor_specs(backbone):
        if name not in w:
            raise ValueError("weight tensor %r missing for backbone %r" % (name, backbone))
        if tuple(w[name].shape) != tuple(shape):
            raise ValueError("weight tensor %r has shape %r, expected %r" % (name, tuple(w[name].shape), shape))


def save_weights(path: str, backbone: str, w: dict) -> None:
    check_weights(backbone, w)
    np.savez(path, __backbone__=np.array(backbone), **{k: w[k] for k, _ in tensor_specs(backbone)})


def load_weights(weight_fn: str, backbone: str) -> dict:
    """``weight_fn`` is a ``.npz`` artefact, ``synthetic:<backbone>:<seed>`` / ``trained-like:<backbone>:<seed>``, or the
    reference's Keras ``.hdf5`` / ``.h5`` inference weights (read through convert_keras; needs h5py)."""
    if weight_fn.startswith("trained-like:"):
        _, bb, seed = weight_fn.split(":")
        if bb != backbone:
            raise ValueError("weights are for backbone %r, not %r" % (bb, backbone))
        return trained_like_weights(backbone, int(seed))
    if weight_fn.startswith("synthetic:"):
        parts = weight_fn.split(":")
        bb = parts[1] if len(parts) > 1 and parts[1] else backbone
        if bb != backbone:
            raise ValueError("weight spec %r does not match backbone %r" % (weight_fn, backbone))
        seed = int(parts[2]) if len(parts) > 2 else 1
        return synthetic_weights(backbone, seed)
    if weight_fn.endswith((".hdf5", ".h5")):
        from . import convert_keras
        return convert_keras.convert_named(convert_keras.read_hdf5(weight_fn), backbone)
    with np.load(weight_fn, allow_pickle=False) as z:
        bb = str(z["__backbone__"]) if "__backbone__" in z.files else backbone
        if bb != backbone:
            raise ValueError("weight file %r was written for backbone %r, not %r" % (weight_fn, bb, backbone))
        w = {k: np.ascontiguousarray(z[k], dtype=np.float32) for k, _ in tensor_specs(backbone)}
    check_weights(backbone, w)
    return w
