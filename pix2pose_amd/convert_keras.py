"""HDF5 (Keras 2.2) -> .npz artefact converter for the reference's trained weights -- SURVEY 8f-2.

The reference stores two kinds of files (tools/3_train_pix2pose.py:273-276,
tools/4_convert_weights_inference.py:38-52; picked up at tools/5_evaluation_bop_basic.py:209-215):

  * ``inference.hdf5``              weights-only (``save_weights``) of ``aemodel_unet_prob``  ("paper")
  * ``inference_resnet_model.hdf5`` full ``model.save`` of ``aemodel_unet_resnet50``: the weights sit
    under ``model_weights`` and the ResNet front is a *nested* ``Model`` layer (``model_N``) whose
    tensors keep their explicit names (``conv1``, ``bn_conv1``, ``res2a_branch2a``, ``bn2a_branch2a`` ...)

Named layers (``conv1_1`` ... ``conv4_2``, ``deconv1..3``, the ResNet ones) are matched by name.
Auto-named layers (``batch_normalization_K``, ``dense_K``, ``conv2d_transpose_K``) carry session-global
counters, so only their *relative* order is meaningful: they are sorted by K and assigned in the
order the reference's builder creates them (ae_model.py:74-146 / :190-236).

h5py is not part of the GPU image; run this where it is available:

    python -m pix2pose_amd.convert_keras inference_resnet_model.hdf5 resnet50 obj01.npz

``convert_named()`` (the mapping itself) has no h5py dependency and is unit-tested.
"""
from __future__ import annotations

import re
import sys

import numpy as np

from . import weights as W

# builder order of the auto-named layers (ae_model.py)
_BN_ORDER = {
    "paper": ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv4_1", "conv4_2",
              "up1", "deconv1", "up2", "deconv2", "up3", "deconv3"],
    "resnet50": ["conv4_1", "conv4_2", "up1", "deconv1", "up2", "deconv2", "up3", "deconv3"],
}
_DENSE_ORDER = ["dense_enc", "dense_dec"]
_DECONV_ORDER = ["up1", "up2", "up3", "head_xyz", "head_prob"]
_KINDS = {"kernel": "kernel", "bias": "bias", "gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}


def _resnet_name(layer: str):
    """'res2a_branch2a' -> ('res2a_2a', conv) ; 'bn2a_branch1' -> ('res2a_1', bn) ; conv1 / bn_conv1."""
    if layer in ("conv1", "bn_conv1"):
        return "conv1"
    m = re.match(r"^(res|bn)(\d[a-z])_branch(2a|2b|2c|1)$", layer)
    if m:
        return "res%s_%s" % (m.group(2), m.group(3))
    return None


def convert_named(keras: dict, backbone: str) -> dict:
    """keras: {'<layer>/<weight>': ndarray} with weight in kernel|bias|gamma|beta|moving_mean|
    moving_variance (any ':0' suffix / nested 'model_N/' prefix already stripped or not).
    Returns the canonical tensor dict of pix2pose_amd.weights.tensor_specs(backbone)."""
    per_layer = {}
    for key, arr in keras.items():
        parts = [p for p in key.replace(":0", "").split("/") if p]
        layer, kind = parts[-2], parts[-1]
        if kind not in _KINDS:
            continue
        per_layer.setdefault(layer, {})[_KINDS[kind]] = np.asarray(arr, np.float32)

    def numbered(prefix):
        found = []
        for name in per_layer:
            m = re.match(r"^%s_(\d+)$" % prefix, name)
            if m:
                found.append((int(m.group(1)), name))
        return [n for _, n in sorted(found)]

    out = {}

    def put(canon, layer, kinds):
        for k in kinds:
            if k not in per_layer[layer]:
                raise ValueError("layer %r has no %r tensor" % (layer, k))
            out["%s.%s" % (canon, k)] = per_layer[layer][k]

    # explicitly named conv layers
    for layer in list(per_layer):
        canon = None
        if re.match(r"^conv[1-4]_[12]$", layer) or re.match(r"^deconv[123]$", layer):
            canon = layer
        elif backbone == "resnet50":
            canon = _resnet_name(layer)
        if canon is None:
            continue
        if layer.startswith("bn"):
            put(canon, layer, ["gamma", "beta", "mean", "var"])
        else:
            put(canon, layer, ["kernel", "bias"])
    # auto-named layers, by creation order
    bns = numbered("batch_normalization")
    if len(bns) != len(_BN_ORDER[backbone]):
        raise ValueError("expected %d batch_normalization_* layers for %s, found %d" % (len(_BN_ORDER[backbone]), backbone, len(bns)))
    for canon, layer in zip(_BN_ORDER[backbone], bns):
        put(canon, layer, ["gamma", "beta", "mean", "var"])
    dn, dc = numbered("dense"), numbered("conv2d_transpose")
    if len(dn) != 2 or len(dc) != 5:
        raise ValueError("expected 2 dense_* and 5 conv2d_transpose_* layers, found %d and %d" % (len(dn), len(dc)))
    for canon, layer in zip(_DENSE_ORDER, dn):
        put(canon, layer, ["kernel", "bias"])
    for canon, layer in zip(_DECONV_ORDER, dc):
        put(canon, layer, ["kernel", "bias"])
    W.check_weights(backbone, out)
    return out


def read_hdf5(path: str) -> dict:
    """Flatten a Keras HDF5 weight / model file into {'layer/weight': ndarray} (needs h5py)."""
    import h5py  # noqa: not available in the GPU image
    flat = {}
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                flat[name] = np.array(obj)
        root.visititems(visit)
    return flat


def main(argv):
    if len(argv) != 4:
        print("usage: python -m pix2pose_amd.convert_keras <keras.hdf5> <paper|resnet50> <out.npz>")
        return 2
    W.save_weights(argv[3], argv[2], convert_named(read_hdf5(argv[1]), argv[2]))
    print("wrote", argv[3])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
