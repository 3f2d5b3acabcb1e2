"""the Keras-name -> artefact mapping (h5py-free part), exercised on a fake HDF5 key
set built the way Keras 2.2 names things (nested 'model_1/' ResNet front, session-global counters
on auto-named layers that do not start at 1)."""
import numpy as np
import pytest

from pix2pose_amd import convert_keras as CK
from pix2pose_amd import weights as W

_K2 = {"kernel": "kernel:0", "bias": "bias:0", "gamma": "gamma:0", "beta": "beta:0", "mean": "moving_mean:0", "var": "moving_variance:0"}


def _fake_keras(backbone, w, offset):
    """Invert the mapping: canonical dict -> Keras-style keys, auto counters starting at `offset`."""
    keras = {}
    bn_i = dense_i = dc_i = 0
    auto_bn = {c: "batch_normalization_%d" % (offset + i) for i, c in enumerate(CK._BN_ORDER[backbone])}
    auto_dense = {c: "dense_%d" % (offset + 3 + i) for i, c in enumerate(CK._DENSE_ORDER)}
    auto_dc = {c: "conv2d_transpose_%d" % (offset + 7 + i) for i, c in enumerate(CK._DECONV_ORDER)}
    for name, arr in w.items():
        canon, kind = name.rsplit(".", 1)
        if kind in ("kernel", "bias"):
            if canon in auto_dense:
                layer = auto_dense[canon]
            elif canon in auto_dc:
                layer = auto_dc[canon]
            elif canon.startswith("res"):
                layer = "model_1/res%s_branch%s" % (canon[3:5], canon.split("_")[1])
            elif canon == "conv1" and backbone == "resnet50":
                layer = "model_1/conv1"
            else:
                layer = canon
        else:
            if canon in auto_bn:
                layer = auto_bn[canon]
            elif canon.startswith("res"):
                layer = "model_1/bn%s_branch%s" % (canon[3:5], canon.split("_")[1])
            else:
                layer = "model_1/bn_conv1"
        lname = layer.split("/")[-1]
        keras["%s/%s/%s" % (layer, lname, _K2[kind])] = arr
    return keras


@pytest.mark.parametrize("backbone,offset", [("paper", 1), ("paper", 15), ("resnet50", 29)])
def test_mapping_roundtrip(backbone, offset):
    w = W.synthetic_weights(backbone, 4)
    out = CK.convert_named(_fake_keras(backbone, w, offset), backbone)
    assert set(out) == set(w)
    for k in w:
        np.testing.assert_array_equal(out[k], w[k])


def test_mapping_rejects_incomplete_files():
    w = W.synthetic_weights("paper", 4)
    keras = _fake_keras("paper", w, 1)
    bad = {k: v for k, v in keras.items() if "batch_normalization_3/" not in k}
    with pytest.raises(ValueError):
        CK.convert_named(bad, "paper")
    assert CK._resnet_name("res3d_branch2c") == "res3d_2c" and CK._resnet_name("bn2a_branch1") == "res2a_1"
    assert CK._resnet_name("bn_conv1") == "conv1" and CK._resnet_name("conv4_1") is None
