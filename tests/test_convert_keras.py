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


def test_read_hdf5_on_keras_file_layouts(monkeypatch, tmp_path):
    """The HDF5 reader (row f-2) on the two file layouts the reference produces (tools/3_train_pix2pose.py:273-276): a
    weights-only file (layer groups at the root) and a full model.save file (everything under 'model_weights', plus an
    'optimizer_weights' group that must be ignored), with Keras' doubled group names ('dense_3/dense_3/kernel:0') and the
    nested 'model_1' ResNet front.  h5py itself is not in this image, so a dict-backed stand-in with the h5py API
    (File / Group / Dataset, `in`, [], visititems) carries the tree; with real files the traversal is the same
    (tools/make_external_vectors.py exercises that where h5py and Keras exist)."""
    import sys
    import types

    class Dataset:
        def __init__(self, a):
            self.a = np.asarray(a)

        def __array__(self, dtype=None, copy=None):
            return self.a if dtype is None else self.a.astype(dtype)

    class Group:
        def __init__(self, tree):
            self.tree = tree

        def __contains__(self, k):
            return k in self.tree

        def __getitem__(self, k):
            v = self.tree[k]
            return Group(v) if isinstance(v, dict) else v

        def visititems(self, fn, prefix=""):
            for k, v in self.tree.items():
                name = prefix + k
                if isinstance(v, dict):
                    fn(name, Group(v))
                    Group(v).visititems(fn, name + "/")
                else:
                    fn(name, v)

    class File(Group):
        trees = {}

        def __init__(self, path, mode="r"):
            super().__init__(File.trees[path])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    h5 = types.ModuleType("h5py")
    h5.File, h5.Group, h5.Dataset = File, Group, Dataset
    monkeypatch.setitem(sys.modules, "h5py", h5)

    def tree_of(keras):
        root = {}
        for key, arr in keras.items():
            node = root
            parts = key.split("/")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = Dataset(arr)
        return root

    for backbone, offset in (("paper", 1), ("resnet50", 22)):
        w = W.synthetic_weights(backbone, 6)
        keras = _fake_keras(backbone, w, offset)
        File.trees["weights.hdf5"] = tree_of(keras)
        File.trees["model.hdf5"] = {"model_weights": tree_of(keras), "optimizer_weights": {"Adam": {"iterations:0": Dataset(np.zeros(1))}}}
        for fn in ("weights.hdf5", "model.hdf5"):
            flat = CK.read_hdf5(fn)
            assert not any("optimizer" in k or "Adam" in k for k in flat)
            out = CK.convert_named(flat, backbone)
            assert set(out) == set(w) and all(np.array_equal(out[k], w[k]) for k in w), (backbone, fn)
    # the command-line entry point writes the artefact the runtime loads
    File.trees["weights.hdf5"] = tree_of(_fake_keras("paper", W.synthetic_weights("paper", 6), 1))
    out_fn = str(tmp_path / "obj.npz")
    assert CK.main(["convert_keras", "weights.hdf5", "paper", out_fn]) == 0
    w2 = W.load_weights(out_fn, "paper")
    assert np.array_equal(w2["deconv2.kernel"], W.synthetic_weights("paper", 6)["deconv2.kernel"])


def test_read_hdf5_on_real_files(tmp_path):
    """The same two layouts written as REAL HDF5 files with plain h5py (no Keras needed): layer groups with Keras' `weight_names`
    attributes, a `layer_names` attribute at the root, the nested `model_1` front, and -- for the model.save layout -- the
    `model_weights` / `optimizer_weights` groups and a `model_config` attribute.  Skipped where h5py is not installed (this image);
    the dict-backed stand-in above imitates h5py by construction, this one does not."""
    h5py = pytest.importorskip("h5py")

    def write(path, keras, full_model):
        with h5py.File(path, "w") as f:
            root = f.create_group("model_weights") if full_model else f
            layers = sorted({k.split("/")[0] for k in keras})
            root.attrs["layer_names"] = [n.encode() for n in layers]
            root.attrs["backend"] = b"tensorflow"
            root.attrs["keras_version"] = b"2.2.1"
            for key, arr in keras.items():
                root.create_dataset(key, data=np.asarray(arr))
            for layer in layers:
                root[layer].attrs["weight_names"] = [k[len(layer) + 1:].encode() for k in keras if k.startswith(layer + "/")]
            if full_model:
                f.attrs["model_config"] = b'{"class_name": "Model"}'
                opt = f.create_group("optimizer_weights")
                opt.create_dataset("Adam/iterations:0", data=np.zeros(1))

    for backbone, offset in (("paper", 1), ("resnet50", 22)):
        w = W.synthetic_weights(backbone, 6)
        keras = _fake_keras(backbone, w, offset)
        for fn, full in (("weights.hdf5", False), ("model.hdf5", True)):
            path = str(tmp_path / ("%s_%s" % (backbone, fn)))
            write(path, keras, full)
            flat = CK.read_hdf5(path)
            assert not any("optimizer" in k or "Adam" in k for k in flat)
            out = CK.convert_named(flat, backbone)
            assert set(out) == set(w) and all(np.array_equal(out[k], w[k]) for k in w), (backbone, fn)
