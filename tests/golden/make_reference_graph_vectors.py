"""Golden vectors from the REFERENCE'S OWN model builders (pix2pose_model/ae_model.py, resnet50_mod.py).

    python tests/golden/make_reference_graph_vectors.py      (needs /root/reference; writes reference_graph.json)

Keras / TensorFlow do not exist in this image, so the builders `aemodel_unet_prob` and `aemodel_unet_resnet50` are
EXECUTED unmodified against a small stand-in for the Keras functional API defined below: every `Layer(...)(tensor)`
call records a node; evaluating the model's outputs walks exactly the graph the reference code wired -- which layer
feeds which, kernel sizes, strides, paddings, the skip slices, the nested ResNet front cut at act_conv1 /
act2c_branch / act3d_branch, Keras' automatic layer names -- and computes each layer with the oracle's layer
functions (oracle/ae_layers.c).  Layer SEMANTICS (TF 'SAME' padding, Conv2DTranspose, BatchNormalization eps 1e-3,
LeakyReLU alpha 0.3 defaults) are therefore the oracle's restatement and stay unpinned; what this pins is
  * the graph wiring of oracle/ae_oracle.py and of the HIP generator, and
  * the Keras-name -> canonical-name mapping of pix2pose_amd/convert_keras.py (row f-2): weights are drawn PER KERAS
    LAYER NAME, pushed through the reference graph, and the converted canonical dict must give the same outputs.
The fixture holds the reachable layer list (names, tensor shapes), the seeds and output probes; no reference text.
"""
import json
import os
import re
import sys
import types
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

from oracle import ae_oracle as A  # noqa: E402

_uid = {}


def _snake(name):
    s = re.sub(r"(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub(r"([a-z])([A-Z])", r"\1_\2", s).lower()


class Node:
    def __init__(self, layer, parents, index=0):
        self.layer, self.parents, self.index = layer, parents, index


class Layer:
    """Stand-in for keras.layers.Layer: records the call graph; `compute` uses the oracle's layer functions."""
    def __init__(self, *args, name=None, **kw):
        prefix = _snake(type(self).__name__)
        if name is None:
            _uid[prefix] = _uid.get(prefix, 0) + 1           # keras.backend.get_uid
            name = "%s_%d" % (prefix, _uid[prefix])
        self.name, self.args, self.kw = name, args, kw
        self.output = None

    def __call__(self, x):
        parents = list(x) if isinstance(x, (list, tuple)) else [x]
        self.output = Node(self, parents)
        return self.output

    def weights_spec(self, in_shapes):
        return {}


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Input(Layer):
    def __init__(self, shape=None, tensor=None, **kw):
        super().__init__(**kw)
        self.shape = shape

    def compute(self, xs, w):
        raise RuntimeError("unbound input")


def make_input(shape=None, **kw):
    return Input(shape=shape, **kw)([])


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", name=None, **kw):
        super().__init__(name=name)
        self.filters, self.k, self.s, self.padding = filters, _pair(kernel_size), _pair(strides), padding

    def weights_spec(self, in_shapes):
        return {"kernel": self.k + (in_shapes[0][-1], self.filters), "bias": (self.filters,)}

    def compute(self, xs, w):
        assert self.s[0] == self.s[1]
        return A.conv2d(xs[0], w["kernel"], w["bias"], self.s[0], self.padding)


class Conv2DTranspose(Layer):
    def __init__(self, filters, kernel_size=None, strides=(1, 1), padding="valid", name=None, **kw):
        super().__init__(name=name)
        self.filters, self.k, self.s, self.padding = filters, _pair(kernel_size), _pair(strides), padding

    def weights_spec(self, in_shapes):
        return {"kernel": self.k + (self.filters, in_shapes[0][-1]), "bias": (self.filters,)}

    def compute(self, xs, w):
        assert self.padding == "same" and self.s == (2, 2)
        return A.conv2d_transpose(xs[0], w["kernel"], w["bias"], 2)


class BatchNormalization(Layer):
    def __init__(self, axis=-1, epsilon=1e-3, name=None, **kw):
        super().__init__(name=name)
        assert axis in (3, -1) and epsilon == A.BN_EPS

    def weights_spec(self, in_shapes):
        c = in_shapes[0][-1]
        return {"gamma": (c,), "beta": (c,), "moving_mean": (c,), "moving_variance": (c,)}

    def compute(self, xs, w):
        d = {"b.gamma": w["gamma"], "b.beta": w["beta"], "b.mean": w["moving_mean"], "b.var": w["moving_variance"]}
        return A.bn_act(np.array(xs[0], np.float32), d, "b", "none")


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3, **kw):
        super().__init__(**kw)
        assert alpha == A.LEAKY

    def compute(self, xs, w):
        return A.bn_act(np.array(xs[0], np.float32), {}, "", "leaky")


class Activation(Layer):
    def __init__(self, activation, name=None, **kw):
        super().__init__(name=name)
        self.activation = activation

    def compute(self, xs, w):
        return A.bn_act(np.array(xs[0], np.float32), {}, "", self.activation)


class Dense(Layer):
    def __init__(self, units, activation=None, name=None, **kw):
        super().__init__(name=name)
        assert activation is None
        self.units = units

    def weights_spec(self, in_shapes):
        return {"kernel": (in_shapes[0][-1], self.units), "bias": (self.units,)}

    def compute(self, xs, w):
        return A.dense(xs[0], w["kernel"], w["bias"])


class Flatten(Layer):
    def compute(self, xs, w):
        return np.ascontiguousarray(xs[0]).reshape(xs[0].shape[0], -1)


class Reshape(Layer):
    def __init__(self, target_shape, **kw):
        super().__init__(**kw)
        self.target = tuple(target_shape)

    def compute(self, xs, w):
        return np.ascontiguousarray(xs[0]).reshape((xs[0].shape[0],) + self.target)


class Concatenate(Layer):
    def __init__(self, axis=-1, **kw):
        super().__init__(**kw)
        assert axis == -1

    def compute(self, xs, w):
        return np.concatenate(xs, -1)


class Lambda(Layer):
    def __init__(self, function, **kw):
        super().__init__(**kw)
        self.fn = function

    def compute(self, xs, w):
        return np.ascontiguousarray(self.fn(xs[0]))


class ZeroPadding2D(Layer):
    def __init__(self, padding=(1, 1), name=None, **kw):
        super().__init__(name=name)
        self.p = _pair(padding)

    def compute(self, xs, w):
        return np.pad(xs[0], ((0, 0), (self.p[0],) * 2, (self.p[1],) * 2, (0, 0)))


class MaxPooling2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", name=None, **kw):
        super().__init__(name=name)
        assert _pair(pool_size) == (3, 3) and _pair(strides) == (2, 2) and padding == "same"

    def compute(self, xs, w):
        return A.maxpool_3x3_s2_same(np.ascontiguousarray(xs[0], np.float32))


class Add(Layer):
    def compute(self, xs, w):
        return (np.asarray(xs[0], np.float32) + np.asarray(xs[1], np.float32)).astype(np.float32)


class _Unused(Layer):
    def compute(self, xs, w):
        raise RuntimeError("%s is not on the inference path" % type(self).__name__)


class Model(Layer):
    def __init__(self, inputs=None, outputs=None, name=None, **kw):
        super().__init__(name=name)
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.multi = isinstance(outputs, (list, tuple))
        self.outputs = list(outputs) if self.multi else [outputs]
        self.input = self.inputs[0]
        self._layers = {}
        for n in walk(self.outputs):
            self._layers[n.layer.name] = n.layer

    def get_layer(self, name=None):
        return self._layers[name]

    def load_weights(self, *a, **k):
        pass

    def __call__(self, x):                       # a model used as a layer (the nested ResNet front)
        nodes = [Node(self, [x], i) for i in range(len(self.outputs))]
        return nodes if self.multi else nodes[0]


def walk(outputs):
    seen, order = set(), []

    def rec(n):
        if id(n) in seen:
            return
        seen.add(id(n))
        for p in n.parents:
            rec(p)
        if isinstance(n.layer, Model):
            for o in n.layer.outputs:
                rec(o)
        order.append(n)
    for o in outputs:
        rec(o)
    return order


def evaluate(model, x, weights, shapes_only=False):
    """Evaluate model.outputs for input array x.  weights: {layer name: {kind: array}}; with shapes_only the layers'
    weight tensors are created on the fly (zeros) and their specs collected."""
    specs = {}

    def run(node, binding, cache):
        key = id(node)
        if key in cache:
            return cache[key]
        L = node.layer
        if isinstance(L, Input):
            val = binding[id(node)]
        elif isinstance(L, Model):
            inner = {id(L.inputs[0]): run(node.parents[0], binding, cache)}
            val = run(L.outputs[node.index], inner, {})
        else:
            xs = [run(p, binding, cache) for p in node.parents]
            spec = L.weights_spec([v.shape for v in xs])
            if spec:
                specs[L.name] = {k: list(v) for k, v in spec.items()}
                w = {k: np.zeros(v, np.float32) for k, v in spec.items()} if shapes_only else weights[L.name]
                if shapes_only and "moving_variance" in w:
                    w["moving_variance"] += 1
            else:
                w = {}
            val = L.compute(xs, w)
        cache[key] = val
        return val
    outs = [run(o, {id(model.inputs[0]): x}, {}) for o in model.outputs]
    return outs, specs


def install():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    L = dict(Input=make_input, Conv2D=Conv2D, MaxPooling2D=MaxPooling2D, UpSampling2D=_Unused, Conv2DTranspose=Conv2DTranspose,
             ZeroPadding2D=ZeroPadding2D, Flatten=Flatten, Dense=Dense, Dropout=_Unused, Activation=Activation, RepeatVector=_Unused,
             Lambda=Lambda, Reshape=Reshape, Subtract=_Unused, Concatenate=Concatenate, Layer=Layer, merge=None,
             AveragePooling2D=_Unused, GlobalAveragePooling2D=_Unused, GlobalMaxPooling2D=_Unused, BatchNormalization=BatchNormalization,
             add=lambda xs: Add()(xs))
    layers = mod("keras.layers", **L)
    mod("keras.layers.normalization", BatchNormalization=BatchNormalization)
    mod("keras.layers.advanced_activations", LeakyReLU=LeakyReLU)
    backend = mod("keras.backend", image_data_format=lambda: "channels_last", is_keras_tensor=lambda t: isinstance(t, Node),
                  backend=lambda: "tensorflow")
    models = mod("keras.models", Model=Model, load_model=None)
    keras = mod("keras", layers=layers, backend=backend, models=models)
    keras.initializers = mod("keras.initializers", glorot_normal=None)
    keras.regularizers = mod("keras.regularizers", l2=lambda v: None)
    keras.losses = mod("keras.losses")
    keras.optimizers = mod("keras.optimizers")
    mod("keras.callbacks", TensorBoard=None, ModelCheckpoint=None)
    mod("keras.engine")
    mod("keras.engine.topology", get_source_inputs=None)
    mod("keras.utils", layer_utils=None)
    mod("keras.utils.layer_utils")
    sys.modules["keras.utils"].layer_utils = sys.modules["keras.utils.layer_utils"]
    mod("keras.utils.data_utils", get_file=lambda *a, **k: "")
    mod("keras.applications")
    mod("keras.applications.imagenet_utils", decode_predictions=None, preprocess_input=None,
        _obtain_input_shape=lambda input_shape, **k: input_shape)
    mod("tensorflow")
    pm = mod("pix2pose_model")
    pm.__path__ = [os.path.join(REF, "pix2pose_model")]


def layer_weights(name, spec, seed):
    """Deterministic weights of one Keras layer, keyed by its NAME (so any name mix-up changes the output)."""
    rs = np.random.RandomState((zlib.crc32(name.encode()) + seed) % (2 ** 31))
    w = {}
    for kind in sorted(spec):
        shp = tuple(spec[kind])
        if kind == "kernel":
            fan_in = int(np.prod(shp[:-1])) if len(shp) != 4 or "transpose" not in name else int(shp[0] * shp[1] * shp[3])
            w[kind] = (rs.randn(*shp) * np.sqrt(1.0 / max(fan_in, 1))).astype(np.float32)
        elif kind == "moving_variance":
            w[kind] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif kind == "gamma":
            w[kind] = rs.uniform(0.8, 1.2, shp).astype(np.float32)
        else:
            w[kind] = (rs.randn(*shp) * 0.05).astype(np.float32)
    return w


def probes(d, p):
    rs = np.random.RandomState(42)
    idx = rs.randint(0, d.shape[0] * 128 * 128, 48)
    return {"pixel_index": idx.tolist(), "decode": d.reshape(-1, 3)[idx].astype(float).tolist(), "prob": p.reshape(-1)[idx].astype(float).tolist(),
            "decode_abs_mean": float(np.abs(d).astype(np.float64).mean()), "prob_mean": float(p.astype(np.float64).mean())}


def main():
    install()
    sys.path.insert(0, REF)
    from pix2pose_model import ae_model as ref_ae
    out = {"note": "outputs of the graphs built by /root/reference/pix2pose_model/ae_model.py (+ resnet50_mod.py) with the Keras API "
                   "stood in and the layers computed by the oracle, see tests/golden/make_reference_graph_vectors.py",
           "weights_seed": 7, "input_seed": 3, "n": 2, "graphs": {}}
    x = ((np.random.RandomState(out["input_seed"]).randint(0, 256, (out["n"], 128, 128, 3)).astype(np.float32) - 128) / 128)
    for backbone, builder in (("paper", ref_ae.aemodel_unet_prob), ("resnet50", ref_ae.aemodel_unet_resnet50)):
        _uid.clear()
        model = builder(p=1.0)
        _, specs = evaluate(model, x[:1], None, shapes_only=True)
        weights = {name: layer_weights(name, sp, out["weights_seed"]) for name, sp in specs.items()}
        (d, p), _ = evaluate(model, x, weights)
        out["graphs"][backbone] = {"layers": specs, "probes": probes(d, p)}
        print(backbone, len(specs), "weighted layers; |decode| mean", out["graphs"][backbone]["probes"]["decode_abs_mean"])
    fn = os.path.join(HERE, "reference_graph.json")
    with open(fn, "w") as f:
        json.dump(out, f)
    print("wrote", fn, os.path.getsize(fn), "bytes")


if __name__ == "__main__":
    main()
