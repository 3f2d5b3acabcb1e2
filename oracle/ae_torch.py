"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Independent second formulation of the generator forward pass (torch CPU ops, fp32 or fp64).  It shares no
code with oracle/ae_oracle.py (different conv engine -- oneDNN --, NCHW layout, padding done with F.pad,
transposed conv via the cropped full output) and serves two purposes:
  * tests cross-check ae_oracle.py against it (fp64, agreement 1e-5);
  * bench.py's `cpu_baseline` leg times it in fp32 on all host cores: it is the closest stand-in available here
    for what the reference's Keras-on-CPU path does ("CPU baseline").
Graph: reference pix2pose_model/ae_model.py:70-150,175-240; resnet50_mod.py:40-118,200-213.
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-3
ALPHA = 0.3


def _same(x, k, s):
    h = x.shape[-2]
    out = -(-h // s)
    tot = max((out - 1) * s + k - h, 0)
    b = tot // 2
    return F.pad(x, (b, tot - b, b, tot - b))


def _t(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt)


def _conv(x, w, name, stride, same, dt):
    k = _t(w[name + ".kernel"], dt).permute(3, 2, 0, 1)      # HWIO -> OIHW
    if same:
        x = _same(x, k.shape[-1], stride)
    return F.conv2d(x, k, _t(w[name + ".bias"], dt), stride=stride)


def _bn(x, w, name, dt):
    g, b, m, v = (_t(w[name + "." + s], dt).view(1, -1, 1, 1) for s in ("gamma", "beta", "mean", "var"))
    return g * (x - m) / torch.sqrt(v + EPS) + b


def _cba(x, w, name, stride, same, act, dt):
    y = _bn(_conv(x, w, name, stride, same, dt), w, name, dt)
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, ALPHA)
    return y


def _deconv(x, w, name, dt):
    k = _t(w[name + ".kernel"], dt).permute(3, 2, 0, 1)      # (kh,kw,Cout,Cin) -> (Cin,Cout,kh,kw)
    n = x.shape[-1]
    full = F.conv_transpose2d(x, k, _t(w[name + ".bias"], dt), stride=2)   # size 2n+3
    return full[..., 1:2 * n + 1, 1:2 * n + 1]


def forward(w, x, backbone, dtype=torch.float64):
    dt = dtype
    x = _t(x, dt).permute(0, 3, 1, 2)
    taps = {}
    if backbone == "resnet50":
        f1 = _cba(F.pad(x, (3, 3, 3, 3)), w, "conv1", 2, False, "relu", dt)
        p = F.max_pool2d(F.pad(f1, (0, 1, 0, 1), value=float("-inf")), 3, 2)

        def block(y, base, stride, shortcut):
            z = _cba(y, w, base + "_2a", stride, False, "relu", dt)
            z = _cba(z, w, base + "_2b", 1, True, "relu", dt)
            z = _cba(z, w, base + "_2c", 1, False, "none", dt)
            sc = _cba(y, w, base + "_1", stride, False, "none", dt) if shortcut else y
            return F.relu(z + sc)

        y = block(p, "res2a", 1, True)
        y = block(y, "res2b", 1, False)
        f2 = block(y, "res2c", 1, False)
        y = block(f2, "res3a", 2, True)
        for b in "bc":
            y = block(y, "res3" + b, 1, False)
        f3 = block(y, "res3d", 1, False)
        s1, s2, s3 = f1[:, :32], f2[:, :128], f3[:, :128]
        f4 = torch.cat([_cba(f3, w, "conv4_1", 2, True, "leaky", dt),
                        _cba(f3, w, "conv4_2", 2, True, "leaky", dt)], 1)
        taps.update(f1=f1, f2=f2, f3=f3)
    else:
        f = x
        sk = []
        for lvl in (1, 2, 3, 4):
            a = _cba(f, w, "conv%d_1" % lvl, 2, True, "leaky", dt)
            b = _cba(f, w, "conv%d_2" % lvl, 2, True, "leaky", dt)
            sk.append(b)
            f = torch.cat([a, b], 1)
        f4 = f
        s1, s2, s3 = sk[0], sk[1], sk[2]
    taps["f4"] = f4
    n = x.shape[0]
    flat = f4.permute(0, 2, 3, 1).reshape(n, -1)              # Flatten of NHWC
    enc = flat @ _t(w["dense_enc.kernel"], dt) + _t(w["dense_enc.bias"], dt)
    d = (enc @ _t(w["dense_dec.kernel"], dt) + _t(w["dense_dec.bias"], dt)).reshape(n, 8, 8, 256).permute(0, 3, 1, 2)

    def up(y, name):
        return F.leaky_relu(_bn(_deconv(y, w, name, dt), w, name, dt), ALPHA)

    d1 = _cba(torch.cat([up(d, "up1"), s3], 1), w, "deconv1", 1, True, "leaky", dt)
    d2 = _cba(torch.cat([up(d1, "up2"), s2], 1), w, "deconv2", 1, True, "leaky", dt)
    d3 = _cba(torch.cat([up(d2, "up3"), s1], 1), w, "deconv3", 1, True, "leaky", dt)
    dec = torch.tanh(_deconv(d3, w, "head_xyz", dt))
    prob = torch.sigmoid(_deconv(d3, w, "head_prob", dt))
    out = lambda t: t.permute(0, 2, 3, 1).contiguous().numpy()
    return out(dec), out(prob), {k: out(v) for k, v in taps.items()}
