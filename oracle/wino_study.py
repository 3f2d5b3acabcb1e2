"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Error study for a multiply-count reduction of the three 5x5 stride-1 decoder layers (`deconv1/2/3`,
reference pix2pose_model/ae_model.py:207-211,217-220,227-230): Winograd / Cook-Toom minimal filtering
F(m, 5) along the row axis only (1D: m + 4 products per m outputs and vertical tap instead of 5 m) or
along both axes (2D).  The study emulates what a kernel on the f16 matrix pipe would compute -- transforms in
fp32, the 22-bit hi/lo f16 split of both operands AFTER the transform, three products per block, fp32
accumulation, fp32 inverse transform -- inside the torch-CPU formulation of the generator (oracle/ae_torch.py),
every other layer in fp64, and reports the distance of the network outputs from the all-fp64 graph.

    python -m oracle.wino_study [--backbone resnet50] [--n 2] [--out profiles/r06_wino_error_study.json]
"""
from __future__ import annotations

import argparse
import json
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F

from . import ae_torch


# ------------------------------------------------------------------ Cook-Toom matrices, exact rationals
def cook_toom(m: int, r: int, pts):
    """F(m, r) with n - 1 = m + r - 2 finite points `pts` and the point at infinity.
    -> AT [m, n], G [n, r], BT [n, n] as float64, with  y = AT ((G g) * (BT d))  the correlation
    y_i = sum_k g_k d_{i+k}."""
    n = m + r - 1
    pts = [Fraction(p) for p in pts]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1
    AT = [[(p ** i if not (p == 0 and i == 0) else Fraction(1)) for p in pts] + [Fraction(1 if i == m - 1 else 0)] for i in range(m)]
    G = []
    for j, p in enumerate(pts):
        nj = Fraction(1)
        for l, q in enumerate(pts):
            if l != j:
                nj *= (p - q)
        G.append([(p ** k if not (p == 0 and k == 0) else Fraction(1)) / nj for k in range(r)])
    G.append([Fraction(1 if k == r - 1 else 0) for k in range(r)])
    # BT from the correctness condition, solved exactly: for every (i, k, p):
    #   sum_j AT[i][j] G[j][k] BT[j][p] = [p == i + k]
    import sympy as sp
    A_ = sp.Matrix(m * r, n, lambda e, j: sp.Rational(AT[e // r][j] * G[j][e % r]))
    BT = sp.zeros(n, n)
    for p in range(n):
        rhs = sp.Matrix(m * r, 1, lambda e, _: 1 if (e // r + e % r) == p else 0)
        sol = A_.solve_least_squares(rhs) if m * r != n else A_.solve(rhs)
        assert (A_ * sol - rhs).norm() == 0
        BT[:, p] = sol
    to = lambda M: np.array([[float(v) for v in row] for row in M], np.float64)
    return to(AT), to(G), np.array(BT.tolist(), dtype=np.float64)


def balance(AT, G, BT):
    """Per-position power-of-two rescaling: row j of G times s_j, column j of AT divided by s_j, chosen so that the
    largest entry of every G row is in [1, 2) -- keeps the transformed weights in one binade range (exact in fp32)."""
    s = 2.0 ** -np.floor(np.log2(np.abs(G).max(1)))
    return AT / s[None, :], G * s[:, None], BT


# ------------------------------------------------------------------ the split-f16 product, emulated
def split22(x: torch.Tensor):
    """fp32 -> (hi, lo) as fp32 tensors holding f16 values: hi = f16(x) toward zero, lo = f16(x - hi)."""
    h = x.half()
    over = h.float().abs() > x.abs()
    bits = h.view(torch.int16)
    bits = torch.where(over, bits - 1, bits)          # one ulp toward zero (sign bit untouched)
    hi = bits.view(torch.float16).float()
    lo = (x - hi).half().float()
    return hi, lo


def mm3(a: torch.Tensor, b: torch.Tensor, eq: str):
    """einsum with both operands split, three products, fp32 accumulation (a_lo b_hi + a_hi b_lo + a_hi b_hi)."""
    ah, al = split22(a)
    bh, bl = split22(b)
    return torch.einsum(eq, al, bh) + torch.einsum(eq, ah, bl) + torch.einsum(eq, ah, bh)


def prescale_cout(k: torch.Tensor):
    """Per-output-channel power-of-two pre-scale of a [..., Cout] weight panel into [0.5, 1) * 2^11 of its largest
    magnitude, like model.hip's panel packing; returns (scaled weights, inverse scale [Cout])."""
    mx = k.abs().reshape(-1, k.shape[-1]).max(0).values.clamp_min(1e-30)
    s = 2.0 ** (10 - torch.floor(torch.log2(mx)))
    return k * s, 1.0 / s


# ------------------------------------------------------------------ the three formulations of a 5x5 SAME conv, NCHW
def conv_direct(x, k):
    """x [N,C,H,W] fp32, k [5,5,Cin,Cout] fp32 -> [N,Cout,H,W]; split-f16 products, fp32 accumulation."""
    n, c, h, w = x.shape
    xp = F.pad(x, (2, 2, 2, 2))
    cols = xp.unfold(2, 5, 1).unfold(3, 5, 1)                     # [N,C,H,W,5,5]
    ks, inv = prescale_cout(k)
    y = mm3(cols, ks, "nchwyx,yxco->nohw")
    return y * inv.view(1, -1, 1, 1)


def conv_wino1d(x, k, mats):
    AT, G, BT = mats
    m, nn = AT.shape
    n, c, h, w = x.shape
    assert w % m == 0
    t = w // m
    xp = F.pad(x, (2, 2 + (nn - m - 4), 2, 2))
    tiles = xp.unfold(3, nn, m)                                    # [N,C,H+4,T,nn]
    v = torch.einsum("nchtp,jp->nchtj", tiles, torch.from_numpy(BT).float())          # fp32 transform
    rows = v.unfold(2, 5, 1)                                       # [N,C,H,T,nn,5(ky)]
    u = torch.einsum("jx,yxco->jyco", torch.from_numpy(G), k.double())                # offline, fp64 -> fp32
    u = u.float()
    us, inv = prescale_cout(u)
    mm = mm3(rows, us, "nchtjy,jyco->nohtj")
    y = torch.einsum("nohtj,ij->nohti", mm, torch.from_numpy(AT).float())
    return y.reshape(n, -1, h, w) * inv.view(1, -1, 1, 1)


def conv_wino2d(x, k, mats):
    AT, G, BT = mats
    m, nn = AT.shape
    n, c, h, w = x.shape
    assert w % m == 0 and h % m == 0
    ty, tx = h // m, w // m
    xp = F.pad(x, (2, 2, 2, 2))
    tiles = xp.unfold(2, nn, m).unfold(3, nn, m)                   # [N,C,Ty,Tx,nn,nn]
    bt = torch.from_numpy(BT).float()
    v = torch.einsum("ncabpq,ip,jq->ncabij", tiles, bt, bt)
    g = torch.from_numpy(G)
    u = torch.einsum("iy,jx,yxco->ijco", g, g, k.double()).float()
    us, inv = prescale_cout(u)
    mm = mm3(v, us, "ncabij,ijco->noabij")
    at = torch.from_numpy(AT).float()
    y = torch.einsum("noabij,yi,xj->noaybx", mm, at, at)
    return y.reshape(n, -1, h, w) * inv.view(1, -1, 1, 1)


VARIANTS = {
    "direct": None,
    "1d_F2_pm1_pm2": ("1d", 2, [0, 1, -1, 2, -2]),
    "1d_F2_pm1_pmhalf": ("1d", 2, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)]),
    "1d_F4_pm1_pm2_pmhalf": ("1d", 4, [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2)]),
    "1d_F4_pm1_pmhalf_pm2_bal": ("1d", 4, [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2)], "bal"),
    "1d_F4_pm1_pmhalf_pm3half": ("1d", 4, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 2), Fraction(-3, 2)]),
    "1d_F4_pmhalf_pm1_pm3quarter": ("1d", 4, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 4), Fraction(-3, 4)]),
    "2d_F2_pm1_pm2": ("2d", 2, [0, 1, -1, 2, -2]),
    "2d_F2_pm1_pmhalf": ("2d", 2, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)]),
    "2d_F4_pm1_pm2_pmhalf": ("2d", 4, [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2)]),
}


def make_conv(variant):
    spec = VARIANTS[variant]
    if spec is None:
        return conv_direct
    mats = cook_toom(spec[1], 5, spec[2])
    if len(spec) > 3:
        mats = balance(*mats)
    fn = conv_wino1d if spec[0] == "1d" else conv_wino2d
    return lambda x, k: fn(x, k, mats)


def forward_with(w, x, backbone, conv5):
    """ae_torch.forward in fp64 with the three 5x5 stride-1 layers computed by `conv5` on fp32 operands."""
    orig = ae_torch._conv
    layer_err = {}

    def patched(xx, ww, name, stride, same, dt):
        if name in ("deconv1", "deconv2", "deconv3"):
            ref = orig(xx, ww, name, stride, same, dt)
            y = conv5(xx.float(), torch.from_numpy(ww[name + ".kernel"]).float())
            y = y.double() + torch.from_numpy(ww[name + ".bias"]).double().view(1, -1, 1, 1)
            layer_err[name] = (float((y - ref).abs().max()), float(ref.abs().max()), float((y - ref).pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt()))
            return y
        return orig(xx, ww, name, stride, same, dt)

    ae_torch._conv = patched
    try:
        dec, prob, _ = ae_torch.forward(w, x, backbone, torch.float64)
    finally:
        ae_torch._conv = orig
    return dec, prob, layer_err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--out", default="")
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pix2pose_amd import weights as W
    torch.set_num_threads(8)
    rs = np.random.RandomState(0)
    x = ((rs.randint(0, 256, (a.n, 128, 128, 3)).astype(np.float32)) - 128) / 128
    names = [v for v in (a.variants.split(",") if a.variants else VARIANTS)]
    report = {}
    for fam, w in (("synthetic", W.synthetic_weights(a.backbone, 3)), ("trained_like", W.trained_like_weights(a.backbone, 5))):
        d0, p0, _ = ae_torch.forward(w, x, a.backbone, torch.float64)
        for v in names:
            dec, prob, le = forward_with(w, x, a.backbone, make_conv(v))
            e = max(float(np.abs(dec - d0).max()), float(np.abs(prob - p0).max()))
            report["%s/%s" % (fam, v)] = {"net_out_max_abs_err": e, "layers": {k: {"max_abs": t[0], "ref_max": t[1], "rms": t[2], "ref_rms": t[3]} for k, t in le.items()}}
            print("%-13s %-30s net |d|max %.3e   " % (fam, v, e) + "  ".join("%s %.2e (rel rms %.1e)" % (k, t[0], t[2] / t[3]) for k, t in le.items()), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(report, f, indent=1)


# ------------------------------------------------------------------ round 6, second study: the stride-2 transposed convolutions
# (`up1/2/3`, reference pix2pose_model/ae_model.py:201-204,212-215,222-225) as four sub-pixel phases, each a (2|3) x (2|3)-tap correlation on
# the INPUT grid; along the row axis the 3-tap (or zero-extended 2-tap) filter in F(4, 3) form: 6 products per 4 outputs and vertical
# tap instead of 12 (8).  `python -m oracle.wino_study --up` reports the network error with deconv1/2/3 in their shipped F(4,5) form and
# the three transposed layers in F(4,3) form on top.
def deconv_phases_wino(x, k, mats):
    """x [N,Cin,H,W] fp32, k (kh,kw,Cout,Cin) fp32 -> [N,Cout,2H,2W] (no bias).  mats = cook_toom(4, 3, ...) or None = direct phases."""
    n, c, h, w = x.shape
    cout = k.shape[2]
    out = torch.zeros(n, cout, 2 * h, 2 * w, dtype=torch.float32)
    for py in (0, 1):
        for px in (0, 1):
            g = torch.zeros(3, 3, c, cout, dtype=torch.float64)       # [dy + 1][dx + 1][ci][co]
            for dy in (-1, 0, 1):
                kh = py + 1 - 2 * dy
                if not 0 <= kh < 5:
                    continue
                for dx in (-1, 0, 1):
                    kw = px + 1 - 2 * dx
                    if 0 <= kw < 5:
                        g[dy + 1, dx + 1] = k[kh, kw].double().t()
            xp = F.pad(x, (1, 1, 1, 1))
            if mats is None:
                cols = xp.unfold(2, 3, 1).unfold(3, 3, 1)            # [N,C,H,W,3,3]
                gs, inv = prescale_cout(g.float())
                y = mm3(cols, gs, "nchwyx,yxco->nohw") * inv.view(1, -1, 1, 1)
            else:
                AT, G, BT = mats
                tiles = xp.unfold(3, 6, 4)                             # [N,C,H+2,T,6]
                v = torch.einsum("nchtp,jp->nchtj", tiles, torch.from_numpy(BT).float())
                rows = v.unfold(2, 3, 1)                               # [N,C,H,T,6,3(dy)]
                u = torch.einsum("jx,yxco->jyco", torch.from_numpy(G), g).float()
                us, inv = prescale_cout(u)
                mm = mm3(rows, us, "nchtjy,jyco->nohtj")
                y = torch.einsum("nohtj,ij->nohti", mm, torch.from_numpy(AT).float()).reshape(n, cout, h, w) * inv.view(1, -1, 1, 1)
            out[:, :, py::2, px::2] = y
    return out


def main_up():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pix2pose_amd import weights as W
    torch.set_num_threads(8)
    rs = np.random.RandomState(0)
    x = ((rs.randint(0, 256, (2, 128, 128, 3)).astype(np.float32)) - 128) / 128
    conv5 = make_conv("1d_F4_pm1_pm2_pmhalf")
    f43 = cook_toom(4, 3, [0, 1, -1, 2, -2])
    report = {}
    for backbone in ("resnet50", "paper"):
        for fam, w in (("synthetic", W.synthetic_weights(backbone, 3)), ("trained_like", W.trained_like_weights(backbone, 5))):
            d0, p0, _ = ae_torch.forward(w, x, backbone, torch.float64)
            for label, ups, mats in (("deconv F(4,5) only", (), None), ("+ up2/up3 direct phases (split-f16)", ("up2", "up3"), None),
                                     ("+ up2/up3 F(4,3)", ("up2", "up3"), f43), ("+ up1/up2/up3 F(4,3)", ("up1", "up2", "up3"), f43)):
                orig_c, orig_d = ae_torch._conv, ae_torch._deconv
                layer_err = {}

                def pc(xx, ww, name, stride, same, dt):
                    if name in ("deconv1", "deconv2", "deconv3"):
                        y = conv5(xx.float(), torch.from_numpy(ww[name + ".kernel"]).float())
                        return y.double() + torch.from_numpy(ww[name + ".bias"]).double().view(1, -1, 1, 1)
                    return orig_c(xx, ww, name, stride, same, dt)

                def pd(xx, ww, name, dt):
                    if name in ups:
                        ref = orig_d(xx, ww, name, dt)
                        y = deconv_phases_wino(xx.float(), torch.from_numpy(ww[name + ".kernel"]).float(), mats)
                        y = y.double() + torch.from_numpy(ww[name + ".bias"]).double().view(1, -1, 1, 1)
                        layer_err[name] = float((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
                        return y
                    return orig_d(xx, ww, name, dt)

                ae_torch._conv, ae_torch._deconv = pc, pd
                try:
                    dec, prob, _ = ae_torch.forward(w, x, backbone, torch.float64)
                finally:
                    ae_torch._conv, ae_torch._deconv = orig_c, orig_d
                e = max(float(np.abs(dec - d0).max()), float(np.abs(prob - p0).max()))
                report["%s/%s/%s" % (backbone, fam, label)] = {"net_out_max_abs_err": e, "layer_rel_rms": layer_err}
                print("%-9s %-13s %-40s net |d|max %.3e  %s" % (backbone, fam, label, e, "  ".join("%s %.1e" % kv for kv in layer_err.items())), flush=True)
    return report


if __name__ == "__main__":
    import sys
    if "--up" in sys.argv:
        rep = main_up()
        with open("profiles/r06_wino_up_error_study.json", "w") as f:
            json.dump(rep, f, indent=1)
    else:
        main()
