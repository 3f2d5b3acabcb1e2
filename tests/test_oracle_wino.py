"""The algebra behind the Winograd forms of the 5x5 layers (csrc/wino.hip, wino3.hip, wino3o.hip), checked on the CPU in float64 where the
only error left is rounding: Cook-Toom matrices from oracle/wino_study.py (exact rationals) and the decompositions the kernels rely on --

  * F(4,5) / F(4,3) along the row axis reproduce the direct correlation (reference layers pix2pose_model/ae_model.py:207-211 etc.);
  * a Conv2DTranspose 5x5 / 2 'SAME' (ae_model.py:201-204,212-215,222-225) equals its four sub-pixel phases, each a (2|3) x (2|3)-tap
    correlation on the INPUT grid whose row filter is a (zero-extended) 3-tap F(4,3) filter -- oracle.wino_study.deconv_phases_wino;
  * a Conv2D 5x5 / 2 'SAME' (ae_model.py:190-195) equals the sum of four such correlations on the input's parity planes, with the kernel
    indices model.hip: pack_wino3_s2 uses (odd planes: kh = 2 ky, even planes: kh = 2 ky + 1; columns likewise).

The matrices hard-coded in the kernels / packers (BT in the input transforms, AT in the epilogues, G in model.hip) are these."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ae_torch
from oracle import wino_study as W


def test_cook_toom_matrices_are_the_kernels_constants():
    AT, G, BT = W.cook_toom(4, 3, [0, 1, -1, 2, -2])
    np.testing.assert_array_equal(BT, np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                                                [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64))      # wino3_input_kernel
    np.testing.assert_array_equal(AT, np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64))      # epilogues
    np.testing.assert_allclose(G * 24, np.array([[6, 0, 0], [-4, -4, -4], [-4, 4, -4], [1, 2, 4], [1, -2, 4], [0, 0, 24]], np.float64), rtol=0, atol=1e-12)      # kWino3G
    AT5, G5, BT5 = W.cook_toom(4, 5, [0, 1, -1, 2, -2, 0.5, -0.5])
    assert AT5.shape == (4, 8) and G5.shape == (8, 5) and BT5.shape == (8, 8)
    np.testing.assert_allclose(G5[0] * 1.0, [-1, 0, 0, 0, 0], atol=1e-12)              # kWinoG row 0
    np.testing.assert_allclose(G5[3] * 90, [1, 2, 4, 8, 16], atol=1e-9)                # kWinoG row 3


@pytest.mark.parametrize("m,r,pts", [(4, 3, [0, 1, -1, 2, -2]), (4, 5, [0, 1, -1, 2, -2, 0.5, -0.5])])
def test_minimal_filtering_reproduces_the_correlation(m, r, pts):
    AT, G, BT = W.cook_toom(m, r, pts)
    rs = np.random.RandomState(1)
    d = rs.randn(m + r - 1)
    g = rs.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[k] * d[i + k] for k in range(r)) for i in range(m)])
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12)


def test_transposed_convolution_is_four_f43_phases():
    rs = np.random.RandomState(2)
    x = torch.from_numpy(rs.randn(2, 16, 8, 8))
    k = torch.from_numpy(rs.randn(5, 5, 32, 16))                        # (kh, kw, Cout, Cin): the Keras Conv2DTranspose kernel
    w = {"t.kernel": k.numpy(), "t.bias": np.zeros(32)}
    ref = ae_torch._deconv(x, w, "t", torch.float64)
    # float64 restatements of the study's float32 emulation: same index algebra, exact arithmetic
    AT, G, BT = W.cook_toom(4, 3, [0, 1, -1, 2, -2])
    n, c, h, wd = x.shape
    out = torch.zeros(n, 32, 2 * h, 2 * wd, dtype=torch.float64)
    for py in (0, 1):
        for px in (0, 1):
            g = torch.zeros(3, 3, c, 32, dtype=torch.float64)
            for dy in (-1, 0, 1):
                kh = py + 1 - 2 * dy
                for dx in (-1, 0, 1):
                    kw = px + 1 - 2 * dx
                    if 0 <= kh < 5 and 0 <= kw < 5:
                        g[dy + 1, dx + 1] = k[kh, kw].t()
            xp = F.pad(x, (1, 1, 1, 1))
            tiles = xp.unfold(3, 6, 4)
            v = torch.einsum("nchtp,jp->nchtj", tiles, torch.from_numpy(BT))
            rows = v.unfold(2, 3, 1)
            u = torch.einsum("jx,yxco->jyco", torch.from_numpy(G), g)
            mm = torch.einsum("nchtjy,jyco->nohtj", rows, u)
            out[:, :, py::2, px::2] = torch.einsum("nohtj,ij->nohti", mm, torch.from_numpy(AT)).reshape(n, 32, h, wd)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=1e-10)
    # and the float32 / split-f16 emulation of the study stays within its own error bar of it
    emu = W.deconv_phases_wino(x.float(), k.float(), (AT, G, BT)).double()
    assert float((emu - ref).abs().max()) < 1e-3 * float(ref.abs().max())


def test_stride2_convolution_is_four_parity_plane_correlations():
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.randn(2, 8, 16, 16))
    k = rs.randn(5, 5, 8, 12)                                           # (kh, kw, Cin, Cout): the Keras Conv2D kernel
    w = {"c.kernel": k, "c.bias": np.zeros(12)}
    ref = ae_torch._conv(x, w, "c", 2, True, torch.float64)             # 'SAME', stride 2: 16x16 -> 8x8
    AT, G, BT = W.cook_toom(4, 3, [0, 1, -1, 2, -2])
    kt = torch.from_numpy(k)
    out = torch.zeros(2, 12, 8, 8, dtype=torch.float64)
    for a in (0, 1):
        for b in (0, 1):
            plane = F.pad(x[:, :, a::2, b::2], (1, 1, 1, 1))            # P(a, b)[r][s] = x[2r + a][2s + b], rows / columns -1 .. 8
            g = torch.zeros(3, 3, 8, 12, dtype=torch.float64)          # [dy + 1][dx + 1]
            for dy in (-1, 0, 1):
                kh = 2 * dy + 2 if a else 2 * dy + 1                     # odd planes: kh = 2 ky with ky = dy + 1; even planes: kh = 2 ky + 1 with ky = dy
                for dx in (-1, 0, 1):
                    kw = 2 * dx + 2 if b else 2 * dx + 1
                    if 0 <= kh < 5 and 0 <= kw < 5:
                        g[dy + 1, dx + 1] = kt[kh, kw]
            tiles = plane.unfold(3, 6, 4)                                # [N, C, 10, 2, 6]
            v = torch.einsum("nchtp,jp->nchtj", tiles, torch.from_numpy(BT))
            rows = v.unfold(2, 3, 1)                                     # [N, C, 8, 2, 6, 3]
            u = torch.einsum("jx,yxco->jyco", torch.from_numpy(G), g)
            mm = torch.einsum("nchtjy,jyco->nohtj", rows, u)
            out += torch.einsum("nohtj,ij->nohti", mm, torch.from_numpy(AT)).reshape(2, 12, 8, 8)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=1e-10)
