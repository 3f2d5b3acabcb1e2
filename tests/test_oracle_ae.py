"""CPU tests pinning oracle/ae_oracle.py (the reference has no tests: parity is unpinned, so
the oracle is checked against (1) hand-computable TF-'SAME' micro cases, (2) an independent
torch formulation of the whole graph)."""
import numpy as np
import pytest
import torch

from oracle import ae_oracle as O
from pix2pose_amd import weights as W
from oracle import ae_torch as torch_ref


def test_same_pad_rules():
    # k=5,s=2 even input -> 1 before/2 after; k=5,s=1 -> 2/2; k=3,s=1 -> 1/1;
    # maxpool 3x3 s2 on 64 -> 0 before / 1 after.
    assert O.same_pad(16, 5, 2) == (8, 1)
    assert O.same_pad(32, 5, 1) == (32, 2)
    assert O.same_pad(32, 3, 1) == (32, 1)
    assert O.same_pad(64, 3, 2) == (32, 0)
    assert O.same_pad(128, 5, 2) == (64, 1)


def test_conv_same_stride2_is_asymmetric():
    # one-hot kernel tap picks x[2*o + i - 1]: asymmetric (1 before) TF padding
    x = np.arange(8 * 8, dtype=np.float32).reshape(1, 8, 8, 1)
    k = np.zeros((5, 5, 1, 1), np.float32)
    k[0, 0, 0, 0] = 1.0
    y = O.conv2d(x, k, np.zeros(1, np.float32), 2, "same")
    assert y.shape == (1, 4, 4, 1)
    exp = np.zeros((4, 4), np.float32)
    for oh in range(4):
        for ow in range(4):
            ih, iw = 2 * oh - 1, 2 * ow - 1
            exp[oh, ow] = x[0, ih, iw, 0] if ih >= 0 and iw >= 0 else 0.0
    np.testing.assert_array_equal(y[0, :, :, 0], exp)


def test_deconv_index_rule():
    # y[o] = sum x[i] w[k], o = 2 i + k - 1, single impulse input
    x = np.zeros((1, 6, 6, 1), np.float32)
    x[0, 2, 3, 0] = 1.0
    k = np.arange(25, dtype=np.float32).reshape(5, 5, 1, 1) + 1
    y = O.conv2d_transpose(x, k, np.zeros(1, np.float32))
    assert y.shape == (1, 12, 12, 1)
    exp = np.zeros((12, 12), np.float32)
    for kh in range(5):
        for kw in range(5):
            oh, ow = 2 * 2 + kh - 1, 2 * 3 + kw - 1
            if 0 <= oh < 12 and 0 <= ow < 12:
                exp[oh, ow] += k[kh, kw, 0, 0]
    np.testing.assert_array_equal(y[0, :, :, 0], exp)


def test_micro_conv_deconv_vs_torch():
    rs = np.random.RandomState(3)
    x = rs.randn(2, 8, 8, 3).astype(np.float32)
    w = {"c.kernel": rs.randn(5, 5, 3, 4).astype(np.float32), "c.bias": rs.randn(4).astype(np.float32)}
    y = O.conv2d(x, w["c.kernel"], w["c.bias"], 2, "same")
    yt = torch_ref._conv(torch.from_numpy(x).double().permute(0, 3, 1, 2), w, "c", 2, True, torch.float64)
    np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).numpy(), rtol=0, atol=2e-6)
    x = rs.randn(1, 6, 6, 3).astype(np.float32)
    w = {"d.kernel": rs.randn(5, 5, 4, 3).astype(np.float32), "d.bias": rs.randn(4).astype(np.float32)}
    y = O.conv2d_transpose(x, w["d.kernel"], w["d.bias"])
    yt = torch_ref._deconv(torch.from_numpy(x).double().permute(0, 3, 1, 2), w, "d", torch.float64)
    assert y.shape == (1, 12, 12, 4)
    np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).numpy(), rtol=0, atol=2e-6)


def test_maxpool_same_ignores_padding():
    x = -np.ones((1, 4, 4, 1), np.float32) * 5
    y = O.maxpool_3x3_s2_same(x)
    assert y.shape == (1, 2, 2, 1)
    np.testing.assert_array_equal(y, -5 * np.ones_like(y))     # zero padding would give 0


@pytest.mark.parametrize("backbone", ["resnet50", "paper"])
def test_forward_vs_independent_torch_formulation(backbone):
    w = W.synthetic_weights(backbone, 1)
    x = (np.random.RandomState(0).randint(0, 256, (2, 128, 128, 3)).astype(np.float32) - 128) / 128
    taps = {}
    d, p = O.forward(w, x, backbone, taps)
    dt, pt, tt = torch_ref.forward(w, x, backbone)
    assert d.shape == (2, 128, 128, 3) and p.shape == (2, 128, 128, 1)
    for k, v in tt.items():
        scale = np.abs(v).max()
        assert np.abs(taps[k] - v).max() <= 2e-6 * scale + 1e-6, k
    assert np.abs(d - dt).max() < 1e-5
    assert np.abs(p - pt).max() < 1e-5
    # sanity: the synthetic net is not saturated / degenerate
    assert 0.2 < np.abs(d).mean() < 0.9


def test_param_counts_match_survey():
    assert W.n_params("resnet50") == 27904452          # 27.90 M
    assert W.n_params("paper") == 25740356             # 25.74 M
