"""The plain-C restatement (oracle/conv_ref.c) agrees with the torch-op oracle: two independent CPU statements of the
same arithmetic, so a mistake in how the oracle calls torch (stride, padding, weight layout of ConvTranspose2d, biased
variance) cannot hide."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import patchgan_oracle as O

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')


@pytest.fixture(scope='module')
def cref():
    so = os.path.join(ORACLE_DIR, 'libconv_ref.so')
    if not os.path.exists(so):
        subprocess.run(['make', '-C', ORACLE_DIR], check=True)
    return ctypes.CDLL(so)


def _p(t):
    return t.numpy().ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize('N,Ci,H,W,Co,s', [(2, 3, 16, 16, 5, 2), (1, 4, 9, 11, 3, 1), (2, 6, 8, 8, 1, 1), (1, 2, 15, 13, 4, 2)])
def test_conv_and_wgrad(cref, N, Ci, H, W, Co, s):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g)
    b = torch.randn(Co, generator=g)
    Ho, Wo = (H - 2) // s + 1, (W - 2) // s + 1
    y = torch.empty(N, Co, Ho, Wo)
    cref.conv4x4(_p(x), _p(w), _p(b), _p(y), N, Ci, H, W, Co, s)
    want = F.conv2d(x, w, b, stride=s, padding=1)
    assert want.shape == y.shape
    np.testing.assert_allclose(y.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, stride=s, padding=1).backward(dy)
    dw = torch.empty_like(w)
    cref.conv4x4_wgrad(_p(x), _p(dy), _p(dw), N, Ci, H, W, Co, s)
    np.testing.assert_allclose(dw.numpy(), wr.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('N,Ci,H,W,Co', [(2, 4, 5, 7, 3), (1, 8, 2, 2, 8)])
def test_conv_transpose_and_instnorm(cref, N, Ci, H, W, Co):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, 4, 4, generator=g)
    y = torch.empty(N, Co, 2 * H, 2 * W)
    cref.convT4x4(_p(x), _p(w), _p(y), N, Ci, H, W, Co)
    want = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    np.testing.assert_allclose(y.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    z = y.clone()
    cref.instnorm(_p(z), N, Co, 4 * H * W, ctypes.c_float(1e-5))
    np.testing.assert_allclose(z.numpy(), O.instance_norm(want).numpy(), rtol=1e-4, atol=1e-5)


def test_down_block_end_to_end(cref):
    """One encoder block (Conv -> InstanceNorm -> LeakyReLU) in C equals the oracle's first block (unet.py:8-35)."""
    g = torch.Generator().manual_seed(2)
    x = torch.rand(1, 3, 32, 32, generator=g)
    w = torch.randn(4, 3, 4, 4, generator=g) * 0.2
    y = torch.empty(1, 4, 16, 16)
    cref.conv4x4(_p(x), _p(w), None, _p(y), 1, 3, 32, 32, 4, 2)
    cref.instnorm(_p(y), 1, 4, 256, ctypes.c_float(1e-5))
    y = torch.where(y > 0, y, 0.2 * y)
    want = O.apply_act(O.instance_norm(F.conv2d(x, w, None, stride=2, padding=1)), 'leakyrelu')
    np.testing.assert_allclose(y.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
