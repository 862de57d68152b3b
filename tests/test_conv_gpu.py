"""The hand-written MFMA 3x3 convolution over 32 channels (csrc/lsi_conv.hip;
the `upcnv1b` and `pred_l` layers of the LDI heads, reference nets.py:104-111,
150-158) against the plain PyTorch fp32 convolution of the same bf16-rounded
operands: the kernel accumulates in fp32, so what separates the two is the
summation order and -- for the bf16 output of the 32 -> 32 layer -- one final
rounding to bf16 (relative 2^-8)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev(built_lib):
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  return torch.device('cuda:0')


def _x(n, h, w, dev, seed):
  g = torch.Generator().manual_seed(seed)
  x = torch.randn((n, 32, h, w), generator=g).to(dev).to(torch.bfloat16)
  return x.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize('shape', [(2, 40, 64), (1, 33, 16), (3, 7, 112), (1, 256, 768)])
@pytest.mark.parametrize('cout', [32, 16])
def test_conv3x3_c32_matches_the_fp32_convolution(shape, cout, dev):
  from lsi.nnutils import _hip_conv
  n, h, w = shape
  x = _x(n, h, w, dev, 1)
  g = torch.Generator().manual_seed(2)
  wt = (torch.randn((cout, 32, 3, 3), generator=g) * 0.1).to(dev)
  assert _hip_conv.supported(x, 32, cout, 3, 1, False)
  got = _hip_conv.conv3x3_c32(x, wt)
  assert got.dtype == torch.bfloat16 and got.shape == (n, cout, h, w)
  assert got.is_contiguous(memory_format=torch.channels_last)
  want = F.conv2d(x.float(), wt.to(torch.bfloat16).float(), None, 1, 1)
  err = (got.float() - want).abs()
  # fp32 accumulate, one rounding to bf16: half an ulp of the result (2^-9
  # relative) plus summation-order noise
  assert float((err - want.abs() * 2.0 ** -8).max()) <= 2e-3, float(err.max())


@pytest.mark.parametrize('shape', [(2, 24, 48), (1, 256, 768)])
def test_prediction_head_bias_sigmoid_rgbd_pixels(shape, dev):
  from lsi.nnutils import _hip_conv
  n, h, w = shape
  x = _x(n, h, w, dev, 3)
  g = torch.Generator().manual_seed(4)
  wt = (torch.randn((4, 32, 3, 3), generator=g) * 0.2).to(dev)
  b = torch.randn((4,), generator=g).to(dev)
  assert _hip_conv.supported(x, 32, 4, 3, 1, True)
  got = _hip_conv.conv3x3_c32_sigmoid(x, wt, b)
  assert got.dtype == torch.float32 and got.shape == (n, 4, h, w)
  # channels innermost: the permuted view is the renderer's RGBD pixel layout
  assert got.permute(0, 2, 3, 1).is_contiguous()
  want = torch.sigmoid(F.conv2d(x.float(), wt.to(torch.bfloat16).float(), b, 1, 1))
  assert float((got - want).abs().max()) <= 2e-5


@pytest.mark.parametrize('shape', [(2, 20, 32), (1, 70, 144)])
def test_gradients_of_both_layers(shape, dev):
  from lsi.nnutils import _hip_conv
  n, h, w = shape
  g = torch.Generator().manual_seed(5)
  for head in (False, True):
    cout = 4 if head else 32
    x = _x(n, h, w, dev, 6).requires_grad_(True)
    wt = (torch.randn((cout, 32, 3, 3), generator=g) * 0.1).to(dev).requires_grad_(True)
    b = torch.randn((cout,), generator=g).to(dev).requires_grad_(True) if head else None
    c = torch.randn((n, cout, h, w), generator=g).to(dev)
    if head:
      y = _hip_conv.conv3x3_c32_sigmoid(x, wt, b)
      (y * c).sum().backward()
    else:
      y = _hip_conv.conv3x3_c32(x, wt)
      (y.float() * c).sum().backward()
    x2 = x.detach().float().requires_grad_(True)
    w2 = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True) if head else None
    z = F.conv2d(x2, w2, b2, 1, 1)
    ((torch.sigmoid(z) if head else z) * c).sum().backward()
    # bf16 gradients (the incoming gradient and the results are rounded to
    # bf16 on the kernel / MIOpen path): 2^-7 of the largest entry
    for got, want in ((x.grad, x2.grad), (wt.grad, w2.grad)) + (((b.grad, b2.grad),) if head else ()):
      scale = float(want.abs().max())
      assert float((got.float() - want).abs().max()) <= 2.0 ** -6 * scale, (head, scale)


def test_unsupported_shapes_stay_on_the_library(dev):
  from lsi.nnutils import _hip_conv
  x = _x(1, 8, 16, dev, 7)
  assert not _hip_conv.supported(x.float(), 32, 32, 3, 1, False)       # fp32
  assert not _hip_conv.supported(x, 32, 32, 5, 1, False)               # 5x5
  assert not _hip_conv.supported(x, 32, 64, 3, 1, False)               # 64 outputs
  assert not _hip_conv.supported(x, 32, 5, 3, 1, True)                 # masks head
  assert not _hip_conv.supported(x[:, :, :, :8], 32, 32, 3, 1, False)  # width 8
  assert not _hip_conv.supported(x.contiguous(), 32, 32, 3, 1, False) or \
      x.contiguous().is_contiguous(memory_format=torch.channels_last)
