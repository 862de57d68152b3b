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


def _cl(t):
  return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize('shape', [
    (2, 20, 48, 32, 32),     # n, h, w, cin, cout
    (1, 37, 100, 64, 64),    # widths that are not multiples of the 64-pixel strip / 32-pixel chunk
    (1, 33, 70, 96, 64),     # three input-channel blocks, rows past one 32-row block
    (2, 8, 16, 32, 96),      # output-channel blocks of 32
    (1, 5, 7, 64, 128),      # output-channel blocks of 64, an image smaller than a chunk
    (2, 256, 768, 32, 32),   # the training step's full-resolution layer (two images)
    (2, 128, 384, 96, 64),   # ... and its half-resolution layer
])
def test_weight_gradient_kernel_matches_the_fp32_reference(shape, dev):
  """lsi_conv3x3_wgrad (MFMA, K = pixels, operands through the LDS transpose
  read) against torch's fp32 weight gradient of the same bf16 operands: exact
  products, fp32 sums in another order (1e-5 of the largest entry)."""
  from lsi import _C
  n, h, w, cin, cout = shape
  g = torch.Generator().manual_seed(11)
  x = _cl(torch.randn((n, cin, h, w), generator=g).to(dev))
  gy = _cl(torch.randn((n, cout, h, w), generator=g).to(dev))
  lib = _C.lib()
  nbytes = lib.lsi_conv3x3_wgrad_workspace_bytes(n, h, w, cin, cout)
  ws = torch.empty((nbytes // 4,), device=dev)
  gw = torch.full((cout, cin, 3, 3), float('nan'), device=dev)
  rc = lib.lsi_conv3x3_wgrad(n, h, w, cin, cout, _C.ptr(x), _C.ptr(gy), _C.ptr(gw),
                             _C.ptr(ws), nbytes, _C.stream_ptr(dev))
  assert rc == 0
  want = torch.nn.grad.conv2d_weight(x.float(), (cout, cin, 3, 3), gy.float(), padding=1)
  scale = float(want.abs().max())
  assert float((gw - want).abs().max()) <= 1e-5 * scale
  # a workspace one byte short, unsupported channel counts
  assert lib.lsi_conv3x3_wgrad(n, h, w, cin, cout, _C.ptr(x), _C.ptr(gy), _C.ptr(gw),
                               _C.ptr(ws), nbytes - 1, _C.stream_ptr(dev)) == -3  # LSI_EWORKSPACE
  assert lib.lsi_conv3x3_wgrad(n, h, w, 48, cout, _C.ptr(x), _C.ptr(gy), _C.ptr(gw),
                               _C.ptr(ws), nbytes, _C.stream_ptr(dev)) == -5  # LSI_EUNSUPPORTED


def test_library_convolution_with_the_own_weight_gradient(dev, monkeypatch):
  """conv3x3_lib_own_wgrad (forward and data gradient on MIOpen, weight
  gradient on the kernel) against fp32 autograd of the same operands."""
  from lsi.nnutils import _hip_conv
  monkeypatch.setattr(_hip_conv, 'WGRAD_MIN_PIXELS', 0)
  g = torch.Generator().manual_seed(12)
  n, h, w, cin, cout = 2, 24, 40, 96, 64
  x = _cl(torch.randn((n, cin, h, w), generator=g).to(dev)).requires_grad_(True)
  wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(dev).requires_grad_(True)
  c = torch.randn((n, cout, h, w), generator=g).to(dev)
  assert _hip_conv.wgrad_supported(x, cin, cout, 3, 1)
  y = _hip_conv.conv3x3_lib_own_wgrad(x, wt)
  (y.float() * c).sum().backward()
  x2 = x.detach().float().requires_grad_(True)
  w2 = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
  z = F.conv2d(x2, w2, None, 1, 1)
  (z * c).sum().backward()
  assert float((y.float() - z).abs().max()) <= 2.0 ** -7 * float(z.abs().max())
  for got, want in ((x.grad, x2.grad), (wt.grad, w2.grad)):
    assert float((got.float() - want).abs().max()) <= 2.0 ** -6 * float(want.abs().max())
  # below the size threshold the layer stays on the library
  monkeypatch.setattr(_hip_conv, 'WGRAD_MIN_PIXELS', 10 ** 9)
  assert not _hip_conv.wgrad_supported(x, cin, cout, 3, 1)


# ---- the implicit-GEMM kernel (csrc/lsi_conv_igemm.hip): every other layer ----------

def _same_pads(size, k, s):
  """TF `SAME`: (before, after, out)."""
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return total // 2, total - total // 2, out


def _clast(t):
  return t.contiguous(memory_format=torch.channels_last)


IGEMM_CASES = [
    # n, cin, cout, h, w, k, stride
    (2, 96, 64, 40, 48, 3, 1),      # upcnv2b
    (1, 192, 128, 20, 36, 3, 1),    # upcnv3b
    (2, 32, 32, 33, 50, 3, 1),      # ragged tile edges
    (1, 32, 32, 30, 40, 7, 1),      # cnv1b
    (1, 64, 64, 18, 24, 5, 1),      # cnv2b
    (2, 64, 128, 32, 48, 3, 2),     # cnv3 (TF SAME: 0 before, 1 after)
    (1, 32, 64, 24, 40, 5, 2),      # cnv2 (1 / 2)
    (2, 512, 512, 4, 12, 3, 1),     # cnv6b: a map smaller than the tile
    (1, 256, 96, 9, 7, 3, 2),       # odd sizes, 96 output channels (BN = 32)
    (8, 1024, 512, 4, 12, 3, 1),    # icnv7
]


@pytest.fixture
def own_wgrad_everywhere(monkeypatch):
  """lsi_conv2d_wgrad also below the size from which the network uses it."""
  from lsi.nnutils import _hip_conv
  monkeypatch.setattr(_hip_conv, 'IGEMM_WGRAD_MIN_PIXELS', 0)
  monkeypatch.setattr(_hip_conv, 'WGRAD_MIN_PIXELS', 1 << 30)   # (not the row-ring kernel)
  monkeypatch.setattr(_hip_conv, '_WGRAD_BYTES', {})


@pytest.mark.parametrize('case', IGEMM_CASES)
def test_igemm_convolution_forward_and_gradients(case, dev, own_wgrad_everywhere):
  """slim.conv2d with TF SAME padding (reference nets.py:244-348): forward,
  data gradient and weight gradient against fp32 autograd of the same
  bf16-rounded operands."""
  from lsi.nnutils import _hip_conv
  n, cin, cout, h, w, k, s = case
  g = torch.Generator().manual_seed(11)
  x = _clast(torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  wt = (torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev).requires_grad_(True)
  pt, pb, oh = _same_pads(h, k, s)
  pl, pr, ow = _same_pads(w, k, s)
  assert _hip_conv.igemm_supported(x, cin, cout, k, s)
  got = _hip_conv.conv2d(x, wt, s, pt, pl, oh, ow)
  assert got.dtype == torch.bfloat16 and got.shape == (n, cout, oh, ow)
  assert got.is_contiguous(memory_format=torch.channels_last)
  xr = x.detach().float().requires_grad_(True)
  wr = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
  want = F.conv2d(F.pad(xr, (pl, pr, pt, pb)), wr, None, s)
  err = (got.float() - want).abs()
  assert float((err - want.abs() * 2.0 ** -8).max()) <= 2e-3, float(err.max())
  c = torch.randn(want.shape, generator=g).to(dev).to(torch.bfloat16)
  (got.float() * c.float()).sum().backward()
  (want * c.float()).sum().backward()
  gx, gxr = x.grad.float(), xr.grad
  tol = float(gxr.abs().max()) * 2.0 ** -7 + 1e-6
  assert float((gx - gxr).abs().max()) <= tol, (float((gx - gxr).abs().max()), tol)
  gw, gwr = wt.grad, wr.grad
  # (c is bf16: the products are exact, the sums fp32 in another order -- where
  # lsi_conv2d_wgrad takes the shape; the library returns a bf16-rounded result)
  own = _hip_conv._igemm_wgrad_bytes(_hip_conv._conv_desc(n, h, w, cin, oh, ow, cout, k, k, s, pt, pl)) > 0
  tolw = float(gwr.abs().max()) * (1e-4 if own else 2e-2) + 1e-6
  assert float((gw - gwr).abs().max()) <= tolw, (float((gw - gwr).abs().max()), tolw)


@pytest.mark.parametrize('case', [(2, 128, 64, 16, 24), (1, 64, 32, 33, 20), (8, 512, 512, 2, 6),
                                  (1, 128, 128, 64, 192)])
def test_igemm_transposed_convolution(case, dev, own_wgrad_everywhere):
  """slim.conv2d_transpose 4 x 4 stride 2 (reference nets.py:100-103) = torch
  ConvTranspose2d(k = 4, stride 2, padding 1): four parity classes of 2 x 2 taps."""
  from lsi.nnutils import _hip_conv
  n, cin, cout, h, w = case
  g = torch.Generator().manual_seed(12)
  x = _clast(torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  wt = (torch.randn((cin, cout, 4, 4), generator=g) * (2.0 / (cin * 4)) ** 0.5).to(dev).requires_grad_(True)
  assert _hip_conv.convt_supported(x, cin, cout, 4, 2)
  got = _hip_conv.conv_transpose2d(x, wt)
  assert got.shape == (n, cout, 2 * h, 2 * w) and got.dtype == torch.bfloat16
  xr = x.detach().float().requires_grad_(True)
  wr = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
  want = F.conv_transpose2d(xr, wr, None, 2, 1)
  err = (got.float() - want).abs()
  assert float((err - want.abs() * 2.0 ** -8).max()) <= 2e-3, float(err.max())
  c = torch.randn(want.shape, generator=g).to(dev).to(torch.bfloat16)
  (got.float() * c.float()).sum().backward()
  (want * c.float()).sum().backward()
  gx, gxr = x.grad.float(), xr.grad
  tol = float(gxr.abs().max()) * 2.0 ** -7 + 1e-6
  assert float((gx - gxr).abs().max()) <= tol, (float((gx - gxr).abs().max()), tol)
  gw, gwr = wt.grad, wr.grad
  tolw = float(gwr.abs().max()) * 2e-2 + 1e-5
  assert float((gw - gwr).abs().max()) <= tolw, (float((gw - gwr).abs().max()), tolw)


# (n, h, w, groups, input dtype, parameter layout): full width of a KITTI image,
# odd sizes (TF SAME pads 3 / 3 there, 2 / 3 for even ones; partial tiles), a map
# smaller than one tile
@pytest.mark.parametrize('case', [(2, 64, 256, 2, 'f32', 'contig'), (4, 37, 91, 2, 'f32', 'clast'),
                                  (1, 9, 11, 1, 'bf16', 'contig'), (3, 130, 70, 3, 'bf16', 'clast'),
                                  (8, 256, 768, 2, 'f32', 'clast')])
def test_first_convolution_forward_statistics_and_weight_gradient(case, dev):
  """`cnv1` (reference nets.py:273: slim.conv2d(inp_img, 32, [7, 7], stride=2)) on
  lsi_conv2d_first_fwd / _wgrad: the fp32 (or bf16) image read in place, rounded
  to bf16 as autocast rounds it, against fp32 F.conv2d of the same rounded
  operands; the batch-norm sums left by the epilogue against fp64 moments; the
  weight gradient (K = all output pixels, fp32 sums) against fp32 autograd."""
  from lsi.nnutils import _hip_bn, _hip_conv
  n, h, w, groups, xdt, layout = case
  g = torch.Generator().manual_seed(17)
  img = torch.rand((n, h, w, 3), generator=g).to(dev)          # N x H x W x 3, like the loader's
  if xdt == 'bf16':
    img = img.to(torch.bfloat16)
  x = img.permute(0, 3, 1, 2)                                  # the module's NCHW view
  wt = (torch.randn((32, 3, 7, 7), generator=g) * (2.0 / 147) ** 0.5).to(dev)
  if layout == 'clast':
    wt = wt.contiguous(memory_format=torch.channels_last)
  wt.requires_grad_(True)
  pt, pb, oh = _same_pads(h, 7, 2)
  pl, pr, ow = _same_pads(w, 7, 2)
  assert _hip_conv.first_supported(x, 3, 32, 7, 2)
  got = _hip_conv.conv2d_first(x, wt, 2, pt, pl, oh, ow)
  assert got.dtype == torch.bfloat16 and got.shape == (n, 32, oh, ow)
  assert got.is_contiguous(memory_format=torch.channels_last)
  xr = x.detach().to(torch.bfloat16).float()
  wr = wt.detach().to(torch.bfloat16).float().contiguous().requires_grad_(True)
  want = F.conv2d(F.pad(xr, (pl, pr, pt, pb)), wr, None, 2)
  err = (got.float() - want).abs()
  assert float((err - want.abs() * 2.0 ** -8).max()) <= 2e-3, float(err.max())
  # weight gradient
  c = torch.randn(want.shape, generator=g).to(dev).to(torch.bfloat16)
  (got.float() * c.float()).sum().backward()
  (want * c.float()).sum().backward()
  gw, gwr = wt.grad, wr.grad
  assert gw.stride() == wt.stride() and gw.dtype == torch.float32
  tolw = float(gwr.abs().max()) * 1e-4 + 1e-6
  assert float((gw - gwr).abs().max()) <= tolw, (float((gw - gwr).abs().max()), tolw)
  # the same call twice: the fold is in a fixed order
  wt.grad = None
  (_hip_conv.conv2d_first(x, wt, 2, pt, pl, oh, ow).float() * c.float()).sum().backward()
  assert torch.equal(wt.grad, gw)
  # statistics from the epilogue + lsi_bn_relu_norm == the two-pass batch norm
  beta = (torch.randn((32,), generator=g) * 0.3).to(dev)
  y1 = _hip_conv.conv2d_first(x, wt, 2, pt, pl, oh, ow, groups)
  assert torch.equal(y1, got)
  z1 = _hip_bn.batch_norm_relu(y1, beta.clone().requires_grad_(True), 1e-3, True, groups, True)
  mr = z1.grad_fn.saved_tensors[2]
  yf = got.detach().double().view(groups, n // groups, 32, oh, ow)
  np.testing.assert_allclose(mr[:, 0].double().cpu().numpy(), yf.mean(dim=(1, 3, 4)).cpu().numpy(),
                             rtol=1e-5, atol=1e-5 * float(yf.abs().max()))
  np.testing.assert_allclose(mr[:, 1].double().cpu().numpy(),
                             torch.rsqrt(yf.var(dim=(1, 3, 4), unbiased=False) + 1e-3).cpu().numpy(),
                             rtol=2e-5)
  z0 = _hip_bn.batch_norm_relu(got, beta, 1e-3, True, groups)
  assert float((z1.detach().float() - z0.float()).abs().max()) <= \
      2.0 ** -7 * float(z0.float().abs().max())


def test_first_layer_module_runs_the_own_kernel_under_autocast(dev, monkeypatch):
  """nets.SlimConv2d(3, 32, 7, 2) -- `cnv1` -- under bf16 autocast with the fp32
  image: no library convolution is called (aten.convolution is made to raise),
  and the result follows the library path's."""
  from lsi.nnutils import nets
  torch.manual_seed(3)
  layer = nets.SlimConv2d(3, 32, 7, 2).to(dev).to(memory_format=torch.channels_last)
  g = torch.Generator().manual_seed(4)
  img = torch.rand((4, 64, 128, 3), generator=g).to(dev)
  x = img.permute(0, 3, 1, 2)
  with nets.bn_groups(2), torch.autocast('cuda', dtype=torch.bfloat16):
    monkeypatch.setattr(nets, 'IGEMM_CONV', False)
    ref = layer(x)
    gref, = torch.autograd.grad(ref.float().square().sum(), layer.conv.weight)
    monkeypatch.setattr(nets, 'IGEMM_CONV', True)

    def boom(*a, **k):
      raise AssertionError('the library convolution was called for cnv1')
    monkeypatch.setattr(F, 'conv2d', boom)
    monkeypatch.setattr(layer.conv, 'forward', boom)
    own = layer(x)
    gown, = torch.autograd.grad(own.float().square().sum(), layer.conv.weight)
  assert own.dtype == torch.bfloat16 and own.shape == ref.shape
  assert float((own.float() - ref.float()).abs().max()) <= 2.0 ** -6 * float(ref.float().abs().max())
  assert float((gown - gref).abs().max()) <= 2e-2 * float(gref.abs().max())


def test_igemm_refuses_what_it_does_not_take(dev):
  import ctypes
  from lsi import _C
  d = _C.LsiConvDesc()
  d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout = 1, 8, 8, 3, 8, 8, 32
  d.KH = d.KW = 3; d.stride = 1; d.pad_t = d.pad_l = 1
  assert _C.lib().lsi_conv2d_supported(ctypes.byref(d)) == 0
  assert _C.lib().lsi_conv2d_packed_bytes(ctypes.byref(d)) == 0
  buf = torch.zeros(4096, device=dev)
  rc = _C.lib().lsi_conv2d_fwd(ctypes.byref(d), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(),
                               None)
  assert rc == -5   # LSI_EUNSUPPORTED
  rc = _C.lib().lsi_conv2d_pack(ctypes.byref(d), 0, buf.data_ptr(), buf.data_ptr(), 16384, None)
  assert rc == -5


def test_packed_weights_follow_the_parameter(dev):
  """The packed form is remembered per parameter and version: an in-place
  update (the optimiser's step) must be seen by the next call."""
  from lsi.nnutils import _hip_conv
  g = torch.Generator().manual_seed(13)
  x = _clast(torch.randn((1, 32, 16, 16), generator=g).to(dev).to(torch.bfloat16))
  wt = (torch.randn((32, 32, 3, 3), generator=g) * 0.1).to(dev)
  a = _hip_conv.conv2d(x, wt, 1, 1, 1, 16, 16).float()
  b = _hip_conv.conv2d(x, wt, 1, 1, 1, 16, 16).float()
  assert torch.equal(a, b)
  with torch.no_grad():
    wt.mul_(2.0)
  c = _hip_conv.conv2d(x, wt, 1, 1, 1, 16, 16).float()
  want = F.conv2d(x.float(), wt.to(torch.bfloat16).float(), None, 1, 1)
  assert float((c - want).abs().max()) <= 2e-3 + float(want.abs().max()) * 2.0 ** -8
  assert float((c - 2 * a).abs().max()) <= float(c.abs().max()) * 2.0 ** -6


def test_repack_all_refreshes_every_layer_with_one_launch(dev):
  """lsi_conv2d_pack_many: after an in-place update of the parameters (the
  optimiser's step) one call brings every packed form up to date."""
  from lsi.nnutils import _hip_conv
  g = torch.Generator().manual_seed(14)
  x = _clast(torch.randn((1, 64, 12, 16), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  # (frozen parameters: their packs are trusted while the version counter stands;
  # a trainable parameter that no optimiser owns is packed on every forward call --
  # test_any_optimizer_loop_keeps_the_packs_fresh)
  w1 = (torch.randn((32, 64, 3, 3), generator=g) * 0.1).to(dev)
  w2 = (torch.randn((64, 32, 4, 4), generator=g) * 0.1).to(dev)
  def run():
    y = _hip_conv.conv2d(x, w1, 1, 1, 1, 12, 16)
    z = _hip_conv.conv_transpose2d(x, w2)
    return y.float(), z.float()
  y0, z0 = run()
  y0.sum().backward()     # (packs the data-gradient form of w1 as well)
  y0, z0 = y0.detach(), z0.detach()
  with torch.no_grad():
    w1.mul_(-2.0); w2.mul_(3.0)
  assert _hip_conv.repack_all(dev) >= 3
  # (the version counters now match: the calls below must not pack again)
  packs = dict((k, e.buf.data_ptr()) for k, e in _hip_conv._PACKED.items())
  y1, z1 = run()
  assert packs == dict((k, e.buf.data_ptr()) for k, e in _hip_conv._PACKED.items())
  assert float((y1 + 2 * y0).abs().max()) <= float(y1.abs().max()) * 2.0 ** -6
  assert float((z1 - 3 * z0).abs().max()) <= float(z1.abs().max()) * 2.0 ** -6


# (n, cin, h, w, cout, k, stride, groups): odd sizes (partial tiles must not
# count), stride 2 with TF's asymmetric padding, 7 x 7, bottleneck maps
_BNSTAT_CASES = [(4, 32, 37, 53, 64, 3, 1, 2), (2, 64, 40, 66, 32, 5, 2, 1),
                 (8, 32, 128, 384, 32, 7, 1, 2), (8, 512, 4, 12, 512, 3, 1, 2),
                 (6, 128, 17, 9, 256, 3, 2, 3)]


@pytest.mark.parametrize('case', _BNSTAT_CASES)
def test_batch_norm_statistics_from_the_convolution_epilogue(case, dev):
  """lsi_conv2d_fwd_bnstats + lsi_bn_relu_norm against lsi_conv2d_fwd +
  lsi_bn_relu_fwd (whose first pass reads the tensor back): the same bf16
  activations out, mean / rstd to fp32 summation noise -- and against fp64
  moments of the convolution's output; gradients through the pair equal."""
  from lsi.nnutils import _hip_bn, _hip_conv, nets
  n, cin, h, w, cout, k, s, groups = case
  g = torch.Generator().manual_seed(11)
  x = torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16)
  x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
  wt = (torch.randn((cout, cin, k, k), generator=g) * (0.5 / k)).to(dev).requires_grad_(True)
  beta = (torch.randn((cout,), generator=g) * 0.3).to(dev).requires_grad_(True)
  ph, pw = nets._same_pad(h, k, s), nets._same_pad(w, k, s)
  oh, ow = -(-h // s), -(-w // s)
  # separate passes
  y0 = _hip_conv.conv2d(x, wt, s, ph[0], pw[0], oh, ow)
  z0 = _hip_bn.batch_norm_relu(y0, beta, 1e-3, True, groups)
  gz = torch.randn(z0.shape, generator=g).to(dev).to(torch.bfloat16)
  gz = gz.contiguous(memory_format=torch.channels_last)
  gx0, gw0, gb0 = torch.autograd.grad(z0, (x, wt, beta), gz)
  # statistics from the epilogue
  y1 = _hip_conv.conv2d(x, wt, s, ph[0], pw[0], oh, ow, groups)
  assert torch.equal(y1, y0)
  z1 = _hip_bn.batch_norm_relu(y1, beta, 1e-3, True, groups, True)
  mr = z1.grad_fn.saved_tensors[2].clone()
  gx1, gw1, gb1 = torch.autograd.grad(z1, (x, wt, beta), gz)
  yf = y0.detach().double().view(groups, n // groups, cout, oh, ow)
  mean = yf.mean(dim=(1, 3, 4))
  var = yf.var(dim=(1, 3, 4), unbiased=False)
  np.testing.assert_allclose(mr[:, 0].double().cpu().numpy(), mean.cpu().numpy(),
                             rtol=1e-5, atol=1e-5 * float(yf.abs().max()))
  np.testing.assert_allclose(mr[:, 1].double().cpu().numpy(),
                             torch.rsqrt(var + 1e-3).cpu().numpy(), rtol=2e-5)
  # activations: the two sets of constants differ in their last bits, so a value
  # may round to the neighbouring bf16 now and then
  dz = (z1.detach().float() - z0.detach().float()).abs()
  assert float((dz - z0.detach().float().abs() * 2.0 ** -7).max()) <= 1e-6, float(dz.max())
  assert float((dz > 0).float().mean()) < 0.02
  for a, b in ((gx1, gx0), (gw1, gw0), (gb1, gb0)):
    scale = float(b.float().abs().max())
    assert float((a.float() - b.float()).abs().max()) <= 2e-2 * scale
    assert float((a.float() - b.float()).abs().mean()) <= 2e-4 * scale
  # a second pair finds accumulators and counter zero again
  y2 = _hip_conv.conv2d(x, wt, s, ph[0], pw[0], oh, ow, groups)
  z2 = _hip_bn.batch_norm_relu(y2, beta, 1e-3, True, groups, True)
  # (fp32 atomics: the order of the sums differs from call to call)
  same = lambda a, b: float((a.detach().float() - b.detach().float()).abs().max()) <= \
      2.0 ** -7 * float(b.detach().float().abs().max())
  assert same(z2, z1) and float((z2 != z1).float().mean()) < 0.02
  # ... and so does the two-pass kernel that shares the workspace
  z3 = _hip_bn.batch_norm_relu(y0, beta, 1e-3, True, groups)
  assert same(z3, z0) and float((z3 != z0).float().mean()) < 0.02


def test_statistics_hand_over_is_checked_on_the_device(dev):
  """The hand-over convolution -> lsi_bn_relu_norm through the shared workspace
  is tagged (C, groups) on the device.  A consumer without its producer, a
  producer whose consumer never ran followed by a self-accumulating pass, and a
  consumer of somebody else's statistics all write NaN -- never numbers from the
  wrong sums -- and leave the workspace clean: the next regular calls are right.
  lsi_bn_stats_discard drops statistics nobody will consume.  (Round 5's ADVICE:
  a missing lsi_bn_relu_norm left the accumulators dirty for the rest of the
  process, silently.)"""
  from lsi.nnutils import _hip_bn, _hip_conv
  g = torch.Generator().manual_seed(21)
  n, cin, h, w, cout, groups = 4, 64, 12, 20, 64, 2
  x = _clast(torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16))
  wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.1).to(dev)
  beta = torch.zeros((cout,), device=dev)
  y = _hip_conv.conv2d(x, wt, 1, 1, 1, h, w)
  want = _hip_bn.batch_norm_relu(y, beta, 1e-3, True, groups)
  ok = lambda z: bool(torch.isfinite(z.float()).all()) and float(
      (z.float() - want.float()).abs().max()) <= 2.0 ** -7 * float(want.float().abs().max())
  # (1) a consumer without a producer
  z = _hip_bn.batch_norm_relu(y, beta, 1e-3, True, groups, True)
  assert bool(torch.isnan(z.float()).all())
  assert ok(_hip_bn.batch_norm_relu(y, beta, 1e-3, True, groups))
  # (2) a producer whose consumer never ran, then the two-pass kernel
  _hip_conv.conv2d(x, wt, 1, 1, 1, h, w, groups)
  z = _hip_bn.batch_norm_relu(y, beta, 1e-3, True, groups)
  assert bool(torch.isnan(z.float()).all())
  assert ok(_hip_bn.batch_norm_relu(y, beta, 1e-3, True, groups))       # clean again
  y1 = _hip_conv.conv2d(x, wt, 1, 1, 1, h, w, groups)
  assert ok(_hip_bn.batch_norm_relu(y1, beta, 1e-3, True, groups, True))  # and the pair works
  # (3) statistics of another layer (32 channels, one group) in the workspace
  w32 = (torch.randn((32, cin, 3, 3), generator=g) * 0.1).to(dev)
  _hip_conv.conv2d(x, w32, 1, 1, 1, h, w, 1)
  z = _hip_bn.batch_norm_relu(y, beta, 1e-3, True, groups, True)
  assert bool(torch.isnan(z.float()).all())
  y1 = _hip_conv.conv2d(x, wt, 1, 1, 1, h, w, groups)
  assert ok(_hip_bn.batch_norm_relu(y1, beta, 1e-3, True, groups, True))
  # (4) discard: what the Python layer does when the consumer cannot run
  _hip_conv.conv2d(x, wt, 1, 1, 1, h, w, groups)
  _hip_bn.discard_stats(tuple(y.shape), dev, 1, groups)
  assert ok(_hip_bn.batch_norm_relu(y, beta, 1e-3, True, groups))
  # ... and the backward's statistics pass shares the workspace
  yb = y.clone().requires_grad_(True)
  zb = _hip_bn.batch_norm_relu(yb, beta, 1e-3, True, groups)
  gz = _clast(torch.randn(zb.shape, generator=g).to(dev).to(torch.bfloat16))
  g0, = torch.autograd.grad(zb, yb, gz, retain_graph=True)
  _hip_conv.conv2d(x, wt, 1, 1, 1, h, w, groups)            # dirty
  g1, = torch.autograd.grad(zb, yb, gz, retain_graph=True)
  assert bool(torch.isnan(g1.float()).all())
  g2, = torch.autograd.grad(zb, yb, gz)
  assert torch.isfinite(g2.float()).all() and float((g2.float() - g0.float()).abs().max()) <= \
      2.0 ** -6 * float(g0.float().abs().max())


def test_transposed_convolution_leaves_the_statistics_too(dev):
  from lsi.nnutils import _hip_bn, _hip_conv
  g = torch.Generator().manual_seed(12)
  n, cin, h, w, cout, groups = 4, 64, 9, 13, 32, 2
  x = torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16)
  x = x.contiguous(memory_format=torch.channels_last)
  wt = (torch.randn((cin, cout, 4, 4), generator=g) * 0.1).to(dev)
  beta = torch.zeros((cout,), device=dev)
  y0 = _hip_conv.conv_transpose2d(x, wt)
  beta.requires_grad_(True)
  y1 = _hip_conv.conv_transpose2d(x, wt, 2, 1, groups)
  assert torch.equal(y1, y0)
  z1 = _hip_bn.batch_norm_relu(y1, beta, 1e-3, True, groups, True)
  mr = z1.grad_fn.saved_tensors[2]
  yf = y0.double().view(groups, n // groups, cout, 2 * h, 2 * w)
  np.testing.assert_allclose(mr[:, 0].double().cpu().numpy(), yf.mean(dim=(1, 3, 4)).cpu().numpy(),
                             rtol=1e-5, atol=1e-5 * float(yf.abs().max()))
  np.testing.assert_allclose(
      mr[:, 1].double().cpu().numpy(),
      torch.rsqrt(yf.var(dim=(1, 3, 4), unbiased=False) + 1e-3).cpu().numpy(), rtol=2e-5)
  z0 = _hip_bn.batch_norm_relu(y0, beta, 1e-3, True, groups)
  assert float((z1.detach().float() - z0.detach().float()).abs().max()) <= 2.0 ** -7 * float(z0.detach().float().abs().max())


# (n, c1, c2, h, w, cout, k): the heads' upcnv2b / upcnv3b and the U-Net's icnv
# layers in small; odd sizes
@pytest.mark.parametrize('case', [(2, 64, 32, 21, 37, 64, 3), (2, 128, 64, 16, 24, 128, 3),
                                  (4, 512, 512, 4, 12, 512, 3), (1, 64, 64, 9, 70, 32, 5)])
def test_convolution_over_a_skip_connection_reads_the_two_tensors(case, dev):
  """lsi_conv2d_fwd_cat / _bwd_data_cat / _wgrad_cat against the same kernels on
  the concatenated tensor: the same products in the same order -- equal bits."""
  from lsi.nnutils import _hip_conv, nets
  n, c1, c2, h, w, cout, k = case
  g = torch.Generator().manual_seed(21)
  mk = lambda c: torch.randn((n, c, h, w), generator=g).to(dev).to(torch.bfloat16) \
      .contiguous(memory_format=torch.channels_last).requires_grad_(True)
  x1, x2 = mk(c1), mk(c2)
  wt = (torch.randn((cout, c1 + c2, k, k), generator=g) * (0.5 / k)).to(dev).requires_grad_(True)
  ph, pw = nets._same_pad(h, k, 1), nets._same_pad(w, k, 1)
  assert _hip_conv.cat_supported(x1, x2, cout, k, 1)
  y0 = _hip_conv.conv2d(torch.cat([x1, x2], dim=1).contiguous(memory_format=torch.channels_last),
                        wt, 1, ph[0], pw[0], h, w)
  y1 = _hip_conv.conv2d_cat(x1, x2, wt, 1, ph[0], pw[0], h, w)
  assert torch.equal(y1, y0)
  gy = torch.randn(y0.shape, generator=g).to(dev).to(torch.bfloat16)
  gy = gy.contiguous(memory_format=torch.channels_last)
  a1, a2, aw = torch.autograd.grad(y0, (x1, x2, wt), gy)
  b1, b2, bw = torch.autograd.grad(y1, (x1, x2, wt), gy)
  assert torch.equal(a1, b1) and torch.equal(a2, b2)
  assert float((aw - bw).abs().max()) <= 1e-6 * float(aw.abs().max())


def test_skip_connection_layer_module_matches_the_concatenated_form(dev, monkeypatch):
  """SlimConv2d.forward_cat (two-tensor kernels + statistics in the epilogue)
  against forward(torch.cat(...)) with LSI_CAT_CONV off."""
  from lsi.nnutils import _hip_conv, nets
  g = torch.Generator().manual_seed(22)
  layer = nets.SlimConv2d(192, 128, 3, 1).to(dev)
  mk = lambda c: torch.randn((4, c, 16, 24), generator=g).to(dev).to(torch.bfloat16) \
      .contiguous(memory_format=torch.channels_last).requires_grad_(True)
  x1, x2 = mk(128), mk(64)
  with nets.bn_groups(2):
    z1 = layer(x1, x2)
    monkeypatch.setattr(_hip_conv, 'CAT_CONV', False)
    z0 = layer(x1, x2)
  gz = torch.randn(z0.shape, generator=g).to(dev).to(torch.bfloat16)
  p = [x1, x2, layer.conv.weight, layer.bn.beta]
  g1 = torch.autograd.grad(z1, p, gz)
  g0 = torch.autograd.grad(z0, p, gz)
  d = (z1.detach().float() - z0.detach().float()).abs()
  assert float(d.max()) <= 2.0 ** -7 * float(z0.detach().float().abs().max())
  for a, b in zip(g1, g0):
    scale = float(b.float().abs().max())
    assert float((a.float() - b.float()).abs().max()) <= 2e-2 * scale
    assert float((a.float() - b.float()).abs().mean()) <= 2e-4 * scale


def test_packs_of_channels_last_parameters_and_of_silent_updates(dev):
  """(1) A parameter with channels-last strides (module.to(memory_format=
  torch.channels_last): what the trainer's model has) is packed in place
  (mode | 2) to the same operand as its contiguous copy.  (2) A parameter that
  is being trained is packed on every forward call while nobody manages its
  packs: an update that does not move the version counter (torch's fused
  optimisers; `.data` here) must be seen by the next call, forward and
  backward."""
  from lsi.nnutils import _hip_conv
  g = torch.Generator().manual_seed(31)
  x = _clast(torch.randn((2, 64, 10, 12), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  w0 = (torch.randn((32, 64, 3, 3), generator=g) * 0.1).to(dev)
  wc = w0.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
  assert not wc.is_contiguous() and _hip_conv._pack_layout(wc) == 2
  yc = _hip_conv.conv2d(x, wc, 1, 1, 1, 10, 12)
  y0 = _hip_conv.conv2d(x, w0, 1, 1, 1, 10, 12)
  assert torch.equal(yc, y0)
  gy = _clast(torch.randn(y0.shape, generator=g).to(dev).to(torch.bfloat16))
  gx_c, = torch.autograd.grad(yc, x, gy)
  v = wc._version
  wc.data.mul_(-2.0)                # (no version bump)
  assert wc._version == v
  y2 = _hip_conv.conv2d(x, wc, 1, 1, 1, 10, 12)
  assert float((y2.float() + 2 * y0.float()).abs().max()) <= float(y2.float().abs().max()) * 2.0 ** -6
  gx_2, = torch.autograd.grad(y2, x, gy)
  assert float((gx_2.float() + 2 * gx_c.float()).abs().max()) <= float(gx_2.float().abs().max()) * 2.0 ** -6
  # a direct repack_all() refreshes the packs but does not make them trusted:
  # only an optimiser that keeps stepping does (the global post-step hook)
  wc.data.mul_(-0.5)
  assert _hip_conv.repack_all(dev) >= 2
  assert not any(e.managed for k, e in _hip_conv._PACKED.items() if k[0] == id(wc))
  y3 = _hip_conv.conv2d(x, wc, 1, 1, 1, 10, 12)
  assert float((y3.float() - y0.float()).abs().max()) <= float(y0.float().abs().max()) * 2.0 ** -6
  wc.data.mul_(3.0)                 # ... so this silent write is seen as well
  y4 = _hip_conv.conv2d(x, wc, 1, 1, 1, 10, 12)
  assert float((y4.float() - 3 * y0.float()).abs().max()) <= float(y4.float().abs().max()) * 2.0 ** -6


def test_any_optimizer_loop_keeps_the_packs_fresh(dev):
  """A plain training loop -- torch.optim.Adam(fused=True), which updates
  parameters WITHOUT moving their version counters, no Trainer, nobody calling
  repack_all() -- has to run every step on the weights of that step: the global
  optimiser post-step hook re-makes the packs of the parameters the stepping
  optimiser owns (one launch) and only then are they trusted.  A second
  optimiser over the same parameters is covered the same way.  (Round 5's
  ADVICE: once packs were marked managed they were trusted for ever, and any loop
  other than the Trainer's trained on stale weights.)"""
  import ctypes
  from lsi import _C
  from lsi.nnutils import _hip_conv, nets
  torch.manual_seed(5)
  layers = torch.nn.ModuleList([nets.SlimConv2d(64, 64, 3, 1), nets.SlimConv2d(64, 32, 3, 2),
                                nets.SlimConvTranspose2d(32, 32)]).to(dev)
  layers = layers.to(memory_format=torch.channels_last)
  params = list(layers.parameters())
  opt = torch.optim.Adam(params, lr=1e-2, fused=True)
  opt2 = torch.optim.SGD(params, lr=1e-2)
  g = torch.Generator().manual_seed(6)
  # (the input wants a gradient: the first layer's data-gradient pack exists, too)
  x = _clast(torch.randn((2, 64, 16, 32), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  lib = _C.lib()

  def fresh(e, w):
    buf = torch.empty_like(e.buf)
    src, cl = _hip_conv._pack_source(w)
    rc = lib.lsi_conv2d_pack(ctypes.byref(e.desc), e.mode | cl, _C.ptr(src), _C.ptr(buf),
                             buf.numel() * 2, _C.stream_ptr(w.device))
    assert rc == 0
    return buf

  mine = lambda: [(k, e) for k, e in _hip_conv._PACKED.items()
                  if e.wref() is not None and any(e.wref() is p for p in params)]
  losses = []
  for step in range(6):
    y = x
    for l in layers:
      y = l(y)
    loss = y.float().square().mean()
    for p in params:
      p.grad = None
    loss.backward()
    (opt if step != 3 else opt2).step()      # step 3: ANOTHER optimiser updates them
    torch.cuda.synchronize()
    packs = mine()
    assert len(packs) >= 6, len(packs)       # three layers, both directions
    for k, e in packs:
      assert e.managed, (step, k)
      assert torch.equal(fresh(e, e.wref()), e.buf), (step, k)
    losses.append(float(loss))
  assert losses[-1] < losses[0]
  # the trusted packs are verified on demand (LSI_PACK_CHECK): a write through
  # .data behind everybody's back is then an error, not a silently stale layer
  old = _hip_conv.PACK_CHECK
  _hip_conv.PACK_CHECK = 1
  try:
    layers[0](x)
    layers[0].conv.weight.data.mul_(1.5)
    with pytest.raises(RuntimeError, match='stale packed weights'):
      layers[0](x)
  finally:
    _hip_conv.PACK_CHECK = old
    _hip_conv.repack_all(dev)


def test_weight_gradient_comes_in_the_parameters_layout(dev):
  """lsi_conv2d_wgrad_cat(weight_layout=2): the gradient of a channels-last
  parameter has its strides (autograd then takes it instead of cloning it) and
  the values of the contiguous one -- plain and two-tensor input, narrow and wide
  layers (the two folds)."""
  from lsi.nnutils import _hip_conv
  g = torch.Generator().manual_seed(41)
  for (n, c1, c2, h, w, cout, k) in [(2, 64, 0, 12, 20, 32, 3), (2, 64, 32, 9, 17, 64, 3),
                                     (4, 512, 0, 4, 12, 512, 3), (1, 32, 0, 11, 33, 32, 7)]:
    mk = lambda c: _clast(torch.randn((n, c, h, w), generator=g).to(dev).to(torch.bfloat16))
    x1 = mk(c1); x2 = mk(c2) if c2 else None
    w0 = (torch.randn((cout, c1 + c2, k, k), generator=g) * 0.1).to(dev)
    wc = w0.clone().contiguous(memory_format=torch.channels_last)
    gy = mk(cout)
    d = _hip_conv._conv_desc(n, h, w, c1 + c2, h, w, cout, k, k, 1, k // 2, k // 2)
    a = _hip_conv._igemm_wgrad(d, x1, gy, w0, x2)
    b = _hip_conv._igemm_wgrad(d, x1, gy, wc, x2)
    assert a.is_contiguous() and b.stride() == wc.stride()
    assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())


# (n, cin, h, w, cout, k, stride): the U-Net's bottleneck maps at 256 x 768
# (cnv5b ... cnv7b, reference nets.py:281-289) -- tiles alone are 64 - 256
# workgroups -- and one launch that fills the chip without a split
_SPLITK_CASES = [(8, 512, 8, 24, 512, 3, 1), (8, 512, 8, 24, 512, 3, 2), (8, 512, 4, 12, 512, 3, 1),
                 (8, 512, 2, 6, 512, 3, 1), (2, 256, 16, 48, 512, 3, 2), (2, 160, 5, 7, 128, 5, 1)]


@pytest.mark.parametrize('case', _SPLITK_CASES)
def test_split_over_the_input_channels_matches_the_unsplit_kernel(case, dev, monkeypatch):
  """lsi_conv2d_run with a workspace (the contraction split over the input
  channels, fp32 partial sums folded by a second kernel) against the same call
  without one: forward, data gradient, the batch-norm sums of the epilogue; and
  against fp32 autograd of the bf16-rounded operands.  The two differ by the
  fp32 summation order only: at most one bf16 ulp where a sum sits on a rounding
  boundary."""
  import ctypes
  from lsi import _C
  from lsi.nnutils import _hip_bn, _hip_conv, nets
  n, cin, h, w, cout, k, s = case
  g = torch.Generator().manual_seed(31)
  x = _clast(torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  wt = (torch.randn((cout, cin, k, k), generator=g) * (0.7 / (k * cin ** 0.5))).to(dev)
  beta = (torch.randn((cout,), generator=g) * 0.3).to(dev)
  ph, pw = nets._same_pad(h, k, s), nets._same_pad(w, k, s)
  oh, ow = -(-h // s), -(-w // s)
  d = _hip_conv._conv_desc(n, h, w, cin, oh, ow, cout, k, k, s, ph[0], pw[0])
  nbytes = [int(_C.lib().lsi_conv2d_workspace_bytes(ctypes.byref(d), m)) for m in (0, 1)]
  assert all(b > 0 for b in nbytes), nbytes     # (these launches do split)
  gy = _clast(torch.randn((n, cout, oh, ow), generator=g).to(dev).to(torch.bfloat16))
  outs = {}
  for split in (True, False):
    monkeypatch.setattr(_hip_conv, 'SPLITK', split)
    y = _hip_conv.conv2d(x, wt, s, ph[0], pw[0], oh, ow, 2)
    z = _hip_bn.batch_norm_relu(y, beta, 1e-3, True, 2, True)
    mr = z.grad_fn.saved_tensors[2].clone()
    gx, = torch.autograd.grad(y, x, gy)
    outs[split] = (y.detach().float(), z.detach().float(), mr, gx.float())
  for a, b in zip(outs[True], outs[False]):
    assert torch.isfinite(a).all()
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
  # the statistics: those of the split call's own rounded outputs
  ys = outs[True][0].double().view(2, n // 2, cout, oh, ow)
  mean = ys.mean(dim=(1, 3, 4))
  assert float((outs[True][2][:, 0].double() - mean).abs().max()) <= 1e-4 * float(
      ys.abs().max())
  # fp32 autograd of the same operands
  xf = x.detach().float().requires_grad_(True)
  xp = F.pad(xf, (pw[0], pw[1], ph[0], ph[1]))
  want = F.conv2d(xp, wt.to(torch.bfloat16).float(), None, s)
  wgx, = torch.autograd.grad(want, xf, gy.float())
  assert float((outs[True][0] - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max()) + 1e-3
  assert float((outs[True][3] - wgx).abs().max()) <= 2.0 ** -7 * float(wgx.abs().max())


def test_split_convolution_over_a_skip_connection_and_without_workspace(dev, monkeypatch):
  """The two-tensor variants split, too (input as two tensors forward, gradient
  into two tensors backward), identically to the split call on the concatenated
  tensor; a workspace that is too small is not an error (no split)."""
  import ctypes
  from lsi import _C
  from lsi.nnutils import _hip_conv
  g = torch.Generator().manual_seed(32)
  n, c1, c2, h, w, cout = 8, 512, 512, 4, 12, 512
  a = _clast(torch.randn((n, c1, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  b = _clast(torch.randn((n, c2, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
  wt = (torch.randn((cout, c1 + c2, 3, 3), generator=g) * 0.01).to(dev)
  gy = _clast(torch.randn((n, cout, h, w), generator=g).to(dev).to(torch.bfloat16))
  y1 = _hip_conv.conv2d_cat(a, b, wt, 1, 1, 1, h, w)
  ga1, gb1 = torch.autograd.grad(y1, (a, b), gy)
  ab = _clast(torch.cat([a.detach(), b.detach()], 1)).requires_grad_(True)
  y2 = _hip_conv.conv2d(ab, wt, 1, 1, 1, h, w)
  gab, = torch.autograd.grad(y2, ab, gy)
  assert torch.equal(y1, y2)
  assert torch.equal(torch.cat([ga1, gb1], 1), gab)
  # a workspace one byte short: the unsplit kernel, no error
  lib = _C.lib()
  d = _hip_conv._conv_desc(n, h, w, c1 + c2, h, w, cout, 3, 3, 1, 1, 1)
  need = int(lib.lsi_conv2d_workspace_bytes(ctypes.byref(d), 0))
  assert need > 0
  ws = torch.empty((need,), dtype=torch.uint8, device=dev)
  packed = _hip_conv._packed(d, 0, wt)
  out = torch.empty_like(y2)
  io = _C.LsiConvIO()
  io.x, io.packed, io.out = ab.data_ptr(), packed.data_ptr(), out.data_ptr()
  io.workspace, io.workspace_bytes = ws.data_ptr(), need - 1
  assert lib.lsi_conv2d_run(ctypes.byref(d), 0, ctypes.byref(io), _C.stream_ptr(dev)) == 0
  monkeypatch.setattr(_hip_conv, 'SPLITK', False)
  y3 = _hip_conv.conv2d(ab, wt, 1, 1, 1, h, w)
  assert torch.equal(out, y3)
  io.workspace_bytes = need
  assert lib.lsi_conv2d_run(ctypes.byref(d), 0, ctypes.byref(io), _C.stream_ptr(dev)) == 0
  assert torch.equal(out, y2)
  io.workspace = None
  assert lib.lsi_conv2d_run(ctypes.byref(d), 0, ctypes.byref(io), _C.stream_ptr(dev)) == -2  # LSI_ENULL


def test_weight_gradients_on_the_side_stream_equal_the_in_stream_ones(dev):
  """_hip_conv.enable_wgrad_stream: a backward pass whose weight gradients run on
  a second stream (forked behind the incoming gradient, joined by the callback at
  the end of the pass) gives bit-identical gradients (a chain without batch
  norms: every kernel of it is deterministic); a parameter that already holds a
  gradient (accumulation) stays on the main stream and accumulates."""
  from lsi.nnutils import _hip_conv, nets
  g = torch.Generator().manual_seed(10)
  n, h, w = 4, 24, 40
  mk = lambda *shape: (torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
                       ).to(dev).requires_grad_(True)
  w1, w2, w3, w5 = mk(64, 64, 3, 3), mk(128, 64, 3, 3), mk(128, 128, 3, 3), mk(32, 64, 5, 5)
  wt = mk(128, 64, 4, 4)     # (ConvTranspose2d: in x out x k x k)
  params = [w1, w2, w3, wt, w5]
  x = _clast(torch.randn((n, 64, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)

  def net():
    y = torch.relu(_hip_conv.conv2d(x, w1, 1, 1, 1, h, w))
    p2 = nets._same_pad(h, 3, 2)[0], nets._same_pad(w, 3, 2)[0]
    y = torch.relu(_hip_conv.conv2d(y, w2, 2, p2[0], p2[1], h // 2, w // 2))
    y = torch.relu(_hip_conv.conv2d(y, w3, 1, 1, 1, h // 2, w // 2))
    y = torch.relu(_hip_conv.conv_transpose2d(y, wt))
    y = _hip_conv.conv2d(y, w5, 1, 2, 2, h, w)
    return y.float().square().mean()

  def grads():
    for p in params:
      p.grad = None
    x.grad = None
    net().backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in params] + [x.grad.clone()]

  old = _hip_conv.enable_wgrad_stream(False)
  try:
    want = grads()
    for a, b in zip(grads(), want):
      assert torch.equal(a, b)                    # (deterministic in-stream)
    _hip_conv.enable_wgrad_stream(True)
    for _ in range(3):
      got = grads()
      assert not _hip_conv._SIDE_PENDING          # (joined and released)
      for a, b in zip(got, want):
        assert torch.equal(a, b)
    # accumulation into existing gradients: main stream, twice the values
    net().backward()
    torch.cuda.synchronize()
    for p, b in zip(params, want):
      assert float((p.grad - 2 * b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-12
    # torch.autograd.grad (no .grad involved) joins, too
    for p in params:
      p.grad = None
    gs = torch.autograd.grad(net(), params)
    torch.cuda.synchronize()
    for a, b in zip(gs, want):
      assert torch.equal(a, b)
    # a backward pass that raises after its weight gradients were forked (the
    # engine may never run its join callback) leaves nothing behind for the next
    def boom(_g):
      raise RuntimeError('boom')
    for p in params:
      p.grad = None
    hk = x.register_hook(boom)
    with pytest.raises(RuntimeError, match='boom'):
      net().backward()
    hk.remove()
    for a, b in zip(grads(), want):
      assert torch.equal(a, b)
    assert not _hip_conv._SIDE_PENDING
  finally:
    _hip_conv.enable_wgrad_stream(old)
