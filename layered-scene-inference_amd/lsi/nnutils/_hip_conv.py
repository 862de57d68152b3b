"""torch.autograd binding of the MFMA 3x3 convolution over 32 input channels
(csrc/lsi_conv.hip; include/lsi_hip.h: lsi_conv3x3_c32_fwd) -- the
full-resolution layers of the LDI heads (reference nets.py:104-111, 150-158)
on bf16 channels-last activations.

  conv3x3_c32(x, weight)            32 -> 16 | 32 channels, bf16 out (pre batch norm)
  conv3x3_c32_sigmoid(x, weight, bias)   32 -> <= 4 channels, bias + sigmoid, fp32 RGBD pixels

Forward and the data gradient of the 32 -> 32 layer run on the hand-written
kernel (the data gradient of a stride-1 SAME convolution is the same
convolution with the weights flipped and transposed); its weight gradient (a
GEMM with K = all pixels) on lsi_conv3x3_wgrad (csrc/lsi_conv_wgrad.hip: both
operands transposed by the LDS transpose read), which also takes the weight
gradients of the other 3x3 layers at full and half resolution
(conv3x3_lib_own_wgrad: forward and data gradient on MIOpen).  The prediction head is hand-written end to
end: lsi_conv3x3_pred_bwd forms sigmoid'(z) * g in registers and computes the
data gradient (K = 9 taps x 4 channels) and the 4 x 288 + 4 weight / bias
gradients (K = pixels) on the matrix cores.
"""
import torch

from lsi import _C


def supported(x, cin, cout, k, stride, pred):
  """bf16 channels-last activations on the GPU, 3x3 stride 1, 32 input
  channels, widths that are multiples of 16 pixels."""
  if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16):
    return False
  if k != 3 or stride != 1 or cin != 32 or x.shape[1] != 32 or x.shape[3] % 16:
    return False
  if pred:
    if not 1 <= cout <= 4:
      return False
  elif cout not in (16, 32):
    return False
  return (x.is_contiguous(memory_format=torch.channels_last) and
          x.data_ptr() % 16 == 0)


def _launch(x, weight, cout, mode, bias=None):
  """mode 0: 32 -> cout, bf16 out; 1: prediction head (bias, sigmoid, fp32 RGBD
  pixels); 2: data gradient of the 32 -> 32 layer.  `weight`: the layer's
  parameter (cout x 32 x 3 x 3); the kernel reads it as fp32 and rounds to bf16
  itself -- no packing or casting launches."""
  n, _, h, w = x.shape
  dev = x.device
  weight = weight.detach()
  if weight.dtype != torch.float32 or not weight.is_contiguous():
    weight = weight.float().contiguous()
  if mode == 1:
    out = torch.empty((n, 4, h, w), dtype=torch.float32, device=dev,
                      memory_format=torch.channels_last)
  else:
    out = torch.empty((n, cout, h, w), dtype=torch.bfloat16, device=dev,
                      memory_format=torch.channels_last)
  rc = _C.lib().lsi_conv3x3_c32_fwd(n, h, w, cout, mode, _C.ptr(x), _C.ptr(weight),
                                    _C.ptr(bias), 1.0, _C.ptr(out),
                                    _C.stream_ptr(dev))
  _C.check(rc, 'lsi_conv3x3_c32_fwd')
  return out


class _Conv3x3C32(torch.autograd.Function):
  """32 -> cout (16 | 32) channels, no bias, bf16 out."""

  @staticmethod
  def forward(ctx, x, weight):
    ctx.save_for_backward(x, weight)
    return _launch(x, weight, weight.shape[0], 0)

  @staticmethod
  def backward(ctx, g):
    x, weight = ctx.saved_tensors
    g = g.contiguous(memory_format=torch.channels_last)
    gx = gw = None
    cout = weight.shape[0]
    if ctx.needs_input_grad[0]:
      if cout == 32 and g.dtype == torch.bfloat16 and g.data_ptr() % 16 == 0:
        # dL/dx = conv(g, W flipped in space, transposed in channels): mode 2
        gx = _launch(g, weight, 32, 2)
      else:
        gx = torch.ops.aten.convolution_backward(
            g, x, weight.to(g.dtype), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
            [True, False, False])[0]
    if ctx.needs_input_grad[1]:
      gw = _weight_grad(x, g, weight)
    return gx, gw


# ---- weight gradient on the matrix cores (csrc/lsi_conv_wgrad.hip) ---------------
import os
WGRAD_MIN_PIXELS = int(os.environ.get('LSI_WGRAD_MIN_PIXELS', '200000'))  # below: aten (tools/time_wgrad.py)
_WGRAD_WS = {}


def wgrad_supported(x, cin, cout, k, stride):
  """3x3 stride-1 layers on bf16 channels-last GPU activations with channel
  counts that are multiples of 32, at resolutions where the K = pixels GEMM
  fills the chip (lsi_conv3x3_wgrad: 72 us against MIOpen's 227 at 8x32x256x768,
  108 against 238 at 8x96x128x384; a tie at 8x192x64x192)."""
  return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and k == 3 and
          stride == 1 and cin % 32 == 0 and cout % 32 == 0 and x.shape[1] == cin and
          x.shape[0] * x.shape[2] * x.shape[3] >= WGRAD_MIN_PIXELS and
          x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0)


def _wgrad_workspace(dev, nbytes):
  """Partial sums of the pixel blocks: per (device, stream) the largest buffer
  asked for so far; smaller ones stay alive (kernels in flight may use them)."""
  key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
  bufs = _WGRAD_WS.setdefault(key, [])
  if not bufs or bufs[-1].numel() * 4 < nbytes:
    bufs.append(torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev))
  return bufs[-1]


def _weight_grad(x, g, weight):
  """dL/dW of a 3x3 stride-1 SAME convolution: lsi_conv3x3_wgrad where it
  applies, aten.convolution_backward (MIOpen) elsewhere."""
  cout, cin = weight.shape[:2]
  if (wgrad_supported(x, cin, cout, 3, 1) and g.dtype == torch.bfloat16 and
      g.data_ptr() % 16 == 0 and g.is_contiguous(memory_format=torch.channels_last)):
    n, _, h, w = x.shape
    dev = x.device
    lib = _C.lib()
    nbytes = lib.lsi_conv3x3_wgrad_workspace_bytes(n, h, w, cin, cout)
    ws = _wgrad_workspace(dev, nbytes)
    gw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
    rc = lib.lsi_conv3x3_wgrad(n, h, w, cin, cout, _C.ptr(x), _C.ptr(g), _C.ptr(gw),
                               _C.ptr(ws), ws.numel() * 4, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_conv3x3_wgrad')
    return gw.to(weight.dtype)
  return torch.ops.aten.convolution_backward(
      g, x, weight.to(g.dtype), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
      [False, True, False])[1].to(weight.dtype)


class _Conv3x3LibOwnWgrad(torch.autograd.Function):
  """A 3x3 stride-1 SAME convolution without bias whose forward and data
  gradient stay on the library (MIOpen) and whose weight gradient runs on
  lsi_conv3x3_wgrad: the layers of the heads the forward kernel does not take
  (`upcnv2b`: 96 -> 64 at half resolution)."""

  @staticmethod
  def forward(ctx, x, weight):
    ctx.save_for_backward(x, weight)
    return torch.ops.aten.convolution(x, weight.to(x.dtype), None, [1, 1], [1, 1], [1, 1],
                                      False, [0, 0], 1)

  @staticmethod
  def backward(ctx, g):
    x, weight = ctx.saved_tensors
    g = g.contiguous(memory_format=torch.channels_last)
    gx = gw = None
    if ctx.needs_input_grad[0]:
      gx = torch.ops.aten.convolution_backward(
          g, x, weight.to(g.dtype), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
          [True, False, False])[0]
    if ctx.needs_input_grad[1]:
      gw = _weight_grad(x, g, weight)
    return gx, gw


def conv3x3_lib_own_wgrad(x, weight):
  return _Conv3x3LibOwnWgrad.apply(x, weight)


def _pred_workspace(dev, n, h, w):
  """Partial sums of the head's weight gradient (the same never-dropped buffers
  as the other weight gradients)."""
  return _wgrad_workspace(dev, _C.lib().lsi_conv3x3_pred_bwd_workspace_bytes(n, h, w))


class _Conv3x3C32Sigmoid(torch.autograd.Function):
  """32 -> cout (<= 4) channels + bias + sigmoid, fp32 out N x 4 x H x W
  (channels last: RGBD pixels; channels past cout hold sigmoid(0))."""

  @staticmethod
  def forward(ctx, x, weight, bias):
    cout = weight.shape[0]
    y = _launch(x, weight, cout, 1,
                None if bias is None else bias.detach().float().contiguous())
    ctx.save_for_backward(x, weight, y)
    ctx.has_bias = bias is not None
    return y[:, :cout] if cout < 4 else y

  @staticmethod
  def backward(ctx, g):
    x, weight, y = ctx.saved_tensors
    cout = weight.shape[0]
    n, _, h, w = x.shape
    if cout == 4 and g.dtype == torch.float32:
      # lsi_conv3x3_pred_bwd: gz = g * y * (1 - y) is formed in registers; the
      # data gradient on the matrix cores (K = 9 taps x 4 channels), weight and
      # bias gradients on the matrix cores too (K = pixels, gz as bf16 hi + lo;
      # partial sums of the pixel blocks in the workspace, folded by a second
      # kernel)
      g = g.contiguous(memory_format=torch.channels_last)
      dev = x.device
      gx = (torch.empty((n, 32, h, w), dtype=torch.bfloat16, device=dev,
                        memory_format=torch.channels_last)
            if ctx.needs_input_grad[0] else None)
      want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
      gwb = ws = None
      if want_w:
        gwb = torch.empty((cout * 288 + cout,), dtype=torch.float32, device=dev)
        ws = _pred_workspace(dev, n, h, w)
      wt = weight.detach()
      if wt.dtype != torch.float32 or not wt.is_contiguous():
        wt = wt.float().contiguous()
      rc = _C.lib().lsi_conv3x3_pred_bwd(n, h, w, cout, _C.ptr(g), _C.ptr(y), _C.ptr(x),
                                         _C.ptr(wt), _C.ptr(gx), _C.ptr(gwb),
                                         _C.ptr(ws), 0 if ws is None else ws.numel() * 4,
                                         _C.stream_ptr(dev))
      _C.check(rc, 'lsi_conv3x3_pred_bwd')
      gw = gb = None
      if want_w:
        gw = gwb[:cout * 288].view(cout, 32, 3, 3).to(weight.dtype)
        gb = gwb[cout * 288:].to(weight.dtype) if ctx.has_bias else None
      return gx, gw, gb
    ys = y[:, :cout]
    gz = (g * ys * (1.0 - ys)).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1],
            ctx.has_bias and ctx.needs_input_grad[2]]
    gx, gw, gb = torch.ops.aten.convolution_backward(
        gz, x, weight.to(torch.bfloat16), [cout] if ctx.has_bias else None, [1, 1],
        [1, 1], [1, 1], False, [0, 0], 1, mask)
    return (gx, None if gw is None else gw.to(weight.dtype),
            None if gb is None else gb.to(weight.dtype))


def conv3x3_c32(x, weight):
  return _Conv3x3C32.apply(x, weight)


def conv3x3_c32_sigmoid(x, weight, bias):
  return _Conv3x3C32Sigmoid.apply(x, weight, bias)
