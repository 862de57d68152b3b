"""torch.autograd binding of the fused batch-norm + ReLU kernels
(csrc/lsi_bn.hip; include/lsi_hip.h: lsi_bn_relu_fwd / _bwd) for channels-last
activations on a ROCm device."""
import threading

import torch

from lsi import _C

_WS = {}
_WS_LOCK = threading.Lock()
_WS_FLOATS = 1 << 20      # 4 MiB: covers every layer of the 256 x 768 networks


_WS_NEED = {}


def _workspace(dev, stream, npix, c, bf16, groups):
  """Zero-filled once per (device, stream), then kept: the kernels leave their
  arrival counter zero, and calls on one stream are ordered.  A buffer that has
  been handed out is never freed (a captured HIP graph has its address baked
  in): a larger need gets a further buffer, the old one stays alive."""
  nkey = (c, bf16, groups)
  need = _WS_NEED.get(nkey)
  if need is None:
    need = int(_C.lib().lsi_bn_workspace_floats(npix, c, bf16, groups))
    if need <= 0:               # (an unsupported shape: never cached; the entry
      return _dummy_ws(dev)     #  point itself reports LSI_EINVAL)
    _WS_NEED[nkey] = need
  key = (dev.index, stream)
  kept = _WS.get(key)
  if kept is not None and kept[-1].numel() >= need:   # (the largest is last)
    return kept[-1]
  with _WS_LOCK:
    kept = _WS.setdefault(key, [])
    for ws in kept:
      if ws.numel() >= need:
        return ws
    ws = torch.zeros((max(need, _WS_FLOATS),), dtype=torch.float32, device=dev)
    kept.append(ws)
  return ws


_DUMMY = {}


def _dummy_ws(dev):
  ws = _DUMMY.get(dev.index)
  if ws is None:
    ws = _DUMMY[dev.index] = torch.zeros((16,), dtype=torch.float32, device=dev)
  return ws


def supported(x, groups=1):
  """Channels-last 4-D fp32 / bf16 CUDA tensor whose channel count the kernels
  take (lsi_hip.h); `groups` sub-batches along N with their own statistics."""
  if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
    return False
  n, c, h, w = x.shape
  if groups < 1 or n % groups:
    return False
  nv = 8 if x.dtype == torch.bfloat16 else 4
  if c % nv or c > 2048:
    return False
  lpp = c // nv
  if lpp > 256 or lpp & (lpp - 1):
    return False
  return (x.is_contiguous(memory_format=torch.channels_last) and
          x.data_ptr() % 16 == 0)


class _BnRelu(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, beta, eps, relu, groups, prestat=False):
    """prestat: a producer has left the sums of x and x * x in the workspace
    (lsi_conv2d_*_bnstats on this stream, just before this call): one pass."""
    if not x.is_cuda:
      raise RuntimeError('fused batch norm needs a tensor on a ROCm GPU')
    dev = x.device
    n, c, h, w = x.shape
    npix = (n // groups) * h * w          # per group: its own statistics
    bf16 = int(x.dtype == torch.bfloat16)
    lib = _C.lib()
    stream = _C.stream_ptr(dev)
    ws = _workspace(dev, stream, npix, c, bf16, groups)
    y = torch.empty_like(x)               # (x is channels-last: supported())
    beta_f = beta_f32(beta)
    mean_rstd = torch.empty((groups, 2, c), dtype=torch.float32, device=dev)
    if prestat:
      rc = lib.lsi_bn_relu_norm(x.data_ptr(), y.data_ptr(), beta_f.data_ptr(), ws.data_ptr(),
                                mean_rstd.data_ptr(), npix, c, bf16, int(relu), float(eps),
                                groups, stream)
      if rc:
        _C.check(rc, 'lsi_bn_relu_norm')
    else:
      # (N-major storage: the groups are consecutive blocks of npix * C values)
      rc = lib.lsi_bn_relu_fwd(x.data_ptr(), y.data_ptr(), beta_f.data_ptr(), ws.data_ptr(),
                               mean_rstd.data_ptr(), npix, c, bf16, int(relu),
                               float(eps), groups, stream)
      if rc:
        _C.check(rc, 'lsi_bn_relu_fwd')
    ctx.save_for_backward(x, beta_f, mean_rstd)
    ctx.relu = int(relu)
    ctx.groups = groups
    return y

  @staticmethod
  def backward(ctx, dy):
    x, beta_f, mean_rstd = ctx.saved_tensors
    dev = x.device
    n, c, h, w = x.shape
    groups = ctx.groups
    npix = (n // groups) * h * w
    bf16 = int(x.dtype == torch.bfloat16)
    if dy.dtype != x.dtype:
      dy = dy.to(x.dtype)
    if not dy.is_contiguous(memory_format=torch.channels_last):
      dy = dy.contiguous(memory_format=torch.channels_last)
    lib = _C.lib()
    stream = _C.stream_ptr(dev)
    ws = _workspace(dev, stream, npix, c, bf16, groups)
    dx = torch.empty_like(x)
    dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
    rc = lib.lsi_bn_relu_bwd(x.data_ptr(), dy.data_ptr(), mean_rstd.data_ptr(),
                             beta_f.data_ptr(), dx.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                             npix, c, bf16, ctx.relu, groups, stream)
    if rc:
      _C.check(rc, 'lsi_bn_relu_bwd')
    return dx, dbeta, None, None, None, None


def beta_f32(beta):
  b = beta.detach()
  if b.dtype != torch.float32 or not b.is_contiguous():
    b = b.float().contiguous()
  return b


def stats_workspace(x_like_shape, dev, bf16, groups):
  """The workspace a producer of statistics (lsi_conv2d_*_bnstats) has to write
  so that batch_norm_relu(..., prestat=) finds them: the one _BnRelu.forward
  will pick for a tensor of this shape on the current stream."""
  n, c, h, w = x_like_shape
  return _workspace(dev, _C.stream_ptr(dev), (n // groups) * h * w, c, bf16, groups)


def discard_stats(x_like_shape, dev, bf16, groups):
  """Drops statistics a producer has left for a batch norm that will not run
  (lsi_bn_stats_discard): the workspace is clean for the next call."""
  ws = stats_workspace(x_like_shape, dev, bf16, groups)
  rc = _C.lib().lsi_bn_stats_discard(ws.data_ptr(), int(groups), _C.stream_ptr(dev))
  if rc:
    _C.check(rc, 'lsi_bn_stats_discard')


def channels_ok(c, bf16=True):
  nv = 8 if bf16 else 4
  lpp = c // nv
  return c % nv == 0 and c <= 2048 and lpp <= 256 and lpp & (lpp - 1) == 0


def batch_norm_relu(x, beta, eps=1e-3, relu=True, groups=1, prestat=False):
  """relu(batch_norm(x) + beta) with batch statistics (slim.batch_norm,
  scale=False) for a channels-last N x C x H x W tensor; same dtype out.
  groups > 1: N is that many sub-batches, each normalised on its own."""
  return _BnRelu.apply(x, beta, eps, relu, groups, prestat)
