"""torch.autograd binding of the fused batch-norm + ReLU kernels
(csrc/lsi_bn.hip; include/lsi_hip.h: lsi_bn_relu_fwd / _bwd) for channels-last
activations on a ROCm device."""
import threading

import torch

from lsi import _C

_WS = {}
_WS_LOCK = threading.Lock()
_WS_FLOATS = 1 << 20      # 4 MiB: covers every layer of the 256 x 768 networks


def _workspace(dev, need):
  """Zero-filled once per (device, stream), then kept: the kernels leave their
  arrival counter zero, and calls on one stream are ordered.  A buffer that has
  been handed out is never freed (a captured HIP graph has its address baked
  in): a larger need gets a further buffer, the old one stays alive."""
  key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
  with _WS_LOCK:
    kept = _WS.setdefault(key, [])
    for ws in kept:
      if ws.numel() >= need:
        return ws
    ws = torch.zeros((max(need, _WS_FLOATS),), dtype=torch.float32, device=dev)
    kept.append(ws)
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
  def forward(ctx, x, beta, eps, relu, groups):
    if not x.is_cuda:
      raise RuntimeError('fused batch norm needs a tensor on a ROCm GPU')
    dev = x.device
    n, c, h, w = x.shape
    npix = (n // groups) * h * w          # per group: its own statistics
    bf16 = int(x.dtype == torch.bfloat16)
    lib = _C.lib()
    ws = _workspace(dev, int(lib.lsi_bn_workspace_floats(npix, c, bf16, groups)))
    y = torch.empty_like(x, memory_format=torch.channels_last)
    mean_rstd = torch.empty((groups, 2, c), dtype=torch.float32, device=dev)
    beta_f = beta.detach().float().contiguous()
    # (N-major storage: the groups are consecutive blocks of npix * C values)
    rc = lib.lsi_bn_relu_fwd(_C.ptr(x), _C.ptr(y), _C.ptr(beta_f), _C.ptr(ws),
                             _C.ptr(mean_rstd), npix, c, bf16, int(relu),
                             float(eps), groups, _C.stream_ptr(dev))
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
    dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
    lib = _C.lib()
    ws = _workspace(dev, int(lib.lsi_bn_workspace_floats(npix, c, bf16, groups)))
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
    rc = lib.lsi_bn_relu_bwd(_C.ptr(x), _C.ptr(dy), _C.ptr(mean_rstd),
                             _C.ptr(beta_f), _C.ptr(dx), _C.ptr(dbeta), _C.ptr(ws),
                             npix, c, bf16, ctx.relu, groups, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_bn_relu_bwd')
    return dx, dbeta, None, None, None


def batch_norm_relu(x, beta, eps=1e-3, relu=True, groups=1):
  """relu(batch_norm(x) + beta) with batch statistics (slim.batch_norm,
  scale=False) for a channels-last N x C x H x W tensor; same dtype out.
  groups > 1: N is that many sub-batches, each normalised on its own."""
  return _BnRelu.apply(x, beta, eps, relu, groups)
