"""Bilinear sampling / splatting (mirror of the reference's
lsi/geometry/sampling.py), executed by HIP kernels in liblsi_hip.so.

Coordinates are (x, y) with pixel centres at +0.5; points outside the grid
contribute / sample zero.
"""
import torch

from lsi import _C


def _dims(b, hs, ws, c, ht, wt):
  return [int(b), int(hs), int(ws), int(c), int(ht), int(wt)]


class _Bilinear(torch.autograd.Function):
  """lsi_bilinear_fwd / lsi_bilinear_bwd."""

  @staticmethod
  def forward(ctx, imgs, coords):
    dev = _C.require_device(imgs, coords)
    imgs, coords = imgs.contiguous(), coords.contiguous()
    b, hs, ws, c = imgs.shape
    _, ht, wt, _ = coords.shape
    out = torch.empty((b, ht, wt, c), dtype=torch.float32, device=dev)
    rc = _C.lib().lsi_bilinear_fwd(*_dims(b, hs, ws, c, ht, wt), _C.ptr(imgs),
                                   _C.ptr(coords), _C.ptr(out),
                                   _C.stream_ptr(dev))
    _C.check(rc, 'lsi_bilinear_fwd')
    ctx.save_for_backward(imgs, coords)
    return out

  @staticmethod
  def backward(ctx, g_out):
    imgs, coords = ctx.saved_tensors
    dev = imgs.device
    b, hs, ws, c = imgs.shape
    _, ht, wt, _ = coords.shape
    g_out = g_out.contiguous()
    g_imgs = torch.zeros_like(imgs) if ctx.needs_input_grad[0] else None
    g_coords = torch.empty_like(coords) if ctx.needs_input_grad[1] else None
    rc = _C.lib().lsi_bilinear_bwd(*_dims(b, hs, ws, c, ht, wt), _C.ptr(imgs),
                                   _C.ptr(coords), _C.ptr(g_out),
                                   _C.ptr(g_imgs), _C.ptr(g_coords),
                                   _C.stream_ptr(dev))
    _C.check(rc, 'lsi_bilinear_bwd')
    return g_imgs, g_coords


def bilinear(imgs, coords, compose=True):
  """Bilinear sampling (reference sampling.py:41-132).

  Args:
    imgs: B x H_s x W_s x C
    coords: B x H_t x W_t x 2, source pixel to copy from
    compose: False returns ([4 masked taps], [4 weights]) like the reference.
  Returns:
    B x H_t x W_t x C; coordinates outside the image sample 0.
  """
  if not compose:
    return _bilinear_taps(imgs, coords)
  return _Bilinear.apply(imgs, coords)


class _BilinearTaps(torch.autograd.Function):
  """lsi_bilinear_taps / lsi_bilinear_taps_bwd: taps [4,B,Ht,Wt,C] and weights
  [4,B,Ht,Wt,1] of sampling.bilinear(compose=False)."""

  @staticmethod
  def forward(ctx, imgs, coords):
    dev = _C.require_device(imgs, coords)
    imgs, coords = imgs.contiguous(), coords.contiguous()
    b, hs, ws, c = imgs.shape
    _, ht, wt, _ = coords.shape
    taps = torch.empty((4, b, ht, wt, c), dtype=torch.float32, device=dev)
    wts = torch.empty((4, b, ht, wt, 1), dtype=torch.float32, device=dev)
    rc = _C.lib().lsi_bilinear_taps(*_dims(b, hs, ws, c, ht, wt), _C.ptr(imgs),
                                    _C.ptr(coords), _C.ptr(taps), _C.ptr(wts),
                                    _C.stream_ptr(dev))
    _C.check(rc, 'lsi_bilinear_taps')
    ctx.save_for_backward(coords)
    ctx.img_shape = tuple(imgs.shape)
    return taps, wts

  @staticmethod
  def backward(ctx, g_taps, g_wts):
    coords, = ctx.saved_tensors
    dev = coords.device
    b, hs, ws, c = ctx.img_shape
    _, ht, wt, _ = coords.shape
    g_imgs = g_coords = None
    if ctx.needs_input_grad[0]:
      g_imgs = torch.zeros(ctx.img_shape, dtype=torch.float32, device=dev)
    if ctx.needs_input_grad[1]:
      g_coords = torch.empty_like(coords)
    if g_imgs is not None or g_coords is not None:
      g_taps = None if g_taps is None else g_taps.contiguous()
      g_wts = None if g_wts is None else g_wts.contiguous()
      rc = _C.lib().lsi_bilinear_taps_bwd(
          *_dims(b, hs, ws, c, ht, wt), _C.ptr(coords),
          _C.ptr(g_taps) if g_imgs is not None else None, _C.ptr(g_wts),
          _C.ptr(g_imgs), _C.ptr(g_coords), _C.stream_ptr(dev))
      _C.check(rc, 'lsi_bilinear_taps_bwd')
    return g_imgs, g_coords


def _bilinear_taps(imgs, coords):
  """compose=False (reference sampling.py:124-130): the four border-masked taps
  and the four un-masked weights, tap order (x0,y0), (x0,y1), (x1,y0), (x1,y1)
  -- lsi_bilinear_taps; differentiable as TF differentiates it
  (lsi_bilinear_taps_bwd: the taps scatter their gradient into the image, the
  weights give the coordinates' gradient)."""
  taps, wts = _BilinearTaps.apply(imgs, coords)
  return list(taps.unbind(0)), list(wts.unbind(0))


def bilinear_wrapper(imgs, coords, compose=True):
  """bilinear for arbitrary leading dims (reference sampling.py:135-168).

  imgs: [...] x H_s x W_s x C, coords: [...] x H_t x W_t x 2.
  """
  init_dims = list(imgs.shape[:-3])
  out = bilinear(imgs.reshape([-1] + list(imgs.shape[-3:])),
                 coords.reshape([-1] + list(coords.shape[-3:])), compose=compose)
  if compose:
    return out.reshape(init_dims + list(out.shape[-3:]))
  ims, wts = out
  return ([t.reshape(init_dims + list(t.shape[-3:])) for t in ims],
          [t.reshape(init_dims + list(t.shape[-3:])) for t in wts])


class _Splat(torch.autograd.Function):
  """lsi_splat_generic / lsi_splat_generic_bwd."""

  @staticmethod
  def forward(ctx, src_image, tgt_coords, init_trg_image):
    dev = _C.require_device(src_image, tgt_coords, init_trg_image)
    src, coords = src_image.contiguous(), tgt_coords.contiguous()
    b, hs, ws, c = src.shape
    _, ht, wt, _ = init_trg_image.shape
    out = init_trg_image.contiguous().clone()
    rc = _C.lib().lsi_splat_generic(*_dims(b, hs, ws, c, ht, wt), _C.ptr(src),
                                    _C.ptr(coords), _C.ptr(out),
                                    _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_generic')
    ctx.save_for_backward(src, coords)
    ctx.trg_hw = (ht, wt)
    return out

  @staticmethod
  def backward(ctx, g_out):
    src, coords = ctx.saved_tensors
    dev = src.device
    b, hs, ws, c = src.shape
    ht, wt = ctx.trg_hw
    g_out = g_out.contiguous()
    g_src = torch.empty_like(src) if ctx.needs_input_grad[0] else None
    g_coords = torch.empty_like(coords) if ctx.needs_input_grad[1] else None
    if g_src is not None or g_coords is not None:
      rc = _C.lib().lsi_splat_generic_bwd(
          *_dims(b, hs, ws, c, ht, wt), _C.ptr(src), _C.ptr(coords),
          _C.ptr(g_out), _C.ptr(g_src), _C.ptr(g_coords), _C.stream_ptr(dev))
      _C.check(rc, 'lsi_splat_generic_bwd')
    return g_src, g_coords, g_out


def splat(src_image, tgt_coords, init_trg_image):
  """Splat pixels of src_image to target coordinates (reference
  sampling.py:171-254).

  Args:
    src_image: [batch, height_s, width_s, channels]
    tgt_coords: [batch, height_s, width_s, 2]
    init_trg_image: [batch, height_t, width_t, channels]
  Returns:
    A new target image.
  """
  return _Splat.apply(src_image, tgt_coords, init_trg_image)


class _BatchScatterAdd(torch.autograd.Function):
  """lsi_scatter_add."""

  @staticmethod
  def forward(ctx, init, indices, updates):
    dev = _C.require_device(init, updates)
    if not indices.is_cuda:
      raise RuntimeError('indices must be on the ROCm device')
    b, p = init.shape
    _, n = updates.shape
    idx = indices.to(torch.int32).contiguous()
    if n > 0:
      lo, hi = int(idx.min()), int(idx.max())
      if lo < 0 or hi >= p:
        raise IndexError('scatter index out of range [0, %d): %d..%d' %
                         (p, lo, hi))
    out = init.contiguous().clone()
    upd = updates.contiguous()
    rc = _C.lib().lsi_scatter_add(int(b), int(p), int(n), _C.ptr(idx),
                                  _C.ptr(upd), _C.ptr(out), _C.stream_ptr(dev))
    _C.check(rc, 'lsi_scatter_add')
    ctx.save_for_backward(idx)
    return out

  @staticmethod
  def backward(ctx, g_out):
    (idx,) = ctx.saved_tensors
    g_upd = torch.gather(g_out, 1, idx.long())
    return g_out, None, g_upd


def scatter_add_tensor(init, indices, updates):
  """init + scatter of updates at indices into init's first dimension;
  duplicates add (reference sampling.py:257-284).  init: [P], indices: [N] or
  [N, 1], updates: [N]."""
  idx = indices.reshape(1, -1)
  return _BatchScatterAdd.apply(init.reshape(1, -1), idx,
                                updates.reshape(1, -1)).reshape(init.shape)


def batch_scatter_add_tensor(init, indices, updates):
  """scatter_add_tensor per batch row (reference sampling.py:287-313).
  init: [batch, #points], indices/updates: [batch, #updates]."""
  return _BatchScatterAdd.apply(init, indices, updates)
