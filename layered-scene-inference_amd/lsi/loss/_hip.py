"""torch.autograd bindings of the fused loss kernels (include/lsi_hip.h,
csrc/lsi_loss.hip).  Used for tensors on a ROCm device; there is no fallback:
a missing library raises in lsi._C.lib()."""
import ctypes

import torch

from lsi import _C


def _workspace(dev):
  n = int(_C.lib().lsi_loss_workspace_bytes())
  return torch.empty((n,), dtype=torch.uint8, device=dev), n


def _f32(t):
  return t if t.dtype == torch.float32 else t.float()


class _ZbufComp(torch.autograd.Function):
  """lsi_zbuf_comp_loss_fwd / _bwd (reference loss.py:66-115)."""

  @staticmethod
  def forward(ctx, imgs, masks, disps, trg, bg_layer_disp, max_disp, zbuf_scale):
    dev = _C.require_device(imgs, masks, disps, trg)
    nl, b, h, w, c = imgs.shape
    if c != 3:
      raise ValueError('zbuffer_composition_loss: 3-channel images (got %d)' % c)
    d = _C.LsiLossDesc()
    d.L, d.B, d.H, d.W = nl, b, h, w
    d.img_sl, d.img_sb, d.img_sy, d.img_sx, d.img_sc = imgs.stride()
    d.disp_sl, d.disp_sb, d.disp_sy, d.disp_sx = disps.stride()[:4]
    if masks is not None:
      d.mask_sl, d.mask_sb, d.mask_sy, d.mask_sx = masks.stride()[:4]
    d.trg_sb, d.trg_sy, d.trg_sx, d.trg_sc = trg.stride()
    d.bg_layer_disp, d.max_disp, d.zbuf_scale = (float(bg_layer_disp),
                                                 float(max_disp),
                                                 float(zbuf_scale))
    out = torch.empty((), dtype=torch.float32, device=dev)
    ws, n = _workspace(dev)
    rc = _C.lib().lsi_zbuf_comp_loss_fwd(
        ctypes.byref(d), _C.ptr(imgs), _C.ptr(masks), _C.ptr(disps), _C.ptr(trg),
        _C.ptr(out), _C.ptr(ws), n, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_zbuf_comp_loss_fwd')
    ctx.desc = d
    ctx.has_mask = masks is not None
    ctx.save_for_backward(imgs, masks if masks is not None else imgs.new_empty(0),
                          disps, trg)
    return out

  @staticmethod
  def backward(ctx, g):
    imgs, masks, disps, trg = ctx.saved_tensors
    masks = masks if ctx.has_mask else None
    dev = imgs.device
    d = ctx.desc
    g = g.contiguous().float()
    g_imgs = torch.empty((d.L, d.B, d.H, d.W, 3), dtype=torch.float32, device=dev)
    g_disps = torch.empty((d.L, d.B, d.H, d.W, 1), dtype=torch.float32, device=dev)
    g_masks = (torch.empty((d.L, d.B, d.H, d.W, 1), dtype=torch.float32,
                           device=dev) if masks is not None else None)
    rc = _C.lib().lsi_zbuf_comp_loss_bwd(
        ctypes.byref(d), _C.ptr(imgs), _C.ptr(masks), _C.ptr(disps), _C.ptr(trg),
        _C.ptr(g), _C.ptr(g_imgs), _C.ptr(g_masks), _C.ptr(g_disps),
        _C.stream_ptr(dev))
    _C.check(rc, 'lsi_zbuf_comp_loss_bwd')
    return g_imgs, g_masks, g_disps, None, None, None, None


def zbuffer_composition_loss(layer_imgs, layer_masks, layer_disps, trg_imgs,
                             bg_layer_disp, max_disp, zbuf_scale):
  # the target images are data (ldi_enc_dec.py:269-294): the kernels produce no
  # gradient for them, and silently dropping one would be wrong
  if torch.is_tensor(trg_imgs) and trg_imgs.requires_grad:
    raise RuntimeError('zbuffer_composition_loss: trg_imgs is not '
                       'differentiable on the HIP path (pass trg_imgs.detach())')
  return _ZbufComp.apply(_f32(layer_imgs),
                         None if layer_masks is None else _f32(layer_masks),
                         _f32(layer_disps), _f32(trg_imgs), bg_layer_disp,
                         max_disp, zbuf_scale)


class _DispReg(torch.autograd.Function):
  """lsi_disp_reg_loss_fwd / _bwd: (smoothness, decreasing) in one read of the
  disparities (reference ldi.py:33-68, loss.py:48-63)."""

  @staticmethod
  def forward(ctx, disp):
    dev = _C.require_device(disp)
    nl, b, h, w = disp.shape[:4]
    out2 = torch.empty((2,), dtype=torch.float32, device=dev)
    ws, n = _workspace(dev)
    st = disp.stride()[:4]
    rc = _C.lib().lsi_disp_reg_loss_fwd(nl, b, h, w, st[0], st[1], st[2], st[3],
                                        _C.ptr(disp), _C.ptr(out2), _C.ptr(ws),
                                        n, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_disp_reg_loss_fwd')
    ctx.save_for_backward(disp)
    return out2

  @staticmethod
  def backward(ctx, g2):
    disp, = ctx.saved_tensors
    dev = disp.device
    nl, b, h, w = disp.shape[:4]
    g2 = g2.contiguous().float()
    g_disp = torch.empty((nl, b, h, w, 1), dtype=torch.float32, device=dev)
    st = disp.stride()[:4]
    rc = _C.lib().lsi_disp_reg_loss_bwd(nl, b, h, w, st[0], st[1], st[2], st[3],
                                        _C.ptr(disp), _C.ptr(g2), _C.ptr(g_disp),
                                        _C.stream_ptr(dev))
    _C.check(rc, 'lsi_disp_reg_loss_bwd')
    return g_disp.reshape(disp.shape)


def disp_regularisers(pred_disp):
  """(disp_smoothness_loss, decreasing_disp_loss) of L x B x H x W x 1."""
  out2 = _DispReg.apply(_f32(pred_disp))
  return out2[0], out2[1]


class _ViewSynth(torch.autograd.Function):
  """lsi_view_synth_loss_fwd / _bwd (reference ldi_enc_dec.py:337-357)."""

  @staticmethod
  def forward(ctx, recons, target, x_min, y_min):
    dev = _C.require_device(recons, target)
    recons = recons.contiguous()
    nl, b, ht, wt, c = recons.shape
    _, h, w, _ = target.shape
    if c != 3:
      raise ValueError('view_synthesis_loss: 3-channel images (got %d)' % c)
    out = torch.empty((), dtype=torch.float32, device=dev)
    ws, n = _workspace(dev)
    ts = target.stride()
    rc = _C.lib().lsi_view_synth_loss_fwd(
        nl, b, ht, wt, h, w, x_min, y_min, _C.ptr(recons), _C.ptr(target),
        ts[0], ts[1], ts[2], ts[3], _C.ptr(out), _C.ptr(ws), n,
        _C.stream_ptr(dev))
    _C.check(rc, 'lsi_view_synth_loss_fwd')
    ctx.save_for_backward(recons, target)
    ctx.crop = (x_min, y_min)
    return out

  @staticmethod
  def backward(ctx, g):
    recons, target = ctx.saved_tensors
    dev = recons.device
    nl, b, ht, wt, _ = recons.shape
    _, h, w, _ = target.shape
    g = g.contiguous().float()
    g_recons = torch.empty_like(recons)
    ts = target.stride()
    rc = _C.lib().lsi_view_synth_loss_bwd(
        nl, b, ht, wt, h, w, ctx.crop[0], ctx.crop[1], _C.ptr(recons),
        _C.ptr(target), ts[0], ts[1], ts[2], ts[3], _C.ptr(g), _C.ptr(g_recons),
        _C.stream_ptr(dev))
    _C.check(rc, 'lsi_view_synth_loss_bwd')
    return g_recons, None, None, None


def view_synthesis_loss(recons_splat, to_recons_img, x_min, y_min):
  return _ViewSynth.apply(_f32(recons_splat), _f32(to_recons_img), int(x_min),
                          int(y_min))


def compose(imgs, masks, dmaps, soft, min_disp, depth_softmax_temp):
  """lsi_compose_fwd (reference layers.py:29-70); forward only (the reference
  uses it in its data renderer, never under a gradient)."""
  dev = _C.require_device(imgs, masks, dmaps)
  if imgs.requires_grad or masks.requires_grad or dmaps.requires_grad:
    raise RuntimeError('layers.compose on the GPU is forward-only')
  nl, c = imgs.shape[0], imgs.shape[-1]
  lead = tuple(imgs.shape[1:-1])
  imgs_c = _f32(imgs).reshape(nl, -1, c).contiguous()
  masks_c = _f32(masks).reshape(nl, -1).contiguous()
  dmaps_c = _f32(dmaps).reshape(nl, -1).contiguous()
  n = imgs_c.shape[1]
  out = torch.empty((n, c), dtype=torch.float32, device=dev)
  rc = _C.lib().lsi_compose_fwd(nl, n, c, _C.ptr(imgs_c), _C.ptr(masks_c),
                                _C.ptr(dmaps_c), int(bool(soft)), float(min_disp),
                                float(depth_softmax_temp), _C.ptr(out),
                                _C.stream_ptr(dev))
  _C.check(rc, 'lsi_compose_fwd')
  return out.reshape(lead + (c,))


def compose_depth(masks, dmaps, bg_layer, min_disp, depth_softmax_temp):
  """lsi_compose_depth_fwd (reference layers.py:73-115); forward only."""
  dev = _C.require_device(masks, dmaps)
  if masks.requires_grad or dmaps.requires_grad:
    raise RuntimeError('layers.compose_depth on the GPU is forward-only')
  nl = masks.shape[0]
  lead = tuple(masks.shape[1:-1])
  masks_c = _f32(masks).reshape(nl, -1).contiguous()
  dmaps_c = _f32(dmaps).reshape(nl, -1).contiguous()
  n = masks_c.shape[1]
  # tf.reduce_max over the relu'd maps with the background layer appended
  dmax = 0.0
  if bg_layer:
    dmax = max(float(torch.relu(dmaps_c).max()), float(min_disp))
  out = torch.empty((n,), dtype=torch.float32, device=dev)
  rc = _C.lib().lsi_compose_depth_fwd(nl, n, _C.ptr(masks_c), _C.ptr(dmaps_c),
                                      int(bool(bg_layer)), dmax, float(min_disp),
                                      float(depth_softmax_temp), _C.ptr(out),
                                      _C.stream_ptr(dev))
  _C.check(rc, 'lsi_compose_depth_fwd')
  return out.reshape(lead + (1,))
