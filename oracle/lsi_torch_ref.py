"""Differentiable torch (CPU, any float dtype) restatement of forward_splat,
splat and bilinear -- the gradient oracle.

TEST INFRASTRUCTURE ONLY.  The reference has no hand-written backward: TF
autodiff differentiates the op graph (train_utils.py:113).  This file restates
the same op graph (ldi.py:71-182, sampling.py:41-254, helpers.py) with torch
ops whose autograd conventions match TF's on the pieces that matter: floor and
comparisons have zero gradient, clamp passes gradient on the closed interval,
scatter-add is linear.  Gradients from it (in fp64) are what the HIP backward
kernels are checked against; it is itself checked against finite differences
(tests/test_oracle_golden.py).
"""
import torch


def divide_safe(num, den):
  den = den + 1e-8 * (den == 0).to(den.dtype)
  return num / den


def zbuffer_weights(x, scale):
  pos = (x > 0).to(x.dtype)
  # TF clip_by_value = minimum(maximum(x, lo), hi): gradient passes on [lo, hi].
  inside = ((x >= 0) & (x <= 1)).to(x.dtype)
  xc = x * inside + (x > 1).to(x.dtype)  # value of clamp, closed-interval grad
  return torch.exp((xc - 0.5) * scale) * pos


def corners(u, v, ht, wt):
  """sampling.py:183-241; returns idx [...,4] (long) and w [...,4]."""
  x, y = u - 0.5, v - 0.5
  x0, y0 = torch.floor(x).detach(), torch.floor(y).detach()
  x1, y1 = x0 + 1, y0 + 1
  x0s, x1s = x0.clamp(0, wt - 1), x1.clamp(0, wt - 1)
  y0s, y1s = y0.clamp(0, ht - 1), y1.clamp(0, ht - 1)
  dt = u.dtype
  wx0 = (x1 - x) * (x0 == x0s).to(dt)
  wx1 = (x - x0) * (x1 == x1s).to(dt)
  wy0 = (y1 - y) * (y0 == y0s).to(dt)
  wy1 = (y - y0) * (y1 == y1s).to(dt)
  ws = [wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1]
  ws = [w * (w > 1e-3).to(dt) for w in ws]
  ids = [x0s + y0s * wt, x1s + y0s * wt, x0s + y1s * wt, x1s + y1s * wt]
  return (torch.stack([i.long() for i in ids], -1), torch.stack(ws, -1))


def splat(src, coords, init):
  """sampling.py:171-254.  src B x Hs x Ws x C, coords B x Hs x Ws x 2."""
  b, _, _, c = src.shape
  _, ht, wt, _ = init.shape
  idx, w = corners(coords[..., 0], coords[..., 1], ht, wt)
  out = init.reshape(b, ht * wt, c)
  srcf = src.reshape(b, -1, c)
  idx = idx.reshape(b, -1, 4)
  w = w.reshape(b, -1, 4)
  res = []
  for bi in range(b):
    o = out[bi]
    for k in range(4):
      o = o.index_add(0, idx[bi, :, k], srcf[bi] * w[bi, :, k:k + 1])
    res.append(o)
  return torch.stack(res).reshape(b, ht, wt, c)


def forward_splat(tex, mask, disp, mat, s, bg_layer_disp, max_disp, zbuf_scale,
                  compose):
  """ldi.py:71-182.  tex L x B x H x W x 3, mask/disp L x B x H x W x 1,
  mat B x 4 x 4.  Returns img, wts, disp_out."""
  nl, b, h, w, c = tex.shape
  dt = tex.dtype
  ht, wt = int(h * s), int(w * s)
  bg = zbuffer_weights(torch.tensor(bg_layer_disp / max_disp, dtype=dt),
                       zbuf_scale)
  xs = (torch.arange(w, dtype=dt) + 0.5).view(1, 1, w).expand(b, h, w)
  ys = (torch.arange(h, dtype=dt) + 0.5).view(1, h, 1).expand(b, h, w)
  imgs, wtss, dsps = [], [], []
  for l in range(nl):
    p = torch.stack([xs, ys, torch.ones_like(xs), disp[l, ..., 0]], -1)
    q = torch.einsum('bhwk,bjk->bhwj', p, mat.to(dt))
    uv = divide_safe(q[..., 0:2], q[..., 2:3]) * s
    dd = divide_safe(q[..., 3:4], q[..., 2:3])
    pw = zbuffer_weights(dd / max_disp, zbuf_scale) * mask[l]
    ones3 = torch.ones((b, ht, wt, c), dtype=dt)
    ones1 = torch.ones((b, ht, wt, 1), dtype=dt)
    imgs.append(splat(tex[l] * pw, uv, ones3 * bg))
    wtss.append(splat(pw, uv, ones1 * bg))
    dsps.append(splat(dd * pw, uv, ones1 * 0))
  img, wts, dsp = torch.stack(imgs), torch.stack(wtss), torch.stack(dsps)
  dsp = divide_safe(dsp, wts)
  if compose:
    img = img.sum(0, keepdim=True)
    wts = wts.sum(0, keepdim=True)
    dsp = dsp.max(0, keepdim=True)[0]
  return divide_safe(img, wts), wts, dsp


def bilinear(imgs, coords):
  """sampling.py:41-132 (compose=True)."""
  b, hs, ws, c = imgs.shape
  dt = imgs.dtype
  x, y = coords[..., 0:1] - 0.5, coords[..., 1:2] - 0.5
  x0, y0 = torch.floor(x).detach(), torch.floor(y).detach()
  x1, y1 = x0 + 1, y0 + 1
  x0s, x1s = x0.clamp(0, ws - 1), x1.clamp(0, ws - 1)
  y0s, y1s = y0.clamp(0, hs - 1), y1.clamp(0, hs - 1)
  wx0, wx1, wy0, wy1 = x1 - x, x - x0, y1 - y, y - y0
  vx0, vx1 = (x0 == x0s).to(dt), (x1 == x1s).to(dt)
  vy0, vy1 = (y0 == y0s).to(dt), (y1 == y1s).to(dt)
  flat = imgs.reshape(b, hs * ws, c)

  def tap(xs_, ys_):
    idx = (xs_ + ys_ * ws).long()[..., 0]
    return torch.stack([flat[bi][idx[bi]] for bi in range(b)])

  out = vx0 * vy0 * wx0 * wy0 * tap(x0s, y0s)
  out = out + vx0 * vy1 * wx0 * wy1 * tap(x0s, y1s)
  out = out + vx1 * vy0 * wx1 * wy0 * tap(x1s, y0s)
  out = out + vx1 * vy1 * wx1 * wy1 * tap(x1s, y1s)
  return out


def bilinear_taps(imgs, coords):
  """sampling.py:124-130 (compose=False): ([4 border-masked taps], [4 weights]) in
  the reference's order (x0,y0), (x0,y1), (x1,y0), (x1,y1) -- the gradient
  oracle of lsi_bilinear_taps_bwd (same op graph: floor / clip / equal carry no
  gradient, gathers scatter theirs)."""
  b, hs, ws, c = imgs.shape
  dt = imgs.dtype
  x, y = coords[..., 0:1] - 0.5, coords[..., 1:2] - 0.5
  x0, y0 = torch.floor(x).detach(), torch.floor(y).detach()
  x1, y1 = x0 + 1, y0 + 1
  x0s, x1s = x0.clamp(0, ws - 1), x1.clamp(0, ws - 1)
  y0s, y1s = y0.clamp(0, hs - 1), y1.clamp(0, hs - 1)
  wx0, wx1, wy0, wy1 = x1 - x, x - x0, y1 - y, y - y0
  vx0, vx1 = (x0 == x0s).to(dt), (x1 == x1s).to(dt)
  vy0, vy1 = (y0 == y0s).to(dt), (y1 == y1s).to(dt)
  flat = imgs.reshape(b, hs * ws, c)

  def tap(xs_, ys_):
    idx = (xs_ + ys_ * ws).long()[..., 0]
    return torch.stack([flat[bi][idx[bi]] for bi in range(b)])

  ims = [vx0 * vy0 * tap(x0s, y0s), vx0 * vy1 * tap(x0s, y1s),
         vx1 * vy0 * tap(x1s, y0s), vx1 * vy1 * tap(x1s, y1s)]
  wts = [wx0 * wy0, wx0 * wy1, wx1 * wy0, wx1 * wy1]
  return ims, wts


# ---------------------------------------------------------------------------
# Losses and layer composition (gradient oracles of csrc/lsi_loss.hip).  Same
# op graphs as the reference, torch ops, any float dtype; their forward values
# are pinned by the reference-generated goldens (tests/test_oracle_golden.py).
# ---------------------------------------------------------------------------
def zbuffer_composition_loss(layer_imgs, layer_masks, layer_disps, trg_imgs,
                             bg_layer_disp=0, max_disp=1, zbuf_scale=10):
  """loss.py:66-115: white background layer at bg_layer_disp appended,
  p_l = zw(d_l / max_disp) * m_l / sum, 0.5 * mean(sum_l (img_l - trg)^2 p_l)."""
  layer_imgs = torch.cat([layer_imgs, torch.ones_like(layer_imgs[:1])], 0)
  layer_masks = torch.cat([layer_masks, torch.ones_like(layer_masks[:1])], 0)
  layer_disps = torch.cat(
      [layer_disps, torch.ones_like(layer_disps[:1]) * bg_layer_disp], 0)
  layer_probs = zbuffer_weights(layer_disps / max_disp, zbuf_scale) * layer_masks
  probs_sum = torch.sum(layer_probs, dim=0, keepdim=True)
  layer_probs = divide_safe(layer_probs, probs_sum)
  layerwise_cost = torch.square(layer_imgs - trg_imgs) * layer_probs
  return 0.5 * torch.sum(layerwise_cost, dim=0).mean()


def decreasing_disp_loss(layer_disps):
  """loss.py:48-63: mean(relu(d_{l+1} - stop_gradient(d_l))); 0 for L = 1."""
  n_layers = layer_disps.shape[0]
  if n_layers == 1:
    return 0
  return torch.relu(layer_disps[1:] - layer_disps[:-1].detach()).mean()


def gradient(pred):
  """ldi.py:33-44: forward differences along W (dx) and H (dy)."""
  dy = pred[:, :, 1:, :, :] - pred[:, :, :-1, :, :]
  dx = pred[:, :, :, 1:, :] - pred[:, :, :, :-1, :]
  return dx, dy


def disp_smoothness_loss(pred_disp):
  """ldi.py:47-68: mean |dxx| + mean |dxy| + mean |dyx| + mean |dyy|, each
  over its own (shrunken) shape; torch.abs has the TF gradient abs'(0) = 0."""
  dx, dy = gradient(pred_disp)
  dx2, dxdy = gradient(dx)
  dydx, dy2 = gradient(dy)
  return (dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() +
          dy2.abs().mean())


def area_downsample(img, ht, wt):
  """tf.image.resize_images(AREA) for integer factors: exact box mean."""
  b, h, w, c = img.shape
  fy, fx = h // ht, w // wt
  return img.reshape(b, ht, fy, wt, fx, c).mean(dim=(2, 4))


def py2_round(x):
  import math
  return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def view_synthesis_loss(recons_splat, to_recons_img, splat_bdry_ignore=0.05):
  """ldi_enc_dec.py:337-357: AREA-downsample the target, mean |diff| over
  channels, min over layers, crop round(size * f) border pixels (py2 round),
  mean.  (torch.min sends the gradient to one of several tied layers where TF
  splits it evenly: the tests account for that.)"""
  _, _, ht, wt, _ = recons_splat.shape
  tgt = area_downsample(to_recons_img, ht, wt)
  pw = torch.min(torch.mean(torch.abs(tgt.unsqueeze(0) - recons_splat), dim=4),
                 dim=0)[0]
  x_min, y_min = py2_round(wt * splat_bdry_ignore), py2_round(ht * splat_bdry_ignore)
  return pw[:, y_min:ht - y_min, x_min:wt - x_min].mean()


def soft_z_buffering(layer_masks, layer_disps, depth_softmax_temp=1):
  """helpers.py:140-160: p_l ~ (mask + 1e-8) * exp(-1 / (relu(d) * temp)),
  max-subtracted softmax over the layers."""
  eps = 1e-8
  layer_disps = torch.relu(layer_disps)
  layer_depths = divide_safe(torch.ones_like(layer_disps), layer_disps)
  log_depth_probs = -layer_depths / depth_softmax_temp
  log_layer_probs = torch.log(layer_masks + eps) + log_depth_probs
  log_layer_probs = log_layer_probs - torch.max(log_layer_probs, dim=0,
                                                keepdim=True)[0]
  layer_probs = torch.exp(log_layer_probs)
  return divide_safe(layer_probs, torch.sum(layer_probs, dim=0, keepdim=True))


def _one_hot_argmax(selection_mask, depth):
  idx = torch.argmax(selection_mask, dim=0)
  return torch.moveaxis(
      torch.nn.functional.one_hot(idx, depth).to(selection_mask.dtype), -1, 0)


def compose(imgs, masks, dmaps, soft=False, min_disp=1e-6, depth_softmax_temp=1):
  """layers.py:29-70."""
  n_layers = imgs.shape[0]
  dmaps = torch.relu(dmaps)
  imgs = torch.cat([imgs, torch.ones_like(imgs[:1])], 0)
  masks = torch.cat([masks, torch.ones_like(masks[:1])], 0)
  dmaps = torch.cat([dmaps, torch.ones_like(dmaps[:1]) * min_disp], 0)
  sel = soft_z_buffering(masks, dmaps, depth_softmax_temp)
  if not soft:
    sel = _one_hot_argmax(sel, n_layers + 1)
  return torch.sum(sel * imgs, dim=0)


def compose_depth(masks, dmaps, bg_layer=False, min_disp=1e-6,
                  depth_softmax_temp=1):
  """layers.py:73-115."""
  n_layers = masks.shape[0]
  dmaps = torch.relu(dmaps)
  bg_disp = torch.ones_like(dmaps[:1]) * min_disp
  masks = torch.cat([masks, torch.ones_like(masks[:1])], 0)
  dmaps = torch.cat([dmaps, bg_disp], 0)
  if bg_layer:
    sel_d = torch.cat([torch.max(dmaps) - dmaps[0:n_layers], bg_disp], 0)
  else:
    sel_d = dmaps
  sel = _one_hot_argmax(soft_z_buffering(masks, sel_d, depth_softmax_temp),
                        n_layers + 1)
  return torch.sum(sel * dmaps, dim=0)
