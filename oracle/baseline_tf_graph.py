"""CPU baseline A: the reference's TF1 graph decomposition, op for op, on
torch-CPU (fp32).

TEST INFRASTRUCTURE ONLY: imported by bench.py's `cpu_baseline` leg and by
tests/.  TensorFlow 1.4 is not installable here, so "the
reference's CPU path" is timed through this stand-in, which executes the SAME
op sequence the reference's graph would -- every elementwise op its own pass
over full tensors, and per layer 5 channels x 4 corners `scatter_nd` into a
FRESH zero canvas followed by an `add` (sampling.py:246-252, 257-313) -- on a
multi-threaded fp32 tensor library.  It is deliberately not fused: the fused C
port is baseline B (oracle/lsi_ref_cpu.c).

Restates (paths relative to /root/reference):
  lsi/nnutils/helpers.py:82-85     divide_safe
  lsi/nnutils/helpers.py:88-113    pixel_coords (materialised constant)
  lsi/nnutils/helpers.py:116-137   transform_pts (batched matmul)
  lsi/nnutils/helpers.py:180-193   zbuffer_weights
  lsi/geometry/sampling.py:171-254 splat
  lsi/geometry/sampling.py:257-313 (batch_)scatter_add_tensor
  lsi/geometry/ldi.py:71-182       forward_splat
Values agree with oracle/lsi_oracle.py to fp32 tolerance
(tests/test_oracle_golden.py); index arithmetic is the same fp32 sequence.
"""
import torch


def divide_safe(num, den):
  """helpers.py:82-85."""
  eps = 1e-8
  den = den + eps * torch.eq(den, 0).to(torch.float32)
  return torch.div(num, den)


def pixel_coords(bs, h, w):
  """helpers.py:88-113: bs x h x w x 3 grid of (x+0.5, y+0.5, 1)."""
  ys = (torch.arange(1, h + 1, dtype=torch.float32) - 0.5).view(h, 1)
  xs = (torch.arange(1, w + 1, dtype=torch.float32) - 0.5).view(1, w)
  grid = torch.stack([xs.expand(h, w), ys.expand(h, w), torch.ones(h, w)], -1)
  return grid.unsqueeze(0).repeat(bs, 1, 1, 1)


def zbuffer_weights(disps, scale):
  """helpers.py:180-193."""
  pos = torch.gt(disps, 0).to(torch.float32)
  d = torch.clamp(disps, 0, 1)
  d = d - 0.5
  return torch.exp(d * scale) * pos


def transform_pts(pts, mat):
  """helpers.py:116-137: pts B x H x W x 4, mat B x 4 x 4 -> B x H x W x 4."""
  b, h, w, d = pts.shape
  flat = pts.reshape(b, h * w, d)
  out = torch.matmul(flat, mat.transpose(1, 2))
  return out.reshape(b, h, w, d)


def batch_scatter_add_tensor(init, indices, updates):
  """sampling.py:287-313 + 257-284: offset the indices per batch row, scatter
  into a fresh zero canvas, add to init."""
  b, p = init.shape
  offs = (torch.arange(b, dtype=torch.int64) * p).view(b, 1)
  flat_idx = (indices.to(torch.int64) + offs).reshape(-1)
  canvas = torch.zeros(b * p, dtype=torch.float32)          # tf.scatter_nd
  canvas.index_add_(0, flat_idx, updates.reshape(-1))
  return torch.add(init.reshape(-1), canvas).reshape(b, p)  # tf.add


def splat(src_image, tgt_coords, init_trg_image):
  """sampling.py:171-254."""
  b, hs, ws, c = src_image.shape
  _, ht, wt, _ = init_trg_image.shape
  n = hs * ws
  coords = tgt_coords - 0.5
  x = coords[..., 0].reshape(b, n)
  y = coords[..., 1].reshape(b, n)
  x0 = torch.floor(x)
  x1 = x0 + 1
  y0 = torch.floor(y)
  y1 = y0 + 1
  y_max, x_max = float(ht - 1), float(wt - 1)
  x0_safe = torch.clamp(x0, 0, x_max)
  y0_safe = torch.clamp(y0, 0, y_max)
  x1_safe = torch.clamp(x1, 0, x_max)
  y1_safe = torch.clamp(y1, 0, y_max)
  wt_x0 = (x1 - x) * torch.eq(x0, x0_safe).to(torch.float32)
  wt_x1 = (x - x0) * torch.eq(x1, x1_safe).to(torch.float32)
  wt_y0 = (y1 - y) * torch.eq(y0, y0_safe).to(torch.float32)
  wt_y1 = (y - y0) * torch.eq(y1, y1_safe).to(torch.float32)
  wt_tl = wt_x0 * wt_y0
  wt_tr = wt_x1 * wt_y0
  wt_bl = wt_x0 * wt_y1
  wt_br = wt_x1 * wt_y1
  eps = 1e-3
  wt_tl = wt_tl * torch.gt(wt_tl, eps).to(torch.float32)
  wt_tr = wt_tr * torch.gt(wt_tr, eps).to(torch.float32)
  wt_bl = wt_bl * torch.gt(wt_bl, eps).to(torch.float32)
  wt_br = wt_br * torch.gt(wt_br, eps).to(torch.float32)
  src_flat = src_image.reshape(b, n, c)
  values_tl = src_flat * wt_tl.unsqueeze(-1)
  values_tr = src_flat * wt_tr.unsqueeze(-1)
  values_bl = src_flat * wt_bl.unsqueeze(-1)
  values_br = src_flat * wt_br.unsqueeze(-1)
  # a non-finite coordinate has all four weights zero or NaN; the build drops
  # such points, the cast below needs a finite value
  fwt = float(wt)
  def idx(xs_, ys_):
    v = xs_ + ys_ * fwt
    return torch.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0).to(torch.int32)
  inds_tl = idx(x0_safe, y0_safe)
  inds_tr = idx(x1_safe, y0_safe)
  inds_bl = idx(x0_safe, y1_safe)
  inds_br = idx(x1_safe, y1_safe)
  init_flat = init_trg_image.reshape(b, ht * wt, c)
  channels = []
  for ch in range(c):
    cur = init_flat[:, :, ch]
    cur = batch_scatter_add_tensor(cur, inds_tl, values_tl[:, :, ch])
    cur = batch_scatter_add_tensor(cur, inds_tr, values_tr[:, :, ch])
    cur = batch_scatter_add_tensor(cur, inds_bl, values_bl[:, :, ch])
    cur = batch_scatter_add_tensor(cur, inds_br, values_br[:, :, ch])
    channels.append(cur)
  return torch.stack(channels, -1).reshape(b, ht, wt, c)


def forward_splat(tex, mask, disp, mat, trg_downsampling, bg_layer_disp,
                  max_disp, zbuf_scale, compose_layers=True,
                  compute_trg_disp=False):
  """ldi.py:71-182.  tex L x B x H x W x 3, disp/mask L x B x H x W x 1 (mask
  None = ones), mat B x 4 x 4.  Returns img, wts, disp (disp None unless
  compute_trg_disp: a TF session prunes the disparity splat when it is not
  fetched, so it is not executed here either)."""
  nl, b, h, w, _ = tex.shape
  s = float(trg_downsampling)
  ht, wt = int(h * s), int(w * s)
  pc = pixel_coords(b, h, w)
  bg_wt = float(zbuffer_weights(torch.tensor(bg_layer_disp / max_disp),
                                zbuf_scale))
  imgs, wtss, dsps = [], [], []
  for l in range(nl):
    pts = torch.cat([pc, disp[l]], -1)                       # ldi.py:134
    q = transform_pts(pts, mat)                              # ldi.py:135
    trg_coords = divide_safe(q[..., 0:2], q[..., 2:3])       # ldi.py:138
    trg_coords = trg_coords * s
    trg_disp = divide_safe(q[..., 3:4], q[..., 2:3])         # ldi.py:140
    wts_src = zbuffer_weights(trg_disp / max_disp, zbuf_scale)
    if mask is not None:
      wts_src = wts_src * mask[l]
    init_img = torch.ones(b, ht, wt, 3) * bg_wt              # ldi.py:122-125
    init_wts = torch.ones(b, ht, wt, 1) * bg_wt
    init_dsp = torch.zeros(b, ht, wt, 1) * bg_wt
    imgs.append(splat(tex[l] * wts_src, trg_coords, init_img))
    wtss.append(splat(wts_src, trg_coords, init_wts))
    if compute_trg_disp:
      dsps.append(splat(trg_disp * wts_src, trg_coords, init_dsp))
  img, wts = torch.stack(imgs), torch.stack(wtss)
  dsp = None
  if compute_trg_disp:
    dsp = divide_safe(torch.stack(dsps), wts)                # ldi.py:165
  if compose_layers:                                         # ldi.py:167-171
    img = torch.sum(img, 0, keepdim=True)
    wts = torch.sum(wts, 0, keepdim=True)
    if compute_trg_disp:
      dsp = torch.max(dsp, 0, keepdim=True)[0]
  img = divide_safe(img, wts)                                # ldi.py:174
  return img, wts, dsp
