"""CPU (NumPy, float32) restatement of the reference's LDI renderer hot path.

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module, and only as the checker.
The product path (`layered-scene-inference_amd/`) never imports it and has no
CPU fallback.

Every function cites the reference lines it restates (paths relative to
/root/reference).  The structure is ours (fused per-pixel formulation, explicit
op order) -- it is a restatement of the algorithm, not a copy of the source.

Parity status: PINNED against golden vectors under `tests/golden/` that were
produced by executing the reference's unchanged modules on the NumPy TF-1.4
stand-in (`oracle/tf1_numpy_shim.py`, `oracle/make_goldens.py`): projected
pixel indices bit-exact, values to <=1e-6 relative (tests/test_oracle_golden.py).
UNPINNED below that: TensorFlow 1.4's own kernels (absent third-party
dependency; the reference ships no tests or golden vectors of its own --
SURVEY.md section 4), covered by the fp32 tolerances stated in the tests.

Floating-point contract shared by this file, `oracle/lsi_ref_cpu.c` and the HIP
kernels (this is what makes the pixel indices bit-exact):
  q_j = ((x*M[j,0] + y*M[j,1]) + 1*M[j,2]) + d*M[j,3]   products and sums
        individually rounded to fp32, no FMA, in this order;
  n'  = n + 1e-8f*[n==0];  u = (q0/n')*s;  v = (q1/n')*s;  D = q3/n'   IEEE div;
  X = u - 0.5f;  x0 = floor(X);  x1 = x0 + 1;  clip to [0, Wt-1];
  idx = int32(x_safe + y_safe*Wt)   computed in fp32, then truncated.
"""
import math

import numpy as np

F = np.float32
EPS = F(1e-8)


def _f32(x):
  return np.asarray(x, dtype=np.float32)


# ---------------------------------------------------------------------------
# lsi/nnutils/helpers.py
# ---------------------------------------------------------------------------
def divide_safe(num, den):
  """helpers.py:82-85 -- num / (den + 1e-8*[den == 0])."""
  num, den = _f32(num), _f32(den)
  den = den + EPS * (den == 0).astype(np.float32)
  with np.errstate(all='ignore'):
    return (num / den).astype(np.float32)


def pixel_coords(bs, h, w):
  """helpers.py:88-113 -- bs x h x w x 3 grid of (x+0.5, y+0.5, 1)."""
  ys = (np.arange(1, h + 1, dtype=np.float32) - F(0.5))[:, None]
  xs = (np.arange(1, w + 1, dtype=np.float32) - F(0.5))[None, :]
  grid = np.stack(
      [np.broadcast_to(xs, (h, w)), np.broadcast_to(ys, (h, w)),
       np.ones((h, w), np.float32)], axis=-1)
  return np.broadcast_to(grid, (bs, h, w, 3)).copy()


def zbuffer_weights(disps, scale=50):
  """helpers.py:180-193 -- exp((clip(x,0,1) - 0.5)*scale) * [x > 0]."""
  x = _f32(disps)
  pos = (x > 0).astype(np.float32)
  x = np.minimum(np.maximum(x, F(0)), F(1))
  x = x - F(0.5)
  with np.errstate(all='ignore'):
    return (np.exp(x * F(scale)) * pos).astype(np.float32)


def matmul_seq(a, b):
  """tf.matmul as read by the oracle: sequential k, no FMA (helpers.py:135,
  projection.py:83-86)."""
  a, b = _f32(a), _f32(b)
  out = a[..., :, 0:1] * b[..., 0:1, :]
  for k in range(1, a.shape[-1]):
    out = out + a[..., :, k:k + 1] * b[..., k:k + 1, :]
  return out.astype(np.float32)


def transform_pts(pts, mat):
  """helpers.py:116-137 -- per-pixel q = M p (as [.., HW, D] @ M^T)."""
  pts, mat = _f32(pts), _f32(mat)
  d = mat.shape[-1]
  lead = mat.shape[:-2]
  flat = pts.reshape(lead + (-1, d))
  out = matmul_seq(flat, np.swapaxes(mat, -1, -2))
  return out.reshape(pts.shape)


def soft_z_buffering(layer_masks, layer_disps, depth_softmax_temp=1):
  """helpers.py:140-160 -- softmax over layers of log(mask+eps) - depth/temp."""
  m, d = _f32(layer_masks), _f32(layer_disps)
  d = np.maximum(d, F(0))
  depth = divide_safe(F(1), d)
  logp = -depth / F(depth_softmax_temp)
  with np.errstate(all='ignore'):
    logp = np.log(m + EPS) + logp
    logp = logp - np.max(logp, axis=0, keepdims=True)
    p = np.exp(logp)
    return (p / np.sum(p, axis=0, keepdims=True)).astype(np.float32)


def enforce_bg_occupied(ldi_masks):
  """helpers.py:163-177 -- last layer's mask forced to 1."""
  m = _f32(ldi_masks).copy()
  m[-1] = m[-1] * F(0) + F(1)
  return m


# ---------------------------------------------------------------------------
# lsi/geometry/projection.py
# ---------------------------------------------------------------------------
def pad_intrinsic(k):
  """projection.py:27-46."""
  k = _f32(k)
  out = np.zeros(k.shape[:-2] + (4, 4), np.float32)
  out[..., :3, :3] = k
  out[..., 3, 3] = 1
  return out


def pad_extrinsic(rot, trans):
  """projection.py:49-68."""
  rot, trans = _f32(rot), _f32(trans)
  out = np.zeros(rot.shape[:-2] + (4, 4), np.float32)
  out[..., :3, :3] = rot
  out[..., :3, 3:4] = trans
  out[..., 3, 3] = 1
  return out


def forward_projection_matrix(k_s, k_t, rot, t):
  """projection.py:71-86 -- pad(K_t) [R t; 0 1] pad(K_s^-1)."""
  k_s_inv = np.linalg.inv(_f32(k_s)).astype(np.float32)
  return matmul_seq(pad_intrinsic(k_t),
                    matmul_seq(pad_extrinsic(rot, t), pad_intrinsic(k_s_inv)))


def inverse_projection_matrix(k_s, k_t, rot, t):
  """projection.py:89-106 -- pad(K_s) [R^T  -R^T t; 0 1] pad(K_t^-1)."""
  k_t_inv = np.linalg.inv(_f32(k_t)).astype(np.float32)
  rot_inv = np.swapaxes(_f32(rot), -1, -2)
  t_inv = F(-1) * matmul_seq(rot_inv, t)
  return matmul_seq(pad_intrinsic(k_s),
                    matmul_seq(pad_extrinsic(rot_inv, t_inv),
                               pad_intrinsic(k_t_inv)))


def project(mat, disp, trg_downsampling=1.0, coords=None):
  """ldi.py:134-140 fused: every source pixel centre (x+.5, y+.5, 1, d) through
  M; returns (u, v, D) with u, v already scaled by trg_downsampling.

  mat: B x 4 x 4, disp: B x H x W (fp32).  coords: B x H x W x 3, the
  `pixel_coords_src` of ldi.py:134 when it is not the pixel-centre grid.
  """
  mat, disp = _f32(mat), _f32(disp)
  b, h, w = disp.shape
  if coords is None:
    xs = (np.arange(w, dtype=np.float32) + F(0.5))[None, None, :]
    ys = (np.arange(h, dtype=np.float32) + F(0.5))[None, :, None]
    ones = F(1)
  else:
    coords = _f32(coords)
    xs, ys, ones = coords[..., 0], coords[..., 1], coords[..., 2]
  m = mat[:, :, :, None, None]  # B x 4 x 4 x 1 x 1

  def row(j):
    acc = xs * m[:, j, 0] + ys * m[:, j, 1]
    acc = acc + ones * m[:, j, 2]
    acc = acc + disp * m[:, j, 3]
    return acc.astype(np.float32)

  q0, q1, n, q3 = row(0), row(1), row(2), row(3)
  s = F(trg_downsampling)
  u = divide_safe(q0, n) * s
  v = divide_safe(q1, n) * s
  dd = divide_safe(q3, n)
  return u.astype(np.float32), v.astype(np.float32), dd


def decisions_are_robust(mat, disp, trg_downsampling, h_trg, w_trg, max_disp):
  """For comparisons of fp32 results with an fp64 evaluation of the same op
  graph (gradient checks): True where a source pixel's discrete decisions --
  floor of the target coordinate and the border masks (sampling.py:189-211),
  the 1e-3 clamp of its four corner weights (:218-222), the z-buffer weight's
  `x > 0` and clip (helpers.py:187-191) -- come out the same when the graph is
  evaluated in fp32 (the reference's arithmetic) and in fp64 (the gradient
  oracle's).  Where they differ the two legitimately disagree by a whole
  corner's contribution.

  mat B x 4 x 4, disp B x H x W; returns bool B x H x W."""

  def decide(dt):
    m = np.asarray(mat, dt)[:, :, :, None, None]
    d = np.asarray(disp, np.float32).astype(dt)
    b, h, w = d.shape
    xs = (np.arange(w, dtype=dt) + dt(0.5))[None, None, :]
    ys = (np.arange(h, dtype=dt) + dt(0.5))[None, :, None]
    with np.errstate(all='ignore'):
      q = [((xs * m[:, j, 0] + ys * m[:, j, 1]) + m[:, j, 2]) + d * m[:, j, 3]
           for j in range(4)]
      n = q[2] + dt(1e-8) * (q[2] == 0)
      s = dt(trg_downsampling)
      x, y = q[0] / n * s - dt(0.5), q[1] / n * s - dt(0.5)
      z = q[3] / n / dt(max_disp)
      x0, y0 = np.floor(x), np.floor(y)
      bits = [x0, y0, (z > 0), (z >= 0) & (z <= 1)]
      wx = [(x0 + 1 - x) * ((x0 >= 0) & (x0 <= w_trg - 1)),
            (x - x0) * ((x0 + 1 >= 0) & (x0 + 1 <= w_trg - 1))]
      wy = [(y0 + 1 - y) * ((y0 >= 0) & (y0 <= h_trg - 1)),
            (y - y0) * ((y0 + 1 >= 0) & (y0 + 1 <= h_trg - 1))]
      for a in wx:
        for c in wy:
          bits.append(a * c > dt(1e-3))
    return bits

  lo, hi = decide(np.float32), decide(np.float64)
  ok = np.ones(np.asarray(disp).shape, bool)
  for a, c in zip(lo, hi):
    with np.errstate(all='ignore'):
      ok &= (np.asarray(a, np.float64) == np.asarray(c, np.float64))
  return ok


# ---------------------------------------------------------------------------
# lsi/geometry/sampling.py
# ---------------------------------------------------------------------------
def splat_corners(u, v, h_trg, w_trg):
  """sampling.py:183-241 -- the four bilinear splat corners of every point.

  Returns idx4 (int32, [...,4] order tl,tr,bl,br; flat x + y*w_trg within one
  batch element) and w4 (float32, [...,4]) after border masking and the 1e-3
  clamp.
  """
  x = _f32(u) - F(0.5)
  y = _f32(v) - F(0.5)
  x0 = np.floor(x)
  x1 = x0 + F(1)
  y0 = np.floor(y)
  y1 = y0 + F(1)
  x_max, y_max = F(w_trg) - F(1), F(h_trg) - F(1)
  x0s = np.minimum(np.maximum(x0, F(0)), x_max)
  x1s = np.minimum(np.maximum(x1, F(0)), x_max)
  y0s = np.minimum(np.maximum(y0, F(0)), y_max)
  y1s = np.minimum(np.maximum(y1, F(0)), y_max)
  with np.errstate(all='ignore'):
    wx0 = (x1 - x) * (x0 == x0s).astype(np.float32)
    wx1 = (x - x0) * (x1 == x1s).astype(np.float32)
    wy0 = (y1 - y) * (y0 == y0s).astype(np.float32)
    wy1 = (y - y0) * (y1 == y1s).astype(np.float32)
    ws = [wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1]
    ws = [wk * (wk > F(1e-3)).astype(np.float32) for wk in ws]
    wt = F(w_trg)
    ids = [x0s + y0s * wt, x1s + y0s * wt, x0s + y1s * wt, x1s + y1s * wt]
    # NaN coordinates: TF's float->int cast of NaN is implementation defined;
    # the weights are NaN there too.  The build defines "NaN coordinate => the
    # point is dropped" and so does the oracle.
    bad = ~(np.isfinite(x) & np.isfinite(y))
    ids = [np.where(bad, F(0), i) for i in ids]
    ws = [np.where(bad, F(0), wk) for wk in ws]
    idx4 = np.stack([np.trunc(i).astype(np.int32) for i in ids], axis=-1)
  w4 = np.stack(ws, axis=-1).astype(np.float32)
  return idx4, w4


def scatter_add_tensor(init, indices, updates):
  """sampling.py:257-284 -- init + scatter_nd(indices, updates); dups add."""
  init = _f32(init)
  scat = np.zeros_like(init)
  np.add.at(scat, np.asarray(indices).reshape(-1), _f32(updates).reshape(-1))
  return init + scat


def batch_scatter_add_tensor(init, indices, updates):
  """sampling.py:287-313 -- per batch row scatter-add on [B, P]."""
  init = _f32(init)
  b, p = init.shape
  idx = np.asarray(indices, dtype=np.int64) + (np.arange(b) * p)[:, None]
  return scatter_add_tensor(init.reshape(-1), idx, updates).reshape(b, p)


def splat(src_image, tgt_coords, init_trg_image):
  """sampling.py:171-254 -- 4-corner bilinear forward scatter-add.

  src_image B x Hs x Ws x C, tgt_coords B x Hs x Ws x 2 (x, y), init B x Ht x Wt x C.
  Corner canvases are accumulated separately and added in tl,tr,bl,br order,
  like the reference's four scatter_nd + add per channel (sampling.py:246-252).
  """
  src, init = _f32(src_image), _f32(init_trg_image)
  coords = _f32(tgt_coords)
  b, _, _, c = src.shape
  _, ht, wt, _ = init.shape
  idx4, w4 = splat_corners(coords[..., 0], coords[..., 1], ht, wt)
  idx4 = idx4.reshape(b, -1, 4)
  w4 = w4.reshape(b, -1, 4)
  srcf = src.reshape(b, -1, c)
  out = init.reshape(b, ht * wt, c).copy()
  for ch in range(c):
    cur = out[:, :, ch]
    for k in range(4):
      cur = batch_scatter_add_tensor(cur, idx4[:, :, k],
                                     srcf[:, :, ch] * w4[:, :, k])
    out[:, :, ch] = cur
  return out.reshape(b, ht, wt, c)


def bilinear(imgs, coords):
  """sampling.py:41-132 (compose=True) -- 4-tap gather, zero outside."""
  imgs, coords = _f32(imgs), _f32(coords)
  b, hs, ws, c = imgs.shape
  x = coords[..., 0:1] - F(0.5)
  y = coords[..., 1:2] - F(0.5)
  x0 = np.floor(x)
  x1 = x0 + F(1)
  y0 = np.floor(y)
  y1 = y0 + F(1)
  x_max, y_max = F(ws - 1), F(hs - 1)
  x0s = np.minimum(np.maximum(x0, F(0)), x_max)
  x1s = np.minimum(np.maximum(x1, F(0)), x_max)
  y0s = np.minimum(np.maximum(y0, F(0)), y_max)
  y1s = np.minimum(np.maximum(y1, F(0)), y_max)
  with np.errstate(all='ignore'):
    wx0, wx1 = x1 - x, x - x0
    wy0, wy1 = y1 - y, y - y0
    vx0 = (x0 == x0s).astype(np.float32)
    vx1 = (x1 == x1s).astype(np.float32)
    vy0 = (y0 == y0s).astype(np.float32)
    vy1 = (y1 == y1s).astype(np.float32)
    bad = ~(np.isfinite(x) & np.isfinite(y))
    flat = imgs.reshape(b, hs * ws, c)
    bidx = np.arange(b).reshape((b,) + (1,) * (coords.ndim - 1))

    def tap(xs_, ys_):
      idx = np.where(bad, F(0), xs_ + ys_ * F(ws))
      idx = np.trunc(idx).astype(np.int64)[..., 0]
      return flat[bidx[..., 0], idx]

    im00, im01 = tap(x0s, y0s), tap(x0s, y1s)
    im10, im11 = tap(x1s, y0s), tap(x1s, y1s)
    out = vx0 * vy0 * wx0 * wy0 * im00
    out = out + vx0 * vy1 * wx0 * wy1 * im01
    out = out + vx1 * vy0 * wx1 * wy0 * im10
    out = out + vx1 * vy1 * wx1 * wy1 * im11
    out = np.where(bad, F(0), out)
  return out.astype(np.float32)


def bilinear_taps(imgs, coords):
  """sampling.py:41-132 with compose=False: the four border-masked taps
  (order 00, 01, 10, 11 = (x0,y0), (x0,y1), (x1,y0), (x1,y1)) and the four
  un-masked weights."""
  imgs, coords = _f32(imgs), _f32(coords)
  b, hs, ws, c = imgs.shape
  x = coords[..., 0:1] - F(0.5)
  y = coords[..., 1:2] - F(0.5)
  x0 = np.floor(x); x1 = x0 + F(1); y0 = np.floor(y); y1 = y0 + F(1)
  x_max, y_max = F(ws - 1), F(hs - 1)
  x0s = np.minimum(np.maximum(x0, F(0)), x_max)
  x1s = np.minimum(np.maximum(x1, F(0)), x_max)
  y0s = np.minimum(np.maximum(y0, F(0)), y_max)
  y1s = np.minimum(np.maximum(y1, F(0)), y_max)
  vx0 = (x0 == x0s).astype(np.float32); vx1 = (x1 == x1s).astype(np.float32)
  vy0 = (y0 == y0s).astype(np.float32); vy1 = (y1 == y1s).astype(np.float32)
  flat = imgs.reshape(b, hs * ws, c)
  bidx = np.arange(b).reshape((b,) + (1,) * (coords.ndim - 2))

  def tap(xs_, ys_):
    idx = np.trunc(xs_ + ys_ * F(ws)).astype(np.int64)[..., 0]
    return flat[bidx, idx]

  ims = [vx0 * vy0 * tap(x0s, y0s), vx0 * vy1 * tap(x0s, y1s),
         vx1 * vy0 * tap(x1s, y0s), vx1 * vy1 * tap(x1s, y1s)]
  wts = [(x1 - x) * (y1 - y), (x1 - x) * (y - y0), (x - x0) * (y1 - y),
         (x - x0) * (y - y0)]
  return ims, wts


def bilinear_wrapper(imgs, coords):
  """sampling.py:135-168 -- arbitrary leading dims."""
  imgs, coords = _f32(imgs), _f32(coords)
  lead = imgs.shape[:-3]
  out = bilinear(imgs.reshape((-1,) + imgs.shape[-3:]),
                 coords.reshape((-1,) + coords.shape[-3:]))
  return out.reshape(lead + out.shape[-3:])


# ---------------------------------------------------------------------------
# lsi/geometry/ldi.py
# ---------------------------------------------------------------------------
def forward_splat(tex, mask, disp, mat, trg_downsampling=1, bg_layer_disp=0,
                  max_disp=1, zbuf_scale=10, compose_layers=True,
                  debug=False, coords=None):
  """ldi.py:71-182 with the projection matrix passed in as data (coords: the
  caller's `pixel_coords_src`, B x H x W x 3; None = the pixel-centre grid).

  tex L x B x H x W x 3, mask/disp L x B x H x W x 1, mat B x 4 x 4.
  Returns dict: img [nl,B,Ht,Wt,3], wts [nl,B,Ht,Wt,1], disp [nl,B,Ht,Wt,1]
  (+ canvases / idx4 / upd4 when debug).
  """
  tex, mask, disp, mat = _f32(tex), _f32(mask), _f32(disp), _f32(mat)
  nl, b, h, w, c = tex.shape
  ht, wt = h * trg_downsampling, w * trg_downsampling
  assert ht == int(ht) and wt == int(wt), 'Ht, Wt must be integral'
  ht, wt = int(ht), int(wt)
  bg_wt = zbuffer_weights(F(bg_layer_disp / max_disp), zbuf_scale)  # :115-116
  cimg = np.empty((nl, b, ht, wt, c), np.float32)
  cwts = np.empty((nl, b, ht, wt, 1), np.float32)
  cdsp = np.empty((nl, b, ht, wt, 1), np.float32)
  idx_all, upd_all = [], []
  for l in range(nl):
    u, v, dd = project(mat, disp[l, ..., 0], trg_downsampling, coords)  # :134-140
    pw = zbuffer_weights(dd / F(max_disp), zbuf_scale) * mask[l, ..., 0]  # :145
    uv = np.stack([u, v], axis=-1)
    img0 = np.ones((b, ht, wt, c), np.float32) * bg_wt  # :123-125
    wts0 = np.ones((b, ht, wt, 1), np.float32) * bg_wt
    dsp0 = np.zeros((b, ht, wt, 1), np.float32) * bg_wt
    cimg[l] = splat(tex[l] * pw[..., None], uv, img0)  # :148-155
    cwts[l] = splat(pw[..., None], uv, wts0)
    # build definition (NaN / Inf inputs, where TF's behaviour is implementation
    # defined): a pixel with zero weight adds nothing to the disparity canvas
    # either, although dd * 0 would be NaN for a non-finite dd
    with np.errstate(all='ignore'):
      keep = (pw != 0) & np.isfinite(u) & np.isfinite(v)
      dterm = np.where(keep, dd * pw, F(0)).astype(np.float32)
    cdsp[l] = splat(dterm[..., None], uv, dsp0)
    if debug:
      idx4, w4 = splat_corners(u, v, ht, wt)
      idx_all.append(idx4.reshape(b, h * w, 4))
      upd_all.append((pw[..., None] * w4).reshape(b, h * w, 4))
  out_disp = divide_safe(cdsp, cwts)  # :165
  out_img, out_wts = cimg, cwts
  if compose_layers:  # :167-171
    out_img = np.sum(cimg, axis=0, keepdims=True)
    out_wts = np.sum(cwts, axis=0, keepdims=True)
    out_disp = np.max(out_disp, axis=0, keepdims=True)
  out_img = divide_safe(out_img, out_wts)  # :173
  res = {'img': out_img, 'wts': out_wts, 'disp': out_disp}
  if debug:
    res.update(canvas_img=cimg, canvas_wts=cwts, canvas_disp=cdsp,
               idx4=np.stack(idx_all), upd4=np.stack(upd_all), bg_wt=bg_wt)
  return res


def gradient(pred):
  """ldi.py:33-44."""
  pred = _f32(pred)
  dy = pred[:, :, 1:, :, :] - pred[:, :, :-1, :, :]
  dx = pred[:, :, :, 1:, :] - pred[:, :, :, :-1, :]
  return dx, dy


def disp_smoothness_loss(pred_disp):
  """ldi.py:47-68 -- sum of mean |second differences|."""
  dx, dy = gradient(pred_disp)
  dx2, dxdy = gradient(dx)
  dydx, dy2 = gradient(dy)
  return F(np.mean(np.abs(dx2)) + np.mean(np.abs(dxdy)) +
           np.mean(np.abs(dydx)) + np.mean(np.abs(dy2)))


# ---------------------------------------------------------------------------
# lsi/geometry/projection.py: disocclusion mask
# ---------------------------------------------------------------------------
def disocclusion_mask(disps_src, disps_trg, src2trg_mat, thresh=1e-2):
  """projection.py:109-150.  disps B x H x W x 1 -> mask B x H x W x 1."""
  ds, dt = _f32(disps_src), _f32(disps_trg)
  _, h_t, w_t, _ = dt.shape
  u, v, d2 = project(src2trg_mat, ds[..., 0], 1.0)
  # project() multiplies by s = 1.0f, which is exact.
  trunc = ((u > w_t).astype(np.float32) + (v > h_t).astype(np.float32) +
           (u < 0).astype(np.float32) + (v < 0).astype(np.float32))
  trunc = (trunc > 0).astype(np.float32)
  samp = bilinear(dt, np.stack([u, v], axis=-1))[..., 0]
  with np.errstate(all='ignore'):
    dis = (np.abs(d2 - samp) > F(thresh)).astype(np.float32)
  return ((F(1) - trunc) * dis)[..., None].astype(np.float32)


# ---------------------------------------------------------------------------
# lsi/geometry/layers.py and homography.py
# ---------------------------------------------------------------------------
def compose(imgs, masks, dmaps, soft=False, min_disp=1e-6,
            depth_softmax_temp=1):
  """layers.py:29-70."""
  imgs, masks, dmaps = _f32(imgs), _f32(masks), _f32(dmaps)
  nl = imgs.shape[0]
  dmaps = np.maximum(dmaps, F(0))
  imgs = np.concatenate([imgs, np.ones_like(imgs[:1])], 0)
  masks = np.concatenate([masks, np.ones_like(masks[:1])], 0)
  dmaps = np.concatenate([dmaps, np.ones_like(dmaps[:1]) * F(min_disp)], 0)
  sel = soft_z_buffering(masks, dmaps, depth_softmax_temp)
  if not soft:
    am = np.argmax(sel, axis=0)
    sel = np.moveaxis((am[..., None] == np.arange(nl + 1)).astype(np.float32),
                      -1, 0)
  return np.sum(sel * imgs, axis=0).astype(np.float32)


def compose_depth(masks, dmaps, bg_layer=False, min_disp=1e-6,
                  depth_softmax_temp=1):
  """layers.py:73-115."""
  masks, dmaps = _f32(masks), _f32(dmaps)
  nl = masks.shape[0]
  dmaps = np.maximum(dmaps, F(0))
  bg_disp = np.ones_like(dmaps[:1]) * F(min_disp)
  masks = np.concatenate([masks, np.ones_like(masks[:1])], 0)
  dmaps = np.concatenate([dmaps, bg_disp], 0)
  if bg_layer:
    dsel = np.max(dmaps) - dmaps[0:nl]
    dsel = np.concatenate([dsel, bg_disp], 0)
  else:
    dsel = dmaps
  sel = soft_z_buffering(masks, dsel, depth_softmax_temp)
  am = np.argmax(sel, axis=0)
  sel = np.moveaxis((am[..., None] == np.arange(nl + 1)).astype(np.float32),
                    -1, 0)
  return np.sum(sel * dmaps, axis=0).astype(np.float32)


def inv_homography(k_s, k_t, rot, t, n_hat, a):
  """homography.py:28-51 -- K_s (R^T + R^T t n R^T / (a - n R^T t)) K_t^-1."""
  rot_t = np.swapaxes(_f32(rot), -1, -2)
  k_t_inv = np.linalg.inv(_f32(k_t)).astype(np.float32)
  denom = _f32(a) - matmul_seq(matmul_seq(n_hat, rot_t), t)
  numer = matmul_seq(matmul_seq(matmul_seq(rot_t, t), n_hat), rot_t)
  return matmul_seq(matmul_seq(k_s, rot_t + divide_safe(numer, denom)),
                    k_t_inv)


def inv_homography_dmat(k_t, rot, t, n_hat, a):
  """homography.py:54-73."""
  rot_t = np.swapaxes(_f32(rot), -1, -2)
  k_t_inv = np.linalg.inv(_f32(k_t)).astype(np.float32)
  denom = _f32(a) - matmul_seq(matmul_seq(n_hat, rot_t), t)
  return divide_safe(F(-1) * matmul_seq(matmul_seq(n_hat, rot_t), k_t_inv),
                     denom)


def normalize_homogeneous(pts):
  """homography.py:76-92."""
  pts = _f32(pts)
  return divide_safe(pts[..., :-1], pts[..., -1:])


def transform_plane_imgs(imgs, pixel_coords_trg, k_s, k_t, rot, t, n_hat, a):
  """homography.py:95-117 -- homography warp = transform_pts + bilinear."""
  hom = inv_homography(k_s, k_t, rot, t, n_hat, a)
  pts = transform_pts(pixel_coords_trg, hom)
  return bilinear_wrapper(imgs, normalize_homogeneous(pts))


def transform_plane_eqns(rot, t, n_hat, a):
  """homography.py:120-136."""
  rot_t = np.swapaxes(_f32(rot), -1, -2)
  n_hat_t = matmul_seq(n_hat, rot_t)
  a_t = _f32(a) - matmul_seq(n_hat, matmul_seq(rot_t, t))
  return n_hat_t, a_t


def trg_disp_maps(pixel_coords_trg, k_t, rot, t, n_hat, a):
  """homography.py:139-156."""
  dm = inv_homography_dmat(k_t, rot, t, n_hat, a)  # [...] x 1 x 3
  prod = np.expand_dims(dm, -2) * _f32(pixel_coords_trg)
  return np.sum(prod, axis=-1, keepdims=True).astype(np.float32)


def planar_transform(imgs, masks, pixel_coords_trg, k_s, k_t, rot, t, n_hat,
                     a):
  """layers.py:118-162."""
  imgs, masks = _f32(imgs), _f32(masks)
  nl = imgs.shape[0]

  def rep(x):
    x = _f32(x)
    return np.broadcast_to(x[None], (nl,) + x.shape).copy()

  k_s, k_t, t, rot = rep(k_s), rep(k_t), rep(t), rep(rot)
  pct = rep(pixel_coords_trg)
  both = np.concatenate([imgs, masks], axis=-1)
  warped = transform_plane_imgs(both, pct, k_s, k_t, rot, t, n_hat, a)
  dmaps = trg_disp_maps(pct, k_t, rot, t, n_hat, a)
  return warped[..., :3], warped[..., 3:4], dmaps


# ---------------------------------------------------------------------------
# lsi/loss/loss.py and the inline losses of ldi_enc_dec.py
# ---------------------------------------------------------------------------
def decreasing_disp_loss(layer_disps):
  """loss.py:48-63."""
  d = _f32(layer_disps)
  if d.shape[0] == 1:
    return F(0)
  return F(np.mean(np.maximum(d[1:] - d[:-1], F(0))))


def zbuffer_composition_loss(layer_imgs, layer_masks, layer_disps, trg_imgs,
                             bg_layer_disp=0, max_disp=1, zbuf_scale=10):
  """loss.py:66-115."""
  imgs, masks, disps = _f32(layer_imgs), _f32(layer_masks), _f32(layer_disps)
  imgs = np.concatenate([imgs, np.ones_like(imgs[:1])], 0)
  masks = np.concatenate([masks, np.ones_like(masks[:1])], 0)
  disps = np.concatenate([disps, np.ones_like(disps[:1]) * F(bg_layer_disp)],
                         0)
  probs = zbuffer_weights(disps / F(max_disp), zbuf_scale) * masks
  probs = divide_safe(probs, np.sum(probs, axis=0, keepdims=True))
  diff = imgs - _f32(trg_imgs)
  cost = np.sum(diff * diff * probs, axis=0)
  return F(F(0.5) * np.mean(cost))


def area_downsample(img, ht, wt):
  """tf.image.resize_images(..., AREA) for integer factors (ldi_enc_dec.py:
  337-340): exact box mean."""
  img = _f32(img)
  b, h, w, c = img.shape
  fy, fx = h // ht, w // wt
  assert fy * ht == h and fx * wt == w, 'integer AREA factors only'
  return img.reshape(b, ht, fy, wt, fx, c).mean(axis=(2, 4)).astype(np.float32)


def py2_round(x):
  """Python-2 round(): half away from zero (ldi_enc_dec.py:348-351)."""
  return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def view_synthesis_loss(recons_splat, to_recons_img, splat_bdry_ignore=0.05):
  """ldi_enc_dec.py:337-357 -- AREA-downsample target, L1, mean over C, min
  over layers, crop the border, mean."""
  r = _f32(recons_splat)
  _, _, ht, wt, _ = r.shape
  tgt = area_downsample(to_recons_img, ht, wt)
  pw = np.min(np.mean(np.abs(tgt[None] - r), axis=4), axis=0)
  x_min = py2_round(wt * splat_bdry_ignore)
  y_min = py2_round(ht * splat_bdry_ignore)
  pw = pw[:, y_min:ht - y_min, x_min:wt - x_min]
  return F(np.mean(pw))


# ---------------------------------------------------------------------------
# Evaluation metrics, ldi_pred_eval.py:297-548 (define_metrics): (sum, norm)
# pairs per metric for ONE direction / one pair of LDIs.
# ---------------------------------------------------------------------------
def eval_view_synthesis_metrics(recons_splat, recons_disp, to_recons_img,
                                splat_bdry_ignore, valid_mask=None,
                                disocc_mask=None, to_recons_disp=None):
  """ldi_pred_eval.py:385-460 for one rendered view: recons_splat 1 x B x Ht x
  Wt x 3 and recons_disp 1 x B x Ht x Wt x 1 from forward_splat(compose=True,
  compute_trg_disp=True).  valid_mask (B x H x W x 1, {0,1}) is thresholded at
  0.95 after the AREA resize (:398); the dis-occlusion mask's DOWNSAMPLED but
  un-thresholded values weight the restricted sums (:403-409, 446-451)."""
  r = _f32(recons_splat)
  _, b, ht, wt, _ = r.shape
  tgt = area_downsample(to_recons_img, ht, wt)
  if valid_mask is None:
    valid = np.ones((b, ht, wt), np.float32)
  else:
    valid = (area_downsample(valid_mask, ht, wt)[..., 0] > F(0.95)).astype(np.float32)
  x_min, y_min = py2_round(wt * splat_bdry_ignore), py2_round(ht * splat_bdry_ignore)
  centre = np.zeros((b, ht, wt), np.float32)
  centre[:, y_min:ht - y_min, x_min:wt - x_min] = 1
  centre = centre * valid
  pw = np.min(np.mean(np.abs(tgt[None] - r), axis=4), axis=0) * centre
  out = {'compose_splat_loss': (float(pw.sum(dtype=np.float64)),
                                float(centre.sum(dtype=np.float64)))}
  dm = None
  if disocc_mask is not None:
    dm = area_downsample(_f32(disocc_mask), ht, wt)[..., 0]
    out['compose_splat_loss_disocc'] = (float((pw * dm).sum(dtype=np.float64)),
                                        float((centre * dm).sum(dtype=np.float64)))
  if to_recons_disp is not None:
    gds = area_downsample(to_recons_disp, ht, wt)
    pd = np.min(np.mean(np.abs(gds[None] - _f32(recons_disp)), axis=4), axis=0) * centre
    out['depth_splat_loss'] = (float(pd.sum(dtype=np.float64)),
                               float(centre.sum(dtype=np.float64)))
    if dm is not None:
      out['depth_splat_loss_disocc'] = (float((pd * dm).sum(dtype=np.float64)),
                                        float((centre * dm).sum(dtype=np.float64)))
  return out


def eval_layer_prediction_metrics(ldi_src, ldi_trg, imgs_src, imgs_trg, gt,
                                  bg_layer_disp):
  """ldi_pred_eval.py:476-531: foreground layer (0) against the images / gt
  disparities where gt_disp > bg_layer_disp; background layer (last) against
  the gt background texture / disparity where the foreground hides it
  (gt_disp > gt_disp_bg).  Texture sums are divided by 3."""
  n = ldi_src[0].shape[0]
  f64 = lambda a: float(np.sum(a, dtype=np.float64))
  v_s = (_f32(gt['src_gt_disp']) > F(bg_layer_disp)).astype(np.float32)
  v_t = (_f32(gt['trg_gt_disp']) > F(bg_layer_disp)).astype(np.float32)
  out = {}
  n_fg = f64(v_s + v_t)
  out['fg_tex_error'] = (f64(np.abs(ldi_src[0][0] - imgs_src) * v_s) / 3 +
                         f64(np.abs(ldi_trg[0][0] - imgs_trg) * v_t) / 3, n_fg)
  out['fg_disp_error'] = (f64(np.abs(ldi_src[2][0] - gt['src_gt_disp']) * v_s) +
                          f64(np.abs(ldi_trg[2][0] - gt['trg_gt_disp']) * v_t), n_fg)
  b_s = (_f32(gt['src_gt_disp']) > _f32(gt['src_gt_disp_bg'])).astype(np.float32)
  b_t = (_f32(gt['trg_gt_disp']) > _f32(gt['trg_gt_disp_bg'])).astype(np.float32)
  n_bg = f64(b_s + b_t)
  out['bg_tex_error'] = (
      f64(np.abs(ldi_src[0][n - 1] - gt['src_gt_tex_bg']) * b_s) / 3 +
      f64(np.abs(ldi_trg[0][n - 1] - gt['trg_gt_tex_bg']) * b_t) / 3, n_bg)
  out['bg_disp_error'] = (
      f64(np.abs(ldi_src[2][n - 1] - gt['src_gt_disp_bg']) * b_s) +
      f64(np.abs(ldi_trg[2][n - 1] - gt['trg_gt_disp_bg']) * b_t), n_bg)
  return out
