"""Generates tests/golden/*.npz by executing the reference's own modules.

TEST INFRASTRUCTURE ONLY.  Run in the build container, where /root/reference
exists:

    python oracle/make_goldens.py

The reference's source files are read and executed IN PLACE (never copied) on
the NumPy TF-1.4 stand-in `oracle/tf1_numpy_shim.py`; only inputs and outputs
(data) are written to `tests/golden/`.  The fixtures travel to the GPU box, this
script's dependency on /root/reference does not (no test imports this file).

Every fixture stores the inputs, the projection matrix the reference computed
(so every implementation consumes the same `M`), and the reference's outputs.
For forward_splat the per-pixel scatter indices / updates of the weight splat
are captured from the reference's own tf.scatter_nd calls (sampling.py:278).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tf1_numpy_shim as tf  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
T = tf.Tensor


def f32(x):
  return np.asarray(x, dtype=np.float32)


def smooth_noise(rs, shape, radius):
  """Box-blurred uniform noise along the two spatial axes (H, W = axes -2,-1)."""
  x = rs.rand(*shape).astype(np.float64)
  for ax in (-2, -1):
    acc = np.zeros_like(x)
    for d in range(-radius, radius + 1):
      acc += np.roll(x, d, axis=ax)
    x = acc / (2 * radius + 1)
  x = (x - x.min()) / (x.max() - x.min() + 1e-12)
  return x.astype(np.float32)


def rot_xyz(ax, ay, az):
  cx, sx, cy, sy, cz, sz = (np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay),
                            np.cos(az), np.sin(az))
  rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
  ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
  rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
  return (rz @ ry @ rx).astype(np.float32)


def kitti_cams(b, h, w, tx=-0.532):
  k = f32([[0.58 * w, 0, w / 2.0], [0, 0.58 * w, h / 2.0], [0, 0, 1]])
  k = np.broadcast_to(k, (b, 3, 3)).copy()
  rot = np.broadcast_to(np.eye(3, dtype=np.float32), (b, 3, 3)).copy()
  t = np.broadcast_to(f32([[tx], [0], [0]]), (b, 3, 1)).copy()
  return k, k.copy(), rot, t


def synth_cams(rs, b, h, w, ang=0.15, tr=0.4, tz=0.2):
  k = f32([[w, 0, w / 2.0], [0, h, h / 2.0], [0, 0, 1]])
  k = np.broadcast_to(k, (b, 3, 3)).copy()
  rot = np.stack([rot_xyz(*(rs.uniform(-ang, ang, 3))) for _ in range(b)])
  t = np.stack([
      f32([[rs.uniform(-tr, tr)], [rs.uniform(-tr, tr)],
           [rs.uniform(-tz, tz)]]) for _ in range(b)
  ])
  return k, k.copy(), f32(rot), f32(t)


def run_forward_splat(mods, name, tex, mask, disp, k_s, k_t, rot, t, s, bg,
                      max_disp, zbuf):
  ldi = mods['lsi.geometry.ldi']
  helpers = mods['lsi.nnutils.helpers']
  proj = mods['lsi.geometry.projection']
  nl, b, h, w, _ = tex.shape
  pc = helpers.pixel_coords(b, h, w)
  mat = proj.forward_projection_matrix(T(k_s), T(k_t), T(rot), T(t)).a
  out = {
      'tex': tex, 'mask': mask, 'disp': disp, 'k_s': k_s, 'k_t': k_t,
      'rot': rot, 't': t, 'M': mat,
      'params': np.array([s, bg, max_disp, zbuf], dtype=np.float64),
  }
  for compose in (True, False):
    tf.SCATTER_LOG = []
    img, wts, dsp = ldi.forward_splat(
        [T(tex), T(mask), T(disp)], pc, T(k_s), T(k_t), T(rot), T(t),
        compose_layers=compose, compute_trg_disp=True, trg_downsampling=s,
        bg_layer_disp=bg, max_disp=max_disp, zbuf_scale=zbuf)
    log = tf.SCATTER_LOG
    tf.SCATTER_LOG = None
    tag = 'compose' if compose else 'indep'
    out[tag + '_img'] = img.a
    out[tag + '_wts'] = wts.a
    out[tag + '_disp'] = dsp.a
    if compose:
      # 20 scatter_nd per layer: 12 (rgb) + 4 (weights) + 4 (disp); take the
      # weight splat's four corners tl,tr,bl,br.
      assert len(log) == 20 * nl, len(log)
      p = log[0][2] // b
      idx4 = np.zeros((nl, b, h * w, 4), np.int32)
      upd4 = np.zeros((nl, b, h * w, 4), np.float32)
      off = (np.arange(b, dtype=np.int32) * p)[:, None]
      for l in range(nl):
        for k in range(4):
          flat, upd, _ = log[20 * l + 12 + k]
          idx4[l, :, :, k] = flat.reshape(b, h * w) - off
          upd4[l, :, :, k] = upd.reshape(b, h * w)
      out['idx4'] = idx4
      out['upd4'] = upd4
  np.savez_compressed(os.path.join(OUT, 'fs_%s.npz' % name), **out)
  print('fs_%s' % name, 'img mean', out['compose_img'].mean())


def make_forward_splat(mods):
  rs = np.random.RandomState(1234)

  # BASELINE config 1: synthetic 1-layer 64x64 batch 2, s = 0.5.
  nl, b, h, w = 1, 2, 64, 64
  k_s, k_t, rot, t = synth_cams(rs, b, h, w)
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = (0.28 + 0.22 * smooth_noise(rs, (nl, b, h, w), 3))[..., None]
  mask = np.ones((nl, b, h, w, 1), np.float32)
  run_forward_splat(mods, 'cfg1_synth_L1_64', tex, mask, f32(disp), k_s, k_t,
                    rot, t, 0.5, 0.2, 1.0, 50)

  # KITTI-like rectified stereo, 2 layers, s = 0.5 (BASELINE config 2 shape,
  # shrunk).
  nl, b, h, w = 2, 2, 32, 96
  k_s, k_t, rot, t = kitti_cams(b, h, w)
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  sm = smooth_noise(rs, (nl, b, h, w), 4)
  scale = np.array([1.0, 0.5], np.float32).reshape(nl, 1, 1, 1)
  disp = (0.4 * sm * scale)[..., None]
  mask = np.ones((nl, b, h, w, 1), np.float32)
  run_forward_splat(mods, 'kitti_L2_s05', tex, mask, f32(disp), k_s, k_t, rot,
                    t, 0.5, 1e-3, 0.4, 50)

  # The trg->src direction used by the training loss (ldi_enc_dec.py:322-334):
  # swapped intrinsics, inverse pose; 1 layer, s = 1, iid disparities.
  nl, b, h, w = 1, 2, 24, 40
  k_s, k_t, rot, t = kitti_cams(b, h, w)
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = (0.4 * rs.rand(nl, b, h, w, 1)).astype(np.float32)
  mask = np.ones((nl, b, h, w, 1), np.float32)
  inv_rot = np.transpose(rot, (0, 2, 1)).copy()
  inv_t = -np.matmul(inv_rot, t)
  run_forward_splat(mods, 'kitti_inv_L1_s1', tex, mask, disp, k_t, k_s,
                    inv_rot, f32(inv_t), 1, 1e-3, 0.4, 50)

  # General pose (rotation + 3-D translation), 3 layers, soft masks.
  nl, b, h, w = 3, 2, 32, 32
  k_s, k_t, rot, t = synth_cams(rs, b, h, w, ang=0.2, tr=0.5, tz=0.3)
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = (0.2 + 0.5 * rs.rand(nl, b, h, w, 1)).astype(np.float32)
  mask = rs.rand(nl, b, h, w, 1).astype(np.float32)
  run_forward_splat(mods, 'general_L3_s05', tex, mask, disp, k_s, k_t, rot, t,
                    0.5, 0.2, 1.0, 50)

  # Edge cases: disparities <= 0 and > max_disp, points behind the camera
  # (normaliser <= 0), exact-zero normaliser, far out-of-image projections.
  nl, b, h, w = 2, 1, 16, 24
  k = f32([[w, 0, w / 2.0], [0, h, h / 2.0], [0, 0, 1]])[None]
  rot = rot_xyz(0.05, -0.1, 0.02)[None]
  t = f32([[0.3], [-0.2], [-2.0]])[None]
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = rs.uniform(-0.3, 1.6, (nl, b, h, w, 1)).astype(np.float32)
  disp[0, 0, 0, :4, 0] = 0.0
  disp[1, 0, 3, 5, 0] = 1e6
  disp[1, 0, 4, 5, 0] = -1e6
  mask = (rs.rand(nl, b, h, w, 1) > 0.2).astype(np.float32)
  run_forward_splat(mods, 'edge_L2_s05', tex, mask, disp, k, k.copy(), rot, t,
                    0.5, 0.0, 1.0, 10)


def make_known_answers(mods):
  samp = mods['lsi.geometry.sampling']
  helpers = mods['lsi.nnutils.helpers']
  out = {}
  one = T(np.ones((1, 1, 1, 1), np.float32))

  def one_px(x, y):
    return samp.splat(one, T(f32([[[[x, y]]]])),
                      T(np.zeros((1, 4, 4, 1), np.float32))).a[0, :, :, 0]

  pts = [(1.75, 2.25), (0.2, 0.5), (4.2, 3.5), (1.5005, 1.5), (-7.0, 1.5),
         (1e9, 1.5), (4.5, 1.5)]
  out['splat_pts'] = f32(pts)
  out['splat_canvases'] = np.stack([one_px(*p) for p in pts])
  two = T(np.ones((1, 1, 2, 1), np.float32))
  out['splat_two_same'] = samp.splat(
      two, T(f32([[[[2.5, 2.5], [2.5, 2.5]]]])),
      T(np.zeros((1, 4, 4, 1), np.float32))).a[0, :, :, 0]
  zin = f32([-.1, 0, .0025, .5, 1, 1.3])
  out['zbuf_in'] = zin
  out['zbuf_out_50'] = helpers.zbuffer_weights(T(zin), 50).a
  out['zbuf_out_10'] = helpers.zbuffer_weights(T(zin), 10).a
  out['bg_wt_kitti'] = helpers.zbuffer_weights(1e-3 / 0.4, scale=50).a
  out['bg_wt_synth'] = helpers.zbuffer_weights(2e-1 / 1.0, scale=50).a
  out['divsafe_num'] = f32([1, 1, -1, 0])
  out['divsafe_den'] = f32([0, 2, 0, 0])
  out['divsafe_out'] = helpers.divide_safe(T(out['divsafe_num']),
                                           T(out['divsafe_den'])).a
  out['pixel_coords_2_3_5'] = helpers.pixel_coords(2, 3, 5).a
  # scatter_add_tensor / batch_scatter_add_tensor
  rs = np.random.RandomState(7)
  init = rs.rand(3, 11).astype(np.float32)
  idx = rs.randint(0, 11, (3, 20)).astype(np.int32)
  upd = rs.rand(3, 20).astype(np.float32)
  out['bsa_init'], out['bsa_idx'], out['bsa_upd'] = init, idx, upd
  out['bsa_out'] = samp.batch_scatter_add_tensor(T(init), T(idx), T(upd)).a
  out['sa_out'] = samp.scatter_add_tensor(T(init[0]), T(idx[0][:, None]),
                                          T(upd[0])).a
  np.savez_compressed(os.path.join(OUT, 'known_answers.npz'), **out)
  print('known_answers', out['splat_canvases'][0])


def make_bilinear(mods):
  samp = mods['lsi.geometry.sampling']
  rs = np.random.RandomState(11)
  imgs = rs.rand(2, 9, 12, 3).astype(np.float32)
  coords = np.stack([rs.uniform(-2, 14, (2, 7, 8)),
                     rs.uniform(-2, 11, (2, 7, 8))], -1).astype(np.float32)
  coords[0, 0, 0] = [0.5, 0.5]
  coords[0, 0, 1] = [11.5, 8.5]
  coords[0, 0, 2] = [12.0, 9.0]
  coords[0, 0, 3] = [0.0, 0.0]
  out = {'imgs': imgs, 'coords': coords,
         'out': samp.bilinear(T(imgs), T(coords)).a}
  imgs5 = rs.rand(2, 3, 6, 7, 4).astype(np.float32)
  coords5 = np.stack([rs.uniform(-1, 8, (2, 3, 5, 4)),
                      rs.uniform(-1, 7, (2, 3, 5, 4))], -1).astype(np.float32)
  ims4, wts4 = samp.bilinear(T(imgs), T(coords), compose=False)
  out['taps_ims'] = np.stack([t.a for t in ims4])
  out['taps_wts'] = np.stack([t.a for t in wts4])
  out['imgs5'], out['coords5'] = imgs5, coords5
  out['out5'] = samp.bilinear_wrapper(T(imgs5), T(coords5)).a
  np.savez_compressed(os.path.join(OUT, 'bilinear.npz'), **out)
  print('bilinear', out['out'].mean())


def make_layers(mods):
  layers = mods['lsi.geometry.layers']
  hom = mods['lsi.geometry.homography']
  helpers = mods['lsi.nnutils.helpers']
  proj = mods['lsi.geometry.projection']
  rs = np.random.RandomState(21)
  nl, b, h, w = 3, 2, 10, 12
  imgs = rs.rand(nl, b, h, w, 3).astype(np.float32)
  masks = (rs.rand(nl, b, h, w, 1) > 0.4).astype(np.float32)
  dmaps = rs.uniform(-0.1, 1.0, (nl, b, h, w, 1)).astype(np.float32)
  out = {'imgs': imgs, 'masks': masks, 'dmaps': dmaps}
  out['compose_hard'] = layers.compose(T(imgs), T(masks), T(dmaps)).a
  out['compose_soft'] = layers.compose(T(imgs), T(masks), T(dmaps), soft=True,
                                       min_disp=1e-3,
                                       depth_softmax_temp=0.4).a
  out['compose_depth'] = layers.compose_depth(T(masks), T(dmaps)).a
  out['compose_depth_bg'] = layers.compose_depth(
      T(masks), T(dmaps), bg_layer=True, min_disp=1e-3,
      depth_softmax_temp=0.4).a
  out['soft_z'] = helpers.soft_z_buffering(T(masks), T(dmaps), 0.4).a
  out['enforce_bg'] = helpers.enforce_bg_occupied(T(masks)).a

  # planar_transform + the homography pieces: 2 planes, batch 2, 16x20.
  nl, b, h, w = 2, 2, 16, 20
  k_s, k_t, rot, t = synth_cams(rs, b, h, w, ang=0.1, tr=0.3, tz=0.1)
  pimgs = rs.rand(nl, b, h, w, 3).astype(np.float32)
  pmasks = (rs.rand(nl, b, h, w, 1) > 0.3).astype(np.float32)
  n_hat = rs.normal(size=(nl, b, 1, 3)).astype(np.float32)
  n_hat[..., 2] = np.abs(n_hat[..., 2]) + 2.0
  n_hat /= np.linalg.norm(n_hat, axis=-1, keepdims=True)
  a = rs.uniform(-3.5, -2.0, (nl, b, 1, 1)).astype(np.float32)
  pc = helpers.pixel_coords(b, h, w)
  ti, tm, td = layers.planar_transform(T(pimgs), T(pmasks), pc, T(k_s), T(k_t),
                                       T(rot), T(t), T(n_hat), T(a))
  out.update(p_imgs=pimgs, p_masks=pmasks, p_k_s=k_s, p_k_t=k_t, p_rot=rot,
             p_t=t, p_n_hat=n_hat, p_a=a, p_out_imgs=ti.a, p_out_masks=tm.a,
             p_out_dmaps=td.a)
  out['inv_hom'] = hom.inv_homography(T(k_s), T(k_t), T(rot), T(t),
                                      T(n_hat[0]), T(a[0])).a
  out['inv_hom_dmat'] = hom.inv_homography_dmat(T(k_t), T(rot), T(t),
                                                T(n_hat[0]), T(a[0])).a
  nt, at = hom.transform_plane_eqns(T(rot), T(t), T(n_hat[0]), T(a[0]))
  out['plane_n_t'], out['plane_a_t'] = nt.a, at.a
  out['fwd_mat'] = proj.forward_projection_matrix(T(k_s), T(k_t), T(rot),
                                                  T(t)).a
  out['inv_mat'] = proj.inverse_projection_matrix(T(k_s), T(k_t), T(rot),
                                                  T(t)).a
  np.savez_compressed(os.path.join(OUT, 'layers.npz'), **out)
  print('layers', out['compose_hard'].mean(), out['p_out_imgs'].mean())


def make_disocclusion(mods):
  proj = mods['lsi.geometry.projection']
  helpers = mods['lsi.nnutils.helpers']
  rs = np.random.RandomState(31)
  b, h, w = 2, 20, 28
  k_s, k_t, rot, t = synth_cams(rs, b, h, w, ang=0.08, tr=0.3, tz=0.1)
  ds = (0.3 + 0.2 * smooth_noise(rs, (b, h, w), 2))[..., None]
  dt = (0.3 + 0.2 * smooth_noise(rs, (b, h, w), 2))[..., None]
  mat = proj.forward_projection_matrix(T(k_s), T(k_t), T(rot), T(t))
  pc = helpers.pixel_coords(b, h, w)
  mask = proj.disocclusion_mask(T(f32(ds)), T(f32(dt)), pc, mat).a
  np.savez_compressed(os.path.join(OUT, 'disocclusion.npz'), disps_src=f32(ds),
                      disps_trg=f32(dt), M=mat.a, mask=mask)
  print('disocclusion', mask.mean())


def make_losses(mods):
  loss = mods['lsi.loss.loss']
  ldi = mods['lsi.geometry.ldi']
  rs = np.random.RandomState(41)
  nl, b, h, w = 3, 2, 12, 16
  imgs = rs.rand(nl, b, h, w, 3).astype(np.float32)
  masks = rs.rand(nl, b, h, w, 1).astype(np.float32)
  disps = (0.4 * rs.rand(nl, b, h, w, 1)).astype(np.float32)
  trg = rs.rand(b, h, w, 3).astype(np.float32)
  out = {'imgs': imgs, 'masks': masks, 'disps': disps, 'trg': trg}
  out['zbuf_comp_loss'] = loss.zbuffer_composition_loss(
      T(imgs), T(masks), T(disps), T(trg), bg_layer_disp=1e-3, max_disp=0.4,
      zbuf_scale=50).a
  out['decr_disp_loss'] = loss.decreasing_disp_loss(T(disps)).a
  out['decr_disp_loss_L1'] = np.float32(loss.decreasing_disp_loss(T(disps[:1])))
  out['smooth_loss'] = ldi.disp_smoothness_loss(T(disps)).a
  dx, dy = ldi.gradient(T(disps))
  out['grad_dx'], out['grad_dy'] = dx.a, dy.a
  np.savez_compressed(os.path.join(OUT, 'losses.npz'), **out)
  print('losses', out['zbuf_comp_loss'], out['decr_disp_loss'],
        out['smooth_loss'])


def make_view_synthesis():
  """The view-synthesis loss is inline in the reference's training script
  (ldi_enc_dec.py:337-351), which cannot be imported (TF1 session program).
  Its statements are read from the script IN PLACE at generation time and
  executed on the shim; only inputs and outputs are stored."""
  import argparse
  import math
  import textwrap
  path = '/root/reference/ldi_enc_dec.py'
  with open(path, 'r') as f:
    lines = f.readlines()
  # from 'to_recons_img_downsampled = ...' through the crop of pwise_splat_loss
  first = next(i for i, l in enumerate(lines)
               if 'to_recons_img_downsampled = tf.image.resize_images(' in l)
  last = next(i for i, l in enumerate(lines)
              if 'pwise_splat_loss = pwise_splat_loss[:, y_min:y_max, x_min:x_max]' in l)
  assert 330 <= first < last <= 360, (first, last)
  code = compile(textwrap.dedent(''.join(lines[first:last + 1])), path, 'exec')

  def py2_round(x):  # python 2: half away from zero, returns a float
    return math.copysign(math.floor(abs(x) + 0.5), x)

  rs = np.random.RandomState(77)
  out = {}
  for tag, nl, b, h, w, s, bdry in (('compose', 1, 2, 16, 24, 0.5, 0.05),
                                    ('indep', 3, 2, 20, 40, 0.5, 0.05),
                                    ('full', 2, 1, 12, 20, 1.0, 0.1)):
    ht, wt = int(h * s), int(w * s)
    recons = rs.rand(nl, b, ht, wt, 3).astype(np.float32)
    target = rs.rand(b, h, w, 3).astype(np.float32)
    if nl > 1:  # exact ties between layers (reduce_min's gradient splits)
      recons[1, :, :2] = recons[0, :, :2]
    ns = {'tf': tf, 'to_recons_img': T(target), 'recons_splat': T(recons),
          'opts': argparse.Namespace(splat_bdry_ignore=bdry),
          'round': py2_round, 'int': int}
    exec(code, ns)  # pylint: disable=exec-used
    pw = ns['pwise_splat_loss']
    out[tag + '_recons'] = recons
    out[tag + '_target'] = target
    out[tag + '_bdry'] = np.float32(bdry)
    out[tag + '_pwise'] = pw.a
    out[tag + '_loss'] = tf.reduce_mean(pw).a
    print('view synthesis', tag, pw.a.shape, float(out[tag + '_loss']))
  np.savez_compressed(os.path.join(OUT, 'view_synthesis.npz'), **out)


def load_reference_nets(mods):
  """Executes the reference's unchanged lsi/nnutils/nets.py on the slim shim
  (oracle/tf1_slim_shim.py); its `from lsi.nnutils import helpers` resolves to
  the reference's own helpers module loaded by load_reference()."""
  import types
  import tf1_slim_shim as slim
  slim.install()
  names = ('lsi', 'lsi.nnutils', 'lsi.nnutils.helpers')
  saved = {n: sys.modules.get(n) for n in names}
  path = '/root/reference/lsi/nnutils/nets.py'
  nets = types.ModuleType('ref_nets')
  nets.__file__ = path
  try:
    for n in names:
      sys.modules[n] = mods[n]
    with open(path, 'r') as f:
      exec(compile(f.read(), path, 'exec'), nets.__dict__)  # pylint: disable=exec-used
  finally:
    for n, old in saved.items():
      if old is None:
        sys.modules.pop(n, None)
      else:
        sys.modules[n] = old
  return nets, slim


def _sample(name, arr, n=96):
  """Fixed pseudo-random sample of a stage's activations (the indices are a
  pure function of the stage name and the shape) + its mean and std."""
  import zlib
  flat = np.asarray(arr, np.float32).reshape(-1)
  rs = np.random.RandomState(zlib.crc32(name.encode('utf-8')) & 0x7fffffff)
  idx = rs.randint(0, flat.size, size=min(n, flat.size))
  return idx.astype(np.int64), flat[idx], np.array(
      [flat.astype(np.float64).mean(), flat.astype(np.float64).std()])


def make_nets(mods):
  """The networks of nets.py:29-348 with seeded weights (tf1_slim_shim.
  seeded_value: a pure function of the TF variable name, so no weights are
  stored): the variable list the reference creates -- names and shapes, dead
  variables included -- and stage-by-stage activation samples of
    unet:   encoder_decoder_unet(nl_diff_enc_dec=3) + ldi_predictor(L=2, 3 steps)
            on 4 x 128 x 128 images  (define_pred_graph, ldi_enc_dec.py:196-221)
    masks:  the same heads with pred_masks=True (double sigmoid, last mask 1)
    simple: encoder_decoder_simple(nl_diff_enc_dec=3) + ldi_predictor(L=1), 8 images."""
  nets, slim = load_reference_nets(mods)
  # (batch statistics over N*H*W values per channel: at the 1 x 1 bottleneck
  # that is the batch size -- 4 / 8 images keep those layers well conditioned)
  # (inputs are not stored either: RandomState(imgs_seed).rand(12, 128, 128, 3),
  # the first 4 images for unet / masks, the last 8 for simple)
  out = {'imgs_seed': np.int64(99)}
  all_imgs = np.random.RandomState(99).rand(12, 128, 128, 3).astype(np.float32)
  imgs, imgs8 = all_imgs[:4], all_imgs[4:]

  def record(tag, end_points, ldi):
    names = list(slim.VARIABLES)
    out[tag + '_var_names'] = np.array(names)
    out[tag + '_var_shapes'] = np.array(
        [','.join(str(d) for d in slim.VARIABLES[n].shape) for n in names])
    stages = sorted(end_points)
    out[tag + '_stages'] = np.array(stages)
    out[tag + '_stage_shapes'] = np.array(
        [','.join(str(d) for d in end_points[k].a.shape) for k in stages])
    for k in stages:
      idx, val, stat = _sample(k, end_points[k].a)
      out['%s_act_idx/%s' % (tag, k)] = idx
      out['%s_act_val/%s' % (tag, k)] = val
      out['%s_act_stat/%s' % (tag, k)] = stat
    for name, t in zip(('tex', 'mask', 'disp'), ldi):
      idx, val, stat = _sample(tag + '/ldi_' + name, t.a, 2048)
      out['%s_ldi_%s_idx' % (tag, name)] = idx
      out['%s_ldi_%s_val' % (tag, name)] = val
      out['%s_ldi_%s_stat' % (tag, name)] = stat
      out['%s_ldi_%s_shape' % (tag, name)] = np.array(t.a.shape)
    print('nets', tag, len(names), 'variables', len(stages), 'stages',
          sum(int(np.prod(slim.VARIABLES[n].shape)) for n in names), 'parameters')

  def heads_end_points():
    ep = {}
    for coll in slim.COLLECTIONS:
      ep.update(slim.convert_collection_to_dict(coll))
    return ep

  slim.reset()
  _, feat_dec, skip_feat, _ = nets.encoder_decoder_unet(T(imgs), nl_diff_enc_dec=3)
  ldi = nets.ldi_predictor(feat_dec, n_layers=2, n_layerwise_steps=3,
                           skip_feat=skip_feat)
  record('unet', heads_end_points(), ldi)

  slim.reset()
  _, feat_dec, skip_feat, _ = nets.encoder_decoder_unet(T(imgs), nl_diff_enc_dec=3)
  ldi = nets.ldi_predictor(feat_dec, n_layers=2, n_layerwise_steps=3,
                           skip_feat=skip_feat, pred_masks=True)
  record('masks', {k: v for k, v in heads_end_points().items()
                   if k.startswith('ldi_tex_disp')}, ldi)

  slim.reset()
  _, feat_dec, skip_feat, _ = nets.encoder_decoder_simple(T(imgs8), nl_diff_enc_dec=3)
  assert skip_feat is None
  ldi = nets.ldi_predictor(feat_dec, n_layers=1, n_layerwise_steps=3,
                           skip_feat=skip_feat)
  record('simple', heads_end_points(), ldi)
  np.savez_compressed(os.path.join(OUT, 'nets.npz'), **out)


def make_focal(mods):
  """forward_splat with focal_disps (ldi.py:130-143: the disparity is shifted by
  a per-sample focal disparity before the projection and shifted back
  afterwards; lytro data).  Both compose modes, with the disparity output."""
  ldi = mods['lsi.geometry.ldi']
  helpers = mods['lsi.nnutils.helpers']
  proj = mods['lsi.geometry.projection']
  rs = np.random.RandomState(4321)
  nl, b, h, w = 2, 2, 16, 32
  out = {}
  for tag, cams in (('kitti', kitti_cams(b, h, w)),
                    ('general', synth_cams(rs, b, h, w, ang=0.1, tr=0.3, tz=0.2))):
    k_s, k_t, rot, t = cams
    tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
    disp = (0.05 + 0.3 * rs.rand(nl, b, h, w, 1)).astype(np.float32)
    mask = rs.rand(nl, b, h, w, 1).astype(np.float32)
    focal = f32([0.07, 0.15]).reshape(b, 1, 1, 1)
    pc = helpers.pixel_coords(b, h, w)
    out[tag + '_M'] = proj.forward_projection_matrix(T(k_s), T(k_t), T(rot), T(t)).a
    for k_, v in (('tex', tex), ('mask', mask), ('disp', disp), ('focal', focal),
                  ('k_s', k_s), ('k_t', k_t), ('rot', rot), ('t', t)):
      out[tag + '_' + k_] = v
    for compose in (True, False):
      img, wts, dsp = ldi.forward_splat(
          [T(tex), T(mask), T(disp)], pc, T(k_s), T(k_t), T(rot), T(t),
          focal_disps=T(focal), compose_layers=compose, compute_trg_disp=True,
          trg_downsampling=0.5, bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50)
      c = 'compose' if compose else 'indep'
      out['%s_%s_img' % (tag, c)] = img.a
      out['%s_%s_wts' % (tag, c)] = wts.a
      out['%s_%s_disp' % (tag, c)] = dsp.a
  out['params'] = np.array([0.5, 1e-3, 0.4, 50], np.float64)
  np.savez_compressed(os.path.join(OUT, 'focal_splat.npz'), **out)
  print('focal_splat', out['kitti_compose_img'].shape)


def make_coords_splat(mods):
  """forward_splat with source coordinates that are NOT the pixel-centre grid
  (ldi.py:134 concatenates whatever `pixel_coords_src` holds): a sub-pixel
  shifted grid and a smoothly warped one, rectified and general cameras, both
  compose modes, with the disparity output."""
  ldi = mods['lsi.geometry.ldi']
  helpers = mods['lsi.nnutils.helpers']
  proj = mods['lsi.geometry.projection']
  rs = np.random.RandomState(777)
  nl, b, h, w = 2, 2, 16, 32
  out = {}
  for tag, cams in (('kitti', kitti_cams(b, h, w)),
                    ('general', synth_cams(rs, b, h, w, ang=0.1, tr=0.3, tz=0.2))):
    k_s, k_t, rot, t = cams
    tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
    disp = (0.05 + 0.3 * rs.rand(nl, b, h, w, 1)).astype(np.float32)
    mask = rs.rand(nl, b, h, w, 1).astype(np.float32)
    pc = helpers.pixel_coords(b, h, w).a.copy()
    if tag == 'kitti':
      pc[..., 0] += 0.25
      pc[..., 1] -= 0.4
    else:
      pc[..., 0] += f32(1.5 * smooth_noise(rs, (b, h, w), 3) - 0.75)
      pc[..., 1] += f32(1.5 * smooth_noise(rs, (b, h, w), 3) - 0.75)
    pc = f32(pc)
    out[tag + '_M'] = proj.forward_projection_matrix(T(k_s), T(k_t), T(rot), T(t)).a
    for k_, v in (('tex', tex), ('mask', mask), ('disp', disp), ('coords', pc),
                  ('k_s', k_s), ('k_t', k_t), ('rot', rot), ('t', t)):
      out[tag + '_' + k_] = v
    for compose in (True, False):
      img, wts, dsp = ldi.forward_splat(
          [T(tex), T(mask), T(disp)], T(pc), T(k_s), T(k_t), T(rot), T(t),
          compose_layers=compose, compute_trg_disp=True,
          trg_downsampling=0.5, bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50)
      c = 'compose' if compose else 'indep'
      out['%s_%s_img' % (tag, c)] = img.a
      out['%s_%s_wts' % (tag, c)] = wts.a
      out['%s_%s_disp' % (tag, c)] = dsp.a
  out['params'] = np.array([0.5, 1e-3, 0.4, 50], np.float64)
  np.savez_compressed(os.path.join(OUT, 'coords_splat.npz'), **out)
  print('coords_splat', out['kitti_compose_img'].shape)


def make_scene_geometry():
  """The procedural scene generator's geometry: the reference's
  pure-NumPy helpers lsi/data/syntheticPlanes/utils.py:29-201 are executed in
  place (the module's imports of tensorflow / absl resolve to the shim, the
  texture-queue class below line 201 is never instantiated), and the view
  sampler data.py:29-52 is executed from its own lines with numpy's global
  generator seeded.  Stored: inputs and outputs."""
  import textwrap
  import types
  tf.install()
  path = '/root/reference/lsi/data/syntheticPlanes/utils.py'
  utils = types.ModuleType('ref_synth_utils')
  utils.__file__ = path
  with open(path, 'r') as f:
    ulines = f.readlines()
  # everything above the texture-queue class (py2-only print statements there)
  stop = next(i for i, l in enumerate(ulines)
              if l.startswith('class QueuedRandomTextureLoader'))
  assert 195 <= stop <= 210, stop
  exec(compile(''.join(ulines[:stop]), path, 'exec'), utils.__dict__)  # pylint: disable=exec-used
  rs = np.random.RandomState(2024)
  out = {}
  # dims2kmat / resize_instrinsic
  dims = rs.uniform(0.5, 4.0, (6, 2))
  tex_sz = rs.randint(64, 512, (6, 2)).astype(np.float64)
  out['kmat_in'] = np.concatenate([dims, tex_sz], 1)
  out['kmat_out'] = np.stack([utils.dims2kmat(*row) for row in out['kmat_in']])
  out['rsz_scale'] = rs.uniform(0.25, 2.0, (6, 2))
  out['rsz_out'] = np.stack([
      utils.resize_instrinsic(k, sx, sy)
      for k, (sx, sy) in zip(out['kmat_out'], out['rsz_scale'])])
  # get_centre / canonical_transform on random (non-normalised) plane frames
  n = 8
  pts = rs.uniform(-2, 2, (n, 3))
  x_dirs = rs.uniform(-1, 1, (n, 3))
  y_raw = rs.uniform(-1, 1, (n, 3))
  # y orthogonal to x (the generator only builds orthogonal frames)
  y_dirs = y_raw - x_dirs * (np.sum(y_raw * x_dirs, 1, keepdims=True) /
                             np.sum(x_dirs * x_dirs, 1, keepdims=True))
  wh = rs.uniform(0.3, 3.0, (n, 2))
  offs = rs.uniform(0, 1, (n, 2))
  out['gc_pt'], out['gc_x'], out['gc_y'] = pts, x_dirs, y_dirs
  out['gc_wh'], out['gc_off'] = wh, offs
  out['gc_out'] = np.stack([
      utils.get_centre(pts[i], x_dirs[i], y_dirs[i], wh[i, 0], wh[i, 1],
                       offs[i, 0], offs[i, 1]) for i in range(n)])
  out['gc_out_default'] = np.stack([
      utils.get_centre(pts[i], x_dirs[i], y_dirs[i], wh[i, 0], wh[i, 1])
      for i in range(n)])
  ct = [utils.canonical_transform(out['gc_out'][i], x_dirs[i], y_dirs[i])
        for i in range(n)]
  out['ct_rot'] = np.stack([c[0] for c in ct])
  out['ct_trans'] = np.stack([c[1] for c in ct])
  ti = rs.uniform(-1, 1, (n, 3))
  ct2 = [utils.canonical_transform(out['gc_out'][i], x_dirs[i], y_dirs[i], ti[i])
         for i in range(n)]
  out['ct_init'] = ti
  out['ct_rot_init'] = np.stack([c[0] for c in ct2])
  out['ct_trans_init'] = np.stack([c[1] for c in ct2])
  # box_planes: two extents, every parameter of the five planes
  extents = np.array([[-1.0, -1.0, 1.0, 1.0, 1.0, 3.5],
                      [-1.7, -0.9, 0.5, 1.2, 1.1, 2.75]])
  out['box_extent'] = extents
  keys = ('pt', 'x_dir', 'y_dir')
  for e, ext in enumerate(extents):
    planes = utils.box_planes(ext)
    assert len(planes) == 5
    for k in keys:
      out['box%d_%s' % (e, k)] = np.stack(
          [np.asarray(pl[k], np.float64) for pl in planes])
    out['box%d_whoff' % e] = np.array(
        [[pl['w'], pl['h'], pl['off_x'], pl['off_y']] for pl in planes])
  # lookat_rotation
  deltas = rs.uniform(-1, 1, (10, 3))
  deltas[:, 2] = rs.uniform(0.5, 4.0, 10)
  out['lookat_delta'] = deltas
  out['lookat_rot'] = np.stack([utils.lookat_rotation(d) for d in deltas])
  # sample_views: data.py:29-52, its own lines, numpy's global generator seeded
  dpath = '/root/reference/lsi/data/syntheticPlanes/data.py'
  with open(dpath, 'r') as f:
    lines = f.readlines()
  first = next(i for i, l in enumerate(lines) if l.startswith('def sample_views('))
  last = next(i for i, l in enumerate(lines)
              if i > first and l.startswith('  return rot_trans_list'))
  assert 25 <= first < last <= 55, (first, last)
  ns = {'np': np, 'utils': utils}
  exec(compile(textwrap.dedent(''.join(lines[first:last + 1])), dpath, 'exec'), ns)  # pylint: disable=exec-used
  np.random.seed(31337)
  views = ns['sample_views'](5)
  out['views_seed'] = np.int64(31337)
  out['views_rot'] = np.stack([v[0] for v in views])
  out['views_trans'] = np.stack([v[1] for v in views])
  np.savez_compressed(os.path.join(OUT, 'scene_geometry.npz'), **out)
  print('scene geometry', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})


def main():
  os.makedirs(OUT, exist_ok=True)
  mods = tf.load_reference()
  make_known_answers(mods)
  make_forward_splat(mods)
  make_bilinear(mods)
  make_layers(mods)
  make_disocclusion(mods)
  make_losses(mods)
  make_view_synthesis()
  make_scene_geometry()
  make_focal(mods)
  make_coords_splat(mods)
  make_nets(mods)
  total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
  print('wrote %d bytes under %s' % (total, OUT))


if __name__ == '__main__':
  main()
