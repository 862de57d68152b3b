"""Parity of lsi.geometry.sampling / layers / homography / projection on the GPU
(HIP bilinear gather, generic splat, scatter-add) with the reference goldens."""
import numpy as np
import pytest
import torch

import lsi_oracle as O
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev(built_lib):
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  return torch.device('cuda:0')


def T(x, dev):
  return torch.tensor(x, device=dev)


def test_splat_known_answers(dev):
  from lsi.geometry import sampling
  g = golden('known_answers.npz')
  one = torch.ones(1, 1, 1, 1, device=dev)
  for (x, y), want in zip(g['splat_pts'], g['splat_canvases']):
    got = sampling.splat(one, T([[[[x, y]]]], dev).float(),
                         torch.zeros(1, 4, 4, 1, device=dev))
    np.testing.assert_array_equal(got[0, :, :, 0].cpu().numpy(), want)
  two = sampling.splat(torch.ones(1, 1, 2, 1, device=dev),
                       T([[[[2.5, 2.5], [2.5, 2.5]]]], dev).float(),
                       torch.zeros(1, 4, 4, 1, device=dev))
  np.testing.assert_array_equal(two[0, :, :, 0].cpu().numpy(),
                                g['splat_two_same'])


def test_splat_random_against_oracle(dev):
  from lsi.geometry import sampling
  rs = np.random.RandomState(2)
  src = rs.rand(2, 20, 30, 5).astype(np.float32)
  coords = np.stack([rs.uniform(-3, 20, (2, 20, 30)),
                     rs.uniform(-3, 14, (2, 20, 30))], -1).astype(np.float32)
  init = rs.rand(2, 11, 17, 5).astype(np.float32)
  want = O.splat(src, coords, init)
  got = sampling.splat(T(src, dev), T(coords, dev), T(init, dev))
  np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_scatter_add_goldens(dev):
  from lsi.geometry import sampling
  g = golden('known_answers.npz')
  got = sampling.batch_scatter_add_tensor(T(g['bsa_init'], dev),
                                          T(g['bsa_idx'], dev),
                                          T(g['bsa_upd'], dev))
  np.testing.assert_allclose(got.cpu().numpy(), g['bsa_out'], rtol=1e-6)
  got1 = sampling.scatter_add_tensor(T(g['bsa_init'][0], dev),
                                     T(g['bsa_idx'][0][:, None], dev),
                                     T(g['bsa_upd'][0], dev))
  np.testing.assert_allclose(got1.cpu().numpy(), g['sa_out'], rtol=1e-6)
  with pytest.raises(IndexError):
    sampling.batch_scatter_add_tensor(torch.zeros(1, 4, device=dev),
                                      T(np.array([[7]], np.int32), dev),
                                      torch.ones(1, 1, device=dev))
  # gradients: d/d init = g, d/d updates = gather(g)
  init = torch.zeros(2, 6, device=dev, requires_grad=True)
  upd = torch.ones(2, 3, device=dev, requires_grad=True)
  idx = T(np.array([[0, 0, 5], [1, 2, 2]], np.int32), dev)
  out = sampling.batch_scatter_add_tensor(init, idx, upd)
  w = torch.arange(12, device=dev, dtype=torch.float32).reshape(2, 6)
  (out * w).sum().backward()
  np.testing.assert_array_equal(upd.grad.cpu().numpy(),
                                [[0, 0, 5], [7, 8, 8]])
  np.testing.assert_array_equal(init.grad.cpu().numpy(), w.cpu().numpy())


def test_bilinear_goldens(dev):
  from lsi.geometry import sampling
  g = golden('bilinear.npz')
  got = sampling.bilinear(T(g['imgs'], dev), T(g['coords'], dev))
  np.testing.assert_allclose(got.cpu().numpy(), g['out'], rtol=1e-6, atol=1e-7)
  got5 = sampling.bilinear_wrapper(T(g['imgs5'], dev), T(g['coords5'], dev))
  np.testing.assert_allclose(got5.cpu().numpy(), g['out5'], rtol=1e-6,
                             atol=1e-7)
  ims, wts = sampling.bilinear(T(g['imgs'], dev), T(g['coords'], dev),
                               compose=False)
  np.testing.assert_array_equal(torch.stack(ims).cpu().numpy(), g['taps_ims'])
  np.testing.assert_allclose(torch.stack(wts).cpu().numpy(), g['taps_wts'],
                             rtol=1e-6, atol=1e-7)
  # arbitrary leading dims through the wrapper (sampling.py:135-168)
  i5, c5 = T(g['imgs5'], dev), T(g['coords5'], dev)
  ims5, wts5 = sampling.bilinear_wrapper(i5, c5, compose=False)
  assert ims5[0].shape == tuple(g['out5'].shape) and wts5[0].shape[-1] == 1
  comp = sum(w * t for w, t in zip(wts5, ims5))   # taps are border-masked already
  np.testing.assert_allclose(comp.cpu().numpy(), g['out5'], rtol=1e-5, atol=1e-6)


def test_bilinear_and_splat_gradients(dev):
  import lsi_torch_ref as TR
  from lsi.geometry import sampling
  rs = np.random.RandomState(4)
  imgs = rs.rand(2, 9, 11, 3)
  coords = np.stack([rs.uniform(-1, 12, (2, 6, 7)),
                     rs.uniform(-1, 10, (2, 6, 7))], -1)
  # keep clear of cell boundaries so that fp32/fp64 floors agree
  coords = np.where(np.abs(coords - 0.5 - np.round(coords - 0.5)) < 1e-3,
                    coords + 0.01, coords)
  cw = rs.rand(2, 6, 7, 3)
  i64 = torch.tensor(imgs, requires_grad=True)
  c64 = torch.tensor(coords, requires_grad=True)
  (TR.bilinear(i64, c64) * torch.tensor(cw)).sum().backward()
  i32 = torch.tensor(imgs, dtype=torch.float32, device=dev, requires_grad=True)
  c32 = torch.tensor(coords, dtype=torch.float32, device=dev,
                     requires_grad=True)
  (sampling.bilinear(i32, c32) * T(cw, dev).float()).sum().backward()
  np.testing.assert_allclose(i32.grad.cpu().numpy(), i64.grad.numpy(),
                             rtol=1e-4, atol=1e-5)
  np.testing.assert_allclose(c32.grad.cpu().numpy(), c64.grad.numpy(),
                             rtol=1e-4, atol=1e-4)

  # compose=False (sampling.py:124-130): the taps scatter their gradient into the
  # image, the weights give the coordinates' gradient (lsi_bilinear_taps_bwd)
  tw = [rs.rand(2, 6, 7, 3) for _ in range(4)]
  ww = [rs.rand(2, 6, 7, 1) for _ in range(4)]
  i64 = torch.tensor(imgs, requires_grad=True)
  c64 = torch.tensor(coords, requires_grad=True)
  ims64, wts64 = TR.bilinear_taps(i64, c64)
  sum((t * torch.tensor(a)).sum() + (w * torch.tensor(b)).sum()
      for t, w, a, b in zip(ims64, wts64, tw, ww)).backward()
  i32 = torch.tensor(imgs, dtype=torch.float32, device=dev, requires_grad=True)
  c32 = torch.tensor(coords, dtype=torch.float32, device=dev, requires_grad=True)
  ims32, wts32 = sampling.bilinear(i32, c32, compose=False)
  for t, t64, w, w64 in zip(ims32, ims64, wts32, wts64):
    np.testing.assert_allclose(t.detach().cpu().numpy(), t64.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(w.detach().cpu().numpy(), w64.detach().numpy(), rtol=1e-4, atol=1e-5)
  sum((t * T(a, dev).float()).sum() + (w * T(b, dev).float()).sum()
      for t, w, a, b in zip(ims32, wts32, tw, ww)).backward()
  np.testing.assert_allclose(i32.grad.cpu().numpy(), i64.grad.numpy(), rtol=1e-4, atol=1e-5)
  np.testing.assert_allclose(c32.grad.cpu().numpy(), c64.grad.numpy(), rtol=1e-4, atol=1e-4)
  # ... through the wrapper's leading dimensions, only the weights used
  i5 = torch.tensor(imgs[None], dtype=torch.float32, device=dev)
  c5 = torch.tensor(coords[None], dtype=torch.float32, device=dev, requires_grad=True)
  _, w5 = sampling.bilinear_wrapper(i5, c5, compose=False)
  (w5[3] * T(ww[3][None], dev).float()).sum().backward()
  c64b = torch.tensor(coords, requires_grad=True)
  (TR.bilinear_taps(torch.tensor(imgs), c64b)[1][3] * torch.tensor(ww[3])).sum().backward()
  np.testing.assert_allclose(c5.grad[0].cpu().numpy(), c64b.grad.numpy(), rtol=1e-4, atol=1e-4)

  src = rs.rand(2, 6, 7, 3)
  init = rs.rand(2, 9, 11, 3)
  cw2 = rs.rand(2, 9, 11, 3)
  s64 = torch.tensor(src, requires_grad=True)
  c64 = torch.tensor(coords, requires_grad=True)
  n64 = torch.tensor(init, requires_grad=True)
  (TR.splat(s64, c64, n64) * torch.tensor(cw2)).sum().backward()
  s32 = torch.tensor(src, dtype=torch.float32, device=dev, requires_grad=True)
  c32 = torch.tensor(coords, dtype=torch.float32, device=dev,
                     requires_grad=True)
  n32 = torch.tensor(init, dtype=torch.float32, device=dev, requires_grad=True)
  (sampling.splat(s32, c32, n32) * T(cw2, dev).float()).sum().backward()
  np.testing.assert_allclose(s32.grad.cpu().numpy(), s64.grad.numpy(),
                             rtol=1e-4, atol=1e-5)
  np.testing.assert_allclose(c32.grad.cpu().numpy(), c64.grad.numpy(),
                             rtol=1e-4, atol=1e-4)
  np.testing.assert_allclose(n32.grad.cpu().numpy(), n64.grad.numpy(),
                             rtol=1e-6)


def test_planar_transform_and_compose_on_gpu(dev):
  from lsi.geometry import layers
  from lsi.nnutils import helpers
  g = golden('layers.npz')
  pc = helpers.pixel_coords(2, 16, 20, device=dev)
  ti, tm, td = layers.planar_transform(
      T(g['p_imgs'], dev), T(g['p_masks'], dev), pc, T(g['p_k_s'], dev),
      T(g['p_k_t'], dev), T(g['p_rot'], dev), T(g['p_t'], dev),
      T(g['p_n_hat'], dev), T(g['p_a'], dev))
  # the homography algebra runs in the oracle's sequential-k order: the warped
  # coordinates are bit-identical, the sampled images differ by fp32 rounding
  assert float(np.abs(ti.cpu().numpy() - g['p_out_imgs']).max()) < 1e-6
  assert float(np.abs(tm.cpu().numpy() - g['p_out_masks']).max()) < 1e-6
  np.testing.assert_allclose(td.cpu().numpy(), g['p_out_dmaps'], rtol=1e-6,
                             atol=1e-7)
  got = layers.compose(T(g['imgs'], dev), T(g['masks'], dev),
                       T(g['dmaps'], dev))
  np.testing.assert_allclose(got.cpu().numpy(), g['compose_hard'], rtol=1e-5,
                             atol=1e-6)


def test_disocclusion_mask_on_gpu(dev):
  from lsi.geometry import projection
  from lsi.nnutils import helpers
  g = golden('disocclusion.npz')
  b, h, w, _ = g['disps_src'].shape
  got = projection.disocclusion_mask(T(g['disps_src'], dev),
                                     T(g['disps_trg'], dev),
                                     helpers.pixel_coords(b, h, w, device=dev),
                                     T(g['M'], dev))
  # a thresholded quantity on bit-identical coordinates: the mask is exact
  assert np.array_equal(got.cpu().numpy(), g['mask'])


def test_eval_metrics_against_numpy_restatement(dev):
  """Masked L1 / PSNR / disparity metrics of ldi_pred_eval.py:297-548 on the
  KITTI-like golden LDI, vs the same arithmetic on the oracle's outputs."""
  import types
  from lsi.nnutils import eval_metrics, helpers
  g = golden('fs_kitti_L2_s05.npz')
  s, bg, md, zb = [float(v) for v in g['params']]
  opts = types.SimpleNamespace(trg_splat_downsampling=s, zbuf_scale=zb,
                               bg_layer_disp=bg, max_disp=md,
                               splat_bdry_ignore=0.1)
  nl, b, h, w, _ = g['tex'].shape
  rs = np.random.RandomState(0)
  target = rs.rand(b, h, w, 3).astype(np.float32)
  gt_disp = (0.4 * rs.rand(b, h, w, 1)).astype(np.float32)
  ldi_src = [T(g[k], dev) for k in ('tex', 'mask', 'disp')]
  got = eval_metrics.view_synthesis_metrics(
      ldi_src, helpers.pixel_coords(b, h, w), torch.tensor(g['k_s']),
      torch.tensor(g['k_t']), torch.tensor(g['rot']), torch.tensor(g['t']),
      T(target, dev), opts, gt_disp_trg=T(gt_disp, dev))
  ht, wt = int(h * s), int(w * s)
  tds = O.area_downsample(target, ht, wt)
  centre = np.zeros((b, ht, wt), np.float32)
  x_min, y_min = O.py2_round(wt * 0.1), O.py2_round(ht * 0.1)
  centre[:, y_min:ht - y_min, x_min:wt - x_min] = 1
  pw = np.mean(np.abs(tds - g['compose_img'][0]), axis=3) * centre
  assert abs(float(got['compose_splat_loss'][0]) - pw.sum()) < 1e-3 * pw.sum()
  assert float(got['compose_splat_loss'][1]) == centre.sum()
  gds = O.area_downsample(gt_disp, ht, wt)
  pd = np.mean(np.abs(gds - g['compose_disp'][0]), axis=3) * centre
  assert abs(float(got['depth_splat_loss'][0]) - pd.sum()) < 1e-3 * pd.sum()
  agg = eval_metrics.aggregate([got, got])
  assert abs(agg['compose_splat_loss'] - pw.sum() / centre.sum()) < 1e-5
  assert 0 < agg['psnr'] < 60


def test_eval_disocclusion_and_layer_metrics_against_the_oracle(dev):
  """The metrics of ldi_pred_eval.py:403-451 (dis-occlusion-restricted L1 and
  depth error, valid-pixel masks) and :476-531 (foreground / background layer
  texture and disparity errors) on random inputs, against the NumPy
  restatement in oracle/lsi_oracle.py: sums AND normalisers."""
  import types
  from lsi.nnutils import eval_metrics, helpers
  g = golden('fs_kitti_L2_s05.npz')
  s, bg, md, zb = [float(v) for v in g['params']]
  opts = types.SimpleNamespace(trg_splat_downsampling=s, zbuf_scale=zb,
                               bg_layer_disp=0.05, max_disp=md,
                               splat_bdry_ignore=0.1)
  nl, b, h, w, _ = g['tex'].shape
  rs = np.random.RandomState(5)
  f32 = lambda a: np.asarray(a, np.float32)
  target = f32(rs.rand(b, h, w, 3))
  gt_disp = f32(0.4 * rs.rand(b, h, w, 1))
  gt_disp[:, : h // 4] = 0.01                    # a band below bg_layer_disp
  disocc = rs.rand(b, h, w, 1) > 0.6             # boolean mask, as the data has
  valid = f32(gt_disp > opts.bg_layer_disp)
  ldi_src = [T(g[k], dev) for k in ('tex', 'mask', 'disp')]
  got = eval_metrics.view_synthesis_metrics(
      ldi_src, helpers.pixel_coords(b, h, w), torch.tensor(g['k_s']),
      torch.tensor(g['k_t']), torch.tensor(g['rot']), torch.tensor(g['t']),
      T(target, dev), opts, valid_mask=T(valid, dev),
      disocc_mask=torch.tensor(disocc, device=dev), gt_disp_trg=T(gt_disp, dev))
  r = O.forward_splat(g['tex'], g['mask'], g['disp'], g['M'], s, 0.05, md, zb, True)
  want = O.eval_view_synthesis_metrics(r['img'], r['disp'], target, 0.1, valid,
                                       disocc, gt_disp)
  assert set(want) <= set(got)
  for k, (ws, wn) in want.items():
    assert abs(float(got[k][0]) - ws) <= 2e-4 * abs(ws), (k, float(got[k][0]), ws)
    assert abs(float(got[k][1]) - wn) <= 1e-5 * abs(wn), (k, float(got[k][1]), wn)
  assert want['compose_splat_loss_disocc'][1] < want['compose_splat_loss'][1]

  # layer metrics: two random LDIs, gt foreground / background layers
  ldis = [[f32(rs.rand(nl, b, h, w, 3)), None, f32(0.4 * rs.rand(nl, b, h, w, 1))]
          for _ in range(2)]
  imgs = [f32(rs.rand(b, h, w, 3)) for _ in range(2)]
  gt = {}
  for side in ('src', 'trg'):
    gt[side + '_gt_disp'] = f32(0.4 * rs.rand(b, h, w, 1))
    gt[side + '_gt_disp_bg'] = f32(0.4 * rs.rand(b, h, w, 1))
    gt[side + '_gt_tex_bg'] = f32(rs.rand(b, h, w, 3))
  dev_ldis = [[T(l[0], dev), None, T(l[2], dev)] for l in ldis]
  got_l = eval_metrics.layer_prediction_metrics(
      dev_ldis[0], dev_ldis[1], T(imgs[0], dev), T(imgs[1], dev),
      {k: T(v, dev) for k, v in gt.items()}, opts)
  want_l = O.eval_layer_prediction_metrics(ldis[0], ldis[1], imgs[0], imgs[1], gt,
                                           opts.bg_layer_disp)
  assert set(got_l) == set(want_l)
  for k, (ws, wn) in want_l.items():
    assert abs(float(got_l[k][0]) - ws) <= 1e-4 * abs(ws), k
    assert float(got_l[k][1]) == wn, k
  agg = eval_metrics.aggregate([got_l, got_l])
  assert abs(agg['bg_disp_error'] - want_l['bg_disp_error'][0] /
             want_l['bg_disp_error'][1]) < 1e-5


def test_scene_generator_views_are_geometrically_consistent(dev):
  """Procedural scene -> two views through planar_transform + compose (HIP
  bilinear).  Splatting the source view with its ground-truth disparity into the
  target camera must reproduce the directly rendered target view wherever the
  source sees the surface (the reference's `debug_synth_texture` check)."""
  from lsi.data import synthetic_planes
  from lsi.geometry import ldi
  from lsi.nnutils import helpers
  gen = synthetic_planes.SceneGenerator(128, 128, n_obj=2, device=dev, seed=3)
  src, trg, k_s, k_t, rot, t, d_src, d_trg = gen.forward(2)
  assert src.shape == (2, 128, 128, 3) and d_src.shape == (2, 128, 128, 1)
  assert float(src.min()) >= 0 and float(src.max()) <= 1 + 1e-5
  # box depth 2 .. 3.5 (disparity 0.29 .. 0.5); outside the box the renderer's
  # background disparity min_disp = 0.2 (syntheticPlanes/data.py:363)
  assert float(d_src.min()) >= 0.2 - 1e-6 and float(d_src.max()) < 0.6
  # rotation matrices are orthonormal, relative pose maps src frame to trg frame
  eye = torch.matmul(rot, rot.transpose(1, 2))
  assert float((eye - torch.eye(3)).abs().max()) < 1e-5
  img, wts = ldi.forward_splat([src[None], None, d_src[None]],
                               helpers.pixel_coords(2, 128, 128), k_s, k_t, rot,
                               t, trg_downsampling=1, bg_layer_disp=1e-3,
                               max_disp=1.0, zbuf_scale=50)
  covered = (wts[0, ..., 0] > 1e-6).float()        # pixels the source reaches
  err = ((img[0] - trg).abs().mean(dim=3) * covered).sum() / covered.sum()
  assert float(covered.mean()) > 0.25   # the box fills the middle of the view
  assert float(err) < 0.06, float(err)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(4))
def test_sampling_ops_under_hostile_coordinates(seed, dev):
  """bilinear and splat with coordinates on cell borders and centres, far
  outside the image, huge, NaN and Inf: same values as the oracle, and finite
  gradients (a non-finite coordinate samples / adds nothing)."""
  from lsi.geometry import sampling
  rs = np.random.RandomState(40 + seed)
  b, hs, ws, c, ht, wt = 2, 9, 13, 3, 7, 11
  imgs = rs.rand(b, hs, ws, c).astype(np.float32)
  coords = np.stack([rs.uniform(-4, ws + 4, (b, ht, wt)),
                     rs.uniform(-4, hs + 4, (b, ht, wt))], -1).astype(np.float32)
  special = np.array([0.0, 0.5, 1.0, -0.5, ws - 0.5, ws, ws + 0.5, 1e9, -1e9,
                      np.nan, np.inf, -np.inf, 3.5, 2.999999], np.float32)
  pick = rs.rand(b, ht, wt, 2) < 0.35
  coords[pick] = special[rs.randint(0, len(special), int(pick.sum()))]
  want = O.bilinear(imgs, coords)
  ti = T(imgs, dev).requires_grad_(True)
  tc = T(coords, dev).requires_grad_(True)
  got = sampling.bilinear(ti, tc)
  np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=1e-6,
                             atol=1e-7)
  got.sum().backward()
  assert bool(torch.isfinite(ti.grad).all() and torch.isfinite(tc.grad).all())
  # splat: source image to hostile target coordinates
  src = rs.rand(b, ht, wt, c).astype(np.float32)
  init = rs.rand(b, hs, ws, c).astype(np.float32)
  want_s = O.splat(src, coords, init)
  tsrc = T(src, dev).requires_grad_(True)
  tc2 = T(coords, dev).requires_grad_(True)
  got_s = sampling.splat(tsrc, tc2, T(init, dev))
  np.testing.assert_allclose(got_s.detach().cpu().numpy(), want_s, rtol=1e-5,
                             atol=1e-6)
  got_s.sum().backward()
  assert bool(torch.isfinite(tsrc.grad).all() and torch.isfinite(tc2.grad).all())
