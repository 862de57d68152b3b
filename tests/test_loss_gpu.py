"""Parity of the fused HIP loss / composition kernels (csrc/lsi_loss.hip,
through the C ABI) with the reference-generated goldens and, for gradients,
with fp64 torch autograd of the same op graph (the reference has no backward
code: TF differentiates the graph).

Bars: loss scalars 2e-6 relative (fp32 sums accumulated in fp64 here, in some
tree order in TF); gradients 1e-5 relative to the largest entry; compose
outputs 1e-6 absolute, hard selections exact.
"""
import numpy as np
import pytest
import torch

import lsi_oracle as O
from conftest import golden

pytestmark = pytest.mark.gpu

LOSS_RTOL = 2e-6


@pytest.fixture(scope='module')
def dev(built_lib):
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  return torch.device('cuda:0')


def T(a, dev, grad=False):
  t = torch.tensor(np.asarray(a, np.float32), device=dev)
  return t.requires_grad_(True) if grad else t


def _product():
  from lsi.geometry import ldi
  from lsi.loss import loss
  return loss, ldi


def _oracle():
  """oracle/lsi_torch_ref.py: the same op graphs in torch ops, differentiated
  in fp64 -- the gradient oracle (forward values pinned by the goldens in
  tests/test_oracle_golden.py)."""
  import lsi_torch_ref as TR
  return TR


def _close(got, want, rtol=1e-5):
  got, want = got.detach().cpu().double().numpy(), want.detach().double().numpy()
  scale = np.abs(want).max() + 1e-30
  assert np.abs(got - want).max() <= rtol * scale, (
      np.abs(got - want).max(), scale)


def test_zbuffer_composition_loss_forward_and_backward(dev):
  loss, _ = _product()
  TR = _oracle()
  g = golden('losses.npz')
  args = dict(bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50)
  imgs, masks, disps = T(g['imgs'], dev, True), T(g['masks'], dev, True), T(g['disps'], dev, True)
  got = loss.zbuffer_composition_loss(imgs, masks, disps, T(g['trg'], dev), **args)
  want = float(g['zbuf_comp_loss'])
  assert abs(float(got) - want) <= LOSS_RTOL * abs(want)
  (got * 3.0).backward()
  ci, cm, cd = [torch.tensor(g[k], dtype=torch.float64, requires_grad=True)
                for k in ('imgs', 'masks', 'disps')]
  ref = TR.zbuffer_composition_loss(ci, cm, cd,
                                    torch.tensor(g['trg'], dtype=torch.float64),
                                    **args)
  (ref * 3.0).backward()
  _close(imgs.grad, ci.grad)
  _close(masks.grad, cm.grad)
  _close(disps.grad, cd.grad)
  # no mask input = ones; and strided (planar, conv-output style) layers
  got1 = loss.zbuffer_composition_loss(T(g['imgs'], dev), None, T(g['disps'], dev),
                                       T(g['trg'], dev), **args)
  want1 = float(O.zbuffer_composition_loss(g['imgs'], np.ones_like(g['masks']),
                                           g['disps'], g['trg'], 1e-3, 0.4, 50))
  assert abs(float(got1) - want1) <= LOSS_RTOL * abs(want1)
  planar = T(g['imgs'], dev).permute(0, 1, 4, 2, 3).contiguous().permute(0, 1, 3, 4, 2)
  assert not planar.is_contiguous()
  got2 = loss.zbuffer_composition_loss(planar, T(g['masks'], dev), T(g['disps'], dev),
                                       T(g['trg'], dev), **args)
  assert abs(float(got2) - want) <= LOSS_RTOL * abs(want)


def test_disparity_regularisers_forward_and_backward(dev):
  loss, ldi = _product()
  TR = _oracle()
  g = golden('losses.npz')
  d = T(g['disps'], dev, True)
  smooth = ldi.disp_smoothness_loss(d)
  decr = loss.decreasing_disp_loss(d)
  for got, key in ((smooth, 'smooth_loss'), (decr, 'decr_disp_loss')):
    want = float(g[key])
    assert abs(float(got) - want) <= LOSS_RTOL * abs(want), key
  assert loss.decreasing_disp_loss(d[:1]) == 0           # L == 1 (loss.py:58)
  (0.7 * smooth + 1.3 * decr).backward()
  c = torch.tensor(g['disps'], dtype=torch.float64, requires_grad=True)
  (0.7 * TR.disp_smoothness_loss(c) + 1.3 * TR.decreasing_disp_loss(c)).backward()
  _close(d.grad, c.grad)
  # a smooth (piecewise-linear) field: second differences that vanish exactly
  # must get the TF gradient abs'(0) = 0
  ramp = np.zeros((1, 1, 6, 8, 1), np.float32)
  ramp[0, 0, :, :, 0] = np.arange(8, dtype=np.float32)[None, :] * 0.25
  r = T(ramp, dev, True)
  ldi.disp_smoothness_loss(r).backward()
  assert float(r.grad.abs().max()) == 0.0


@pytest.mark.parametrize('tag', ['compose', 'indep', 'full'])
def test_view_synthesis_loss_forward_and_backward(tag, dev):
  loss, _ = _product()
  TR = _oracle()
  g = golden('view_synthesis.npz')
  recons_np, target_np = g[tag + '_recons'], g[tag + '_target']
  bdry = float(g[tag + '_bdry'])
  r = T(recons_np, dev, True)
  got = loss.view_synthesis_loss(r, T(target_np, dev), bdry)
  want = float(g[tag + '_loss'])
  assert abs(float(got) - want) <= LOSS_RTOL * abs(want)
  (got * 2.0).backward()
  c = torch.tensor(recons_np, dtype=torch.float64, requires_grad=True)
  (TR.view_synthesis_loss(c, torch.tensor(target_np, dtype=torch.float64),
                          bdry) * 2.0).backward()
  gd, cd = r.grad.cpu().double(), c.grad
  if recons_np.shape[0] > 1:
    # rows 0-1: layers 0 and 1 are identical (exact ties).  TF's reduce_min
    # splits the gradient evenly among tied layers; torch.min gives it to one.
    tied = slice(0, 2)
    both = gd[0, :, tied] + gd[1, :, tied]
    torch.testing.assert_close(both, cd[0, :, tied] + cd[1, :, tied], rtol=1e-5,
                               atol=1e-9)
    torch.testing.assert_close(gd[0, :, tied], gd[1, :, tied], rtol=0, atol=0)
    gd, cd = gd[:, :, 2:], cd[:, :, 2:]
  torch.testing.assert_close(gd, cd, rtol=1e-5, atol=1e-9)


def test_compose_variants_match_goldens(dev):
  from lsi.geometry import layers
  g = golden('layers.npz')
  imgs, masks, dmaps = T(g['imgs'], dev), T(g['masks'], dev), T(g['dmaps'], dev)
  hard = layers.compose(imgs, masks, dmaps)
  assert np.array_equal(hard.cpu().numpy(), g['compose_hard'])   # a selection
  # (parameters of the fixtures: oracle/make_goldens.py:make_layers)
  soft = layers.compose(imgs, masks, dmaps, soft=True, min_disp=1e-3,
                        depth_softmax_temp=0.4)
  np.testing.assert_allclose(soft.cpu().numpy(), g['compose_soft'], rtol=0,
                             atol=1e-6)
  depth = layers.compose_depth(masks, dmaps)
  assert np.array_equal(depth.cpu().numpy(), g['compose_depth'])
  depth_bg = layers.compose_depth(masks, dmaps, bg_layer=True, min_disp=1e-3,
                                  depth_softmax_temp=0.4)
  assert np.array_equal(depth_bg.cpu().numpy(), g['compose_depth_bg'])
  with pytest.raises(RuntimeError, match='forward-only'):
    layers.compose(imgs.clone().requires_grad_(True), masks, dmaps)
  with pytest.raises(RuntimeError, match='forward-only'):
    layers.compose_depth(masks, dmaps.clone().requires_grad_(True))
  # the target images of the self-consistency loss are data: asking for their
  # gradient is an error, not a silent None
  from lsi.loss import _hip
  with pytest.raises(RuntimeError, match='not differentiable'):
    _hip.zbuffer_composition_loss(imgs, masks, dmaps,
                                  imgs[0].clone().requires_grad_(True), 1e-3, 1.0, 10.0)


def test_losses_and_composition_have_no_cpu_path():
  """a tensor that does not live on a ROCm device raises in
  every HIP-backed op -- there is no second (torch) implementation."""
  from lsi.geometry import layers, ldi
  from lsi.loss import loss
  d = torch.rand(2, 1, 8, 8, 1)
  im = torch.rand(2, 1, 8, 8, 3)
  for fn in (lambda: loss.decreasing_disp_loss(d),
             lambda: ldi.disp_smoothness_loss(d),
             lambda: loss.zbuffer_composition_loss(im, d, d, im[0]),
             lambda: loss.view_synthesis_loss(im, torch.rand(1, 16, 16, 3)),
             lambda: layers.compose(im, d, d),
             lambda: layers.compose_depth(d, d)):
    with pytest.raises(RuntimeError):
      fn()


def test_six_loss_scalars_of_a_fixed_batch_match_the_oracle(dev):
  """The six scalars the training script logs (ldi_enc_dec.py:398-410) for one
  fixed 'batch': the KITTI-like golden LDI rendered into the target view and
  back, every term through the HIP path, against the NumPy oracle."""
  from lsi.geometry import ldi
  from lsi.loss import loss
  g = golden('fs_kitti_L2_s05.npz')
  s, bg, md, zb = [float(v) for v in g['params']]
  tex, mask, disp, M = g['tex'], g['mask'], g['disp'], g['M']
  rs = np.random.RandomState(3)
  nl, b, h, w, _ = tex.shape
  img_src = rs.rand(b, h, w, 3).astype(np.float32)
  img_trg = rs.rand(b, h, w, 3).astype(np.float32)
  wts = dict(self_cons=10.0, compose=1.0, indep=1.0, incr=1.0, smooth=0.1)

  ldi_dev = [T(tex, dev), T(mask, dev), T(disp, dev)]
  got = {}
  got['self_cons_loss'] = loss.zbuffer_composition_loss(
      ldi_dev[0], ldi_dev[1], ldi_dev[2], T(img_src, dev), bg_layer_disp=bg,
      max_disp=md, zbuf_scale=zb)
  for compose, key in ((True, 'compose_splat_loss'), (False, 'indep_splat_loss')):
    img, _ = ldi.forward_splat_matrix(ldi_dev, torch.tensor(M),
                                      compose_layers=compose, trg_downsampling=s,
                                      bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
    got[key] = loss.view_synthesis_loss(img, T(img_trg, dev), 0.05)
  got['disp_smoothness_loss'] = ldi.disp_smoothness_loss(ldi_dev[2])
  got['incr_depth_loss'] = loss.decreasing_disp_loss(ldi_dev[2])
  got['total_loss'] = (wts['self_cons'] * got['self_cons_loss'] +
                       wts['compose'] * got['compose_splat_loss'] +
                       wts['indep'] * got['indep_splat_loss'] +
                       wts['incr'] / md * got['incr_depth_loss'] +
                       wts['smooth'] / (md * md) * got['disp_smoothness_loss'])

  want = {}
  want['self_cons_loss'] = O.zbuffer_composition_loss(tex, mask, disp, img_src,
                                                      bg, md, zb)
  for compose, key in ((True, 'compose_splat_loss'), (False, 'indep_splat_loss')):
    r = O.forward_splat(tex, mask, disp, M, s, bg, md, zb, compose)
    want[key] = O.view_synthesis_loss(r['img'], img_trg, 0.05)
  want['disp_smoothness_loss'] = O.disp_smoothness_loss(disp)
  want['incr_depth_loss'] = O.decreasing_disp_loss(disp)
  want['total_loss'] = (wts['self_cons'] * want['self_cons_loss'] +
                        wts['compose'] * want['compose_splat_loss'] +
                        wts['indep'] * want['indep_splat_loss'] +
                        wts['incr'] / md * want['incr_depth_loss'] +
                        wts['smooth'] / (md * md) * want['disp_smoothness_loss'])
  for k, v in want.items():
    assert abs(float(got[k]) - float(v)) <= 1e-5 * abs(float(v)), (k, float(got[k]), float(v))
