"""Parity of the HIP forward-splat renderer (through the C ABI) with the oracle.

Bars: projected pixel indices BIT-EXACT; rendered RGB |err| <= 2e-5 (values in
[0,1]); un-normalised weights and disparity 1e-4 relative (fp32 sums of
positive terms in a different order than the reference; exp within ~2 ulp).
"""
import glob
import os

import numpy as np
import pytest
import torch

import lsi_oracle as O
from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu

FS_CASES = sorted(os.path.basename(f)
                  for f in glob.glob(os.path.join(GOLDEN, 'fs_*.npz')))
IMG_ATOL, WTS_RTOL, DSP_RTOL = 2e-5, 1e-4, 1e-4


@pytest.fixture(scope='module')
def dev(built_lib):
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  return torch.device('cuda:0')


def _params(g):
  s, bg, md, zb = [float(v) for v in g['params']]
  s = int(s) if s == int(s) else s
  return s, bg, md, zb


def _cmp(got, want, tag):
  img, wts, dsp = [t.cpu().numpy() for t in got]
  np.testing.assert_allclose(img, want[tag + '_img'], rtol=0, atol=IMG_ATOL)
  np.testing.assert_allclose(wts, want[tag + '_wts'], rtol=WTS_RTOL, atol=0)
  np.testing.assert_allclose(dsp, want[tag + '_disp'], rtol=DSP_RTOL, atol=1e-7)


def _rowband_ok(g):
  m = g['M']
  return bool(np.all(m[:, 1, 3] == 0) and np.all(m[:, 2, 3] == 0))


def _stream_ok(g):
  m = g['M']
  return bool(_rowband_ok(g) and np.all(m[:, 1, 0] == 0) and
              np.all(m[:, 2, 0] == 0) and g['tex'].shape[3] % 4 == 0)


@pytest.mark.parametrize('case', FS_CASES)
@pytest.mark.parametrize('compose', [True, False])
@pytest.mark.parametrize('path', ['atomic', 'rowband', 'stream', 'tile'])
def test_forward_splat_matches_reference_goldens(case, compose, path, dev):
  from lsi.geometry import ldi
  g = golden(case)
  if path == 'rowband' and not _rowband_ok(g):
    pytest.skip('projection is not row-band (general pose)')
  if path == 'stream' and not _stream_ok(g):
    pytest.skip('projection is not row-uniform (stream path does not apply)')
  s, bg, md, zb = _params(g)
  ldi_src = [torch.tensor(g[k], device=dev) for k in ('tex', 'mask', 'disp')]
  if path != 'stream':  # the stream path renders RGB + weights only
    got = ldi.forward_splat_matrix(
        ldi_src, torch.tensor(g['M']), compose_layers=compose,
        compute_trg_disp=True, trg_downsampling=s, bg_layer_disp=bg,
        max_disp=md, zbuf_scale=zb, path=path)
    _cmp(got, g, 'compose' if compose else 'indep')
  # without the disparity output (the training configuration)
  img, wts = ldi.forward_splat_matrix(
      ldi_src, torch.tensor(g['M']), compose_layers=compose,
      trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb,
      path=path)
  tag = 'compose' if compose else 'indep'
  np.testing.assert_allclose(img.cpu().numpy(), g[tag + '_img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts.cpu().numpy(), g[tag + '_wts'], rtol=WTS_RTOL)


@pytest.mark.parametrize('case', FS_CASES)
def test_projected_pixel_indices_are_bit_exact(case, dev):
  from lsi.geometry import ldi
  g = golden(case)
  s, _, md, zb = _params(g)
  idx4, upd4 = ldi.project_indices(torch.tensor(g['disp'], device=dev),
                                   torch.tensor(g['mask'], device=dev),
                                   torch.tensor(g['M']), s, md, zb)
  np.testing.assert_array_equal(idx4.cpu().numpy(), g['idx4'])
  np.testing.assert_allclose(upd4.cpu().numpy(), g['upd4'], rtol=2e-5, atol=0)


def test_camera_api_matches_golden(dev):
  from lsi.geometry import ldi
  from lsi.nnutils import helpers
  g = golden('fs_kitti_L2_s05.npz')
  s, bg, md, zb = _params(g)
  ldi_src = [torch.tensor(g[k], device=dev) for k in ('tex', 'mask', 'disp')]
  nl, b, h, w, _ = g['tex'].shape
  got = ldi.forward_splat(
      ldi_src, helpers.pixel_coords(b, h, w), torch.tensor(g['k_s']),
      torch.tensor(g['k_t']), torch.tensor(g['rot']), torch.tensor(g['t']),
      compose_layers=True, compute_trg_disp=True, trg_downsampling=s,
      bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  _cmp(got, g, 'compose')
  # focal_disps = 0 is the identity (ldi.py:130-143)
  zero = ldi.forward_splat(
      ldi_src, None, torch.tensor(g['k_s']), torch.tensor(g['k_t']),
      torch.tensor(g['rot']), torch.tensor(g['t']),
      focal_disps=torch.zeros(b, 1, 1, 1), compose_layers=True,
      compute_trg_disp=True, trg_downsampling=s, bg_layer_disp=bg, max_disp=md,
      zbuf_scale=zb)
  _cmp(zero, g, 'compose')


@pytest.mark.parametrize('tag', ['kitti', 'general'])
@pytest.mark.parametrize('compose', [True, False])
def test_caller_source_coordinates_match_the_reference(tag, compose, dev):
  """forward_splat with a pixel_coords_src that is not the pixel grid (reference
  ldi.py:134 renders whatever it holds): against outputs of the reference
  itself (tests/golden/coords_splat.npz: a sub-pixel shifted grid with rectified
  cameras, a smoothly warped one with general cameras); differentiable."""
  from lsi.geometry import ldi
  g = golden('coords_splat.npz')
  s, bg, md, zb = [float(v) for v in g['params']]
  ldi_src = [torch.tensor(g[tag + '_' + k], device=dev) for k in ('tex', 'mask', 'disp')]
  ldi_src[0].requires_grad_(True)
  ldi_src[2].requires_grad_(True)
  coords = torch.tensor(g[tag + '_coords'], device=dev)
  got = ldi.forward_splat(
      ldi_src, coords, torch.tensor(g[tag + '_k_s']), torch.tensor(g[tag + '_k_t']),
      torch.tensor(g[tag + '_rot']), torch.tensor(g[tag + '_t']), compose_layers=compose,
      compute_trg_disp=True, trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  c = 'compose' if compose else 'indep'
  img, wts, dsp = [t.detach().cpu().numpy() for t in got]
  np.testing.assert_allclose(img, g['%s_%s_img' % (tag, c)], rtol=0, atol=IMG_ATOL)
  np.testing.assert_allclose(wts, g['%s_%s_wts' % (tag, c)], rtol=WTS_RTOL, atol=0)
  np.testing.assert_allclose(dsp, g['%s_%s_disp' % (tag, c)], rtol=DSP_RTOL, atol=1e-7)
  got[0].mean().backward()
  assert bool(torch.isfinite(ldi_src[0].grad).all() and torch.isfinite(ldi_src[2].grad).all())
  assert float(ldi_src[0].grad.abs().max()) > 0 and float(ldi_src[2].grad.abs().max()) > 0
  # the same tensors with the grid itself take the fused kernels and differ
  from lsi.nnutils import helpers
  b, h, w = g[tag + '_tex'].shape[1:4]
  plain = ldi.forward_splat(
      [t.detach() for t in ldi_src], helpers.pixel_coords(b, h, w, device=dev),
      torch.tensor(g[tag + '_k_s']), torch.tensor(g[tag + '_k_t']),
      torch.tensor(g[tag + '_rot']), torch.tensor(g[tag + '_t']), compose_layers=compose,
      trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  assert float((plain[0] - got[0].detach()).abs().max()) > 1e-3


@pytest.mark.parametrize('tag', ['kitti', 'general'])
@pytest.mark.parametrize('compose', [True, False])
def test_focal_disps_matches_the_reference(tag, compose, dev):
  """forward_splat(focal_disps=...) (ldi.py:130-143) against outputs of the
  reference itself (tests/golden/focal_splat.npz)."""
  from lsi.geometry import ldi
  from lsi.nnutils import helpers
  g = golden('focal_splat.npz')
  s, bg, md, zb = [float(v) for v in g['params']]
  ldi_src = [torch.tensor(g[tag + '_' + k], device=dev) for k in ('tex', 'mask', 'disp')]
  b, h, w = g[tag + '_tex'].shape[1:4]
  got = ldi.forward_splat(
      ldi_src, helpers.pixel_coords(b, h, w), torch.tensor(g[tag + '_k_s']),
      torch.tensor(g[tag + '_k_t']), torch.tensor(g[tag + '_rot']),
      torch.tensor(g[tag + '_t']), focal_disps=torch.tensor(g[tag + '_focal']),
      compose_layers=compose, compute_trg_disp=True, trg_downsampling=s,
      bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  c = 'compose' if compose else 'indep'
  img, wts, dsp = [t.cpu().numpy() for t in got]
  np.testing.assert_allclose(img, g['%s_%s_img' % (tag, c)], rtol=0, atol=IMG_ATOL)
  np.testing.assert_allclose(wts, g['%s_%s_wts' % (tag, c)], rtol=WTS_RTOL, atol=0)
  np.testing.assert_allclose(dsp, g['%s_%s_disp' % (tag, c)], rtol=DSP_RTOL,
                             atol=1e-7)
  # and it differs from the rendering without the focal shift
  plain = ldi.forward_splat(
      ldi_src, helpers.pixel_coords(b, h, w), torch.tensor(g[tag + '_k_s']),
      torch.tensor(g[tag + '_k_t']), torch.tensor(g[tag + '_rot']),
      torch.tensor(g[tag + '_t']), compose_layers=compose,
      trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  assert float((plain[0] - got[0]).abs().max()) > 1e-3


def _synth(rs, nl, b, h, w, kitti=True, max_disp=0.4):
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = (max_disp * rs.rand(nl, b, h, w, 1)).astype(np.float32)
  if kitti:
    k = np.array([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1]],
                 np.float32)
    k = np.broadcast_to(k, (b, 3, 3)).copy()
    rot = np.broadcast_to(np.eye(3, dtype=np.float32), (b, 3, 3)).copy()
    t = np.broadcast_to(np.array([[-0.532], [0], [0]], np.float32),
                        (b, 3, 1)).copy()
  else:
    k = np.array([[w, 0, w / 2], [0, h, h / 2], [0, 0, 1]], np.float32)
    k = np.broadcast_to(k, (b, 3, 3)).copy()
    ang = rs.uniform(-0.1, 0.1, (b, 3))
    rot = np.stack([_rot(*a) for a in ang]).astype(np.float32)
    t = rs.uniform(-0.3, 0.3, (b, 3, 1)).astype(np.float32)
  mat = O.forward_projection_matrix(k, k, rot, t)
  return tex, disp, mat


def _rot(ax, ay, az):
  cx, sx, cy, sy, cz, sz = (np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay),
                            np.cos(az), np.sin(az))
  rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
  ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
  rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
  return rz @ ry @ rx


@pytest.mark.parametrize('path', ['atomic', 'rowband', 'stream', 'tile'])
@pytest.mark.parametrize('compose', [True, False])
def test_config2_size_against_c_oracle(path, compose, dev, ref_cpu):
  """BASELINE config 2 shape (2-layer 256x768, s=0.5), batch 2, vs the C oracle."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(5)
  tex, disp, mat = _synth(rs, 2, 2, 256, 768)
  want = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50,
                               compose)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  out = ldi.forward_splat_matrix(
      ldi_src, torch.tensor(mat), compose_layers=compose,
      compute_trg_disp=(path != 'stream'), trg_downsampling=0.5,
      bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50, path=path)
  img, wts = out[0], out[1]
  np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
  if path != 'stream':
    np.testing.assert_allclose(out[2].cpu().numpy(), want['disp'],
                               rtol=DSP_RTOL, atol=1e-7)
  mse = float(np.mean((img.cpu().numpy() - want['img'])**2))
  assert mse < 1e-12          # PSNR(build, oracle) > 120 dB


def test_general_pose_against_c_oracle(dev, ref_cpu):
  from lsi.geometry import ldi
  rs = np.random.RandomState(6)
  tex, disp, mat = _synth(rs, 3, 2, 128, 128, kitti=False, max_disp=1.0)
  disp = (0.28 + 0.22 * disp).astype(np.float32)
  mask = rs.rand(3, 2, 128, 128, 1).astype(np.float32)
  want = ref_cpu.forward_splat(tex, mask, disp, mat, 0.5, 0.2, 1.0, 50, True)
  ldi_src = [torch.tensor(x, device=dev) for x in (tex, mask, disp)]
  img, wts, dsp = ldi.forward_splat_matrix(
      ldi_src, torch.tensor(mat), compose_layers=True, compute_trg_disp=True,
      trg_downsampling=0.5, bg_layer_disp=0.2, max_disp=1.0, zbuf_scale=50)
  np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
  np.testing.assert_allclose(dsp.cpu().numpy(), want['disp'], rtol=DSP_RTOL,
                             atol=1e-7)
  with pytest.raises(RuntimeError, match='precondition'):
    ldi.forward_splat_matrix(ldi_src, torch.tensor(mat), path='rowband')
  # the gather kernel (what more than 16 composed layers fall back to; forced
  # here with the experiment bit) and the sweep kernel, per-layer outputs
  for compose in (True, False):
    want = ref_cpu.forward_splat(tex, mask, disp, mat, 0.5, 0.2, 1.0, 50, compose)
    for experiment in (0, 256):
      img, wts, dsp = ldi.forward_splat_matrix(
          ldi_src, torch.tensor(mat), compose_layers=compose,
          compute_trg_disp=True, trg_downsampling=0.5, bg_layer_disp=0.2,
          max_disp=1.0, zbuf_scale=50, path='tile', experiment=experiment)
      np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                                 atol=IMG_ATOL)
      np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
      np.testing.assert_allclose(dsp.cpu().numpy(), want['disp'], rtol=DSP_RTOL,
                                 atol=1e-7)


@pytest.mark.parametrize('path', ['atomic', 'rowband', 'stream', 'tile'])
def test_planar_nchw_inputs_need_no_copy(path, dev, ref_cpu):
  """A permuted NCHW conv output (planar RGB+disparity) renders identically."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(7)
  tex, disp, mat = _synth(rs, 2, 2, 64, 192)
  want = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50, True)
  nchw = torch.tensor(np.concatenate([tex, disp], -1), device=dev)
  nchw = nchw.permute(0, 1, 4, 2, 3).contiguous()      # L x B x 4 x H x W
  view = nchw.permute(0, 1, 3, 4, 2)                   # L x B x H x W x 4 view
  tex_v, disp_v = view[..., :3], view[..., 3:4]
  assert not tex_v.is_contiguous()
  img, wts = ldi.forward_splat_matrix(
      [tex_v, None, disp_v], torch.tensor(mat), trg_downsampling=0.5,
      bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50, path=path)
  np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)


@pytest.mark.parametrize('shape', [(1, 1, 1, 1), (1, 1, 2, 3), (2, 1, 7, 13),
                                   (1, 3, 33, 65)])
@pytest.mark.parametrize('path', ['atomic', 'rowband', 'tile'])
def test_ragged_and_tiny_shapes(shape, path, dev):
  from lsi.geometry import ldi
  nl, b, h, w = shape
  rs = np.random.RandomState(h * w)
  tex, disp, mat = _synth(rs, nl, b, h, w)
  want = O.forward_splat(tex, np.ones_like(disp), disp, mat, 1, 1e-3, 0.4, 50,
                         False)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  img, wts, dsp = ldi.forward_splat_matrix(
      ldi_src, torch.tensor(mat), compose_layers=False, compute_trg_disp=True,
      trg_downsampling=1, bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50,
      path=path)
  np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
  np.testing.assert_allclose(dsp.cpu().numpy(), want['disp'], rtol=DSP_RTOL,
                             atol=1e-7)


@pytest.mark.parametrize('shape,s', [((1, 1, 2, 4), 0.5), ((1, 1, 1, 4), 1),
                                     ((2, 1, 6, 12), 0.5), ((1, 2, 34, 260), 0.5),
                                     ((3, 1, 8, 2048), 0.5), ((1, 1, 8, 1028), 1),
                                     ((2, 1, 516, 8), 0.5)])
@pytest.mark.parametrize('compose', [True, False])
def test_stream_path_at_the_extremes_of_its_planner(shape, s, compose, dev):
  """One-row images, one-segment rows, eight segments per row, target tiles
  that only fit the LDS with one-row bands, more bands than CUs: the planner
  must find a launch for each and the result must be the oracle's."""
  from lsi.geometry import ldi
  nl, b, h, w = shape
  rs = np.random.RandomState(h * w + nl)
  tex, disp, mat = _synth(rs, nl, b, h, w)
  want = O.forward_splat(tex, np.ones_like(disp), disp, mat, s, 1e-3, 0.4, 50,
                         compose)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  img, wts = ldi.forward_splat_matrix(
      ldi_src, torch.tensor(mat), compose_layers=compose, trg_downsampling=s,
      bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50, path='stream')
  np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)


def test_non_integral_target_size_is_rejected(dev):
  from lsi.geometry import ldi
  tex = torch.rand(1, 1, 5, 7, 3, device=dev)
  disp = torch.rand(1, 1, 5, 7, 1, device=dev)
  with pytest.raises(ValueError, match='integral'):
    ldi.forward_splat_matrix([tex, None, disp], torch.eye(4).unsqueeze(0),
                             trg_downsampling=0.5)


def test_nonfinite_disparity_is_dropped_not_propagated(dev):
  from lsi.geometry import ldi
  rs = np.random.RandomState(9)
  tex, disp, mat = _synth(rs, 1, 1, 16, 24)
  disp[0, 0, 3, 4, 0] = np.nan
  disp[0, 0, 5, 6, 0] = np.inf
  for path in ('atomic', 'rowband', 'stream', 'tile'):
    img, wts = ldi.forward_splat_matrix(
        [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)],
        torch.tensor(mat), trg_downsampling=0.5, bg_layer_disp=1e-3,
        max_disp=0.4, zbuf_scale=50, path=path)
    assert bool(torch.isfinite(img).all()) and bool(torch.isfinite(wts).all())


@pytest.mark.parametrize('path', ['atomic', 'rowband', 'stream', 'tile'])
def test_full_size_properties(path, dev):
  """BASELINE config 3's per-GPU shard (4-layer 256x768, batch 4, s=0.5):
  size-independent properties instead of a CPU oracle."""
  from lsi.geometry import ldi
  gen = torch.Generator(device='cpu').manual_seed(11)
  nl, b, h, w = 4, 4, 256, 768
  tex = torch.rand(nl, b, h, w, 3, generator=gen).to(dev)
  disp = (0.4 * torch.rand(nl, b, h, w, 1, generator=gen)).to(dev)
  k = torch.tensor([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1.0]])
  k = k.expand(b, 3, 3)
  eye, t = torch.eye(3).expand(b, 3, 3), torch.tensor([[-0.532], [0], [0]]).expand(b, 3, 1)
  from lsi.geometry import projection
  mat = projection.forward_projection_matrix(k, k, eye, t)
  img, wts = ldi.forward_splat_matrix([tex, None, disp], mat,
                                      trg_downsampling=0.5, bg_layer_disp=1e-3,
                                      max_disp=0.4, zbuf_scale=50, path=path)
  # (1) mass conservation: sum of un-normalised weights = L*bg*P + all updates
  idx4, upd4 = ldi.project_indices(disp, None, mat, 0.5, 0.4, 50)
  from lsi import _C
  bg = _C.bg_weight(1e-3, 0.4, 50)
  total = float(wts.double().sum())
  want = float(upd4.double().sum()) + nl * bg * b * (h // 2) * (w // 2)
  assert abs(total - want) <= 1e-5 * want
  # (2) convexity: every rendered colour is a weighted mean of colours in [0,1]
  assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0 + 1e-6
  # (3) linearity in texture at fixed geometry: render(a*tex) == a*render(tex)
  #     up to the white background's share, checked through A = img*wts.
  img2, wts2 = ldi.forward_splat_matrix([0.5 * tex, None, disp], mat,
                                        trg_downsampling=0.5,
                                        bg_layer_disp=1e-3, max_disp=0.4,
                                        zbuf_scale=50, path=path)
  a1 = (img * wts - nl * bg).double()
  a2 = (img2 * wts2 - nl * bg).double()
  torch.testing.assert_close(wts2, wts, rtol=1e-5, atol=0)
  assert float((a2 - 0.5 * a1).abs().max()) <= 2e-5 * float(a1.abs().max())
  # (4) both kernel families agree with each other
  other = 'stream' if path == 'atomic' else 'atomic'
  img3, wts3 = ldi.forward_splat_matrix([tex, None, disp], mat,
                                        trg_downsampling=0.5,
                                        bg_layer_disp=1e-3, max_disp=0.4,
                                        zbuf_scale=50, path=other)
  torch.testing.assert_close(img3, img, rtol=0, atol=IMG_ATOL)
  torch.testing.assert_close(wts3, wts, rtol=WTS_RTOL, atol=0)


def test_identity_pose_reproduces_texture(dev):
  from lsi.geometry import ldi
  gen = torch.Generator().manual_seed(3)
  tex = torch.rand(1, 2, 32, 48, 3, generator=gen).to(dev)
  disp = torch.full((1, 2, 32, 48, 1), 0.3, device=dev)
  img, _ = ldi.forward_splat_matrix([tex, None, disp],
                                    torch.eye(4).expand(2, 4, 4),
                                    bg_layer_disp=1e-3, max_disp=0.4,
                                    zbuf_scale=50)
  assert float((img[0] - tex[0]).abs().max()) < 1e-5


@pytest.mark.parametrize('compose', [True, False])
@pytest.mark.parametrize('case', ['fs_general_L3_s05.npz', 'fs_kitti_L2_s05.npz'])
def test_backward_matches_autograd_of_the_op_graph(case, compose, dev):
  """lsi_splat_bwd vs torch autograd (fp64) over the reference's op graph."""
  import lsi_torch_ref as TR
  from lsi.geometry import ldi
  g = golden(case)
  s, bg, md, zb = _params(g)
  sl = (slice(None), slice(0, 1), slice(0, 16), slice(0, 24))
  tex, mask, disp = g['tex'][sl], g['mask'][sl], g['disp'][sl]
  mask = (0.5 + 0.5 * mask).astype(np.float32)
  mat = g['M'][:1]
  t64 = [torch.tensor(x, dtype=torch.float64, requires_grad=True)
         for x in (tex, mask, disp)]
  img, wts, _ = TR.forward_splat(t64[0], t64[1], t64[2],
                                 torch.tensor(mat, dtype=torch.float64), s, bg,
                                 md, zb, compose)
  gen = torch.Generator().manual_seed(0)
  cimg = torch.rand(img.shape, generator=gen, dtype=torch.float64)
  cwts = torch.rand(wts.shape, generator=gen, dtype=torch.float64) * 1e-3
  ((img * cimg).sum() + (torch.log(wts) * cwts).sum()).backward()

  t32 = [torch.tensor(x, device=dev, requires_grad=True)
         for x in (tex, mask, disp)]
  img_g, wts_g = ldi.forward_splat_matrix(
      t32, torch.tensor(mat), compose_layers=compose, trg_downsampling=s,
      bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  ((img_g * cimg.float().to(dev)).sum() +
   (torch.log(wts_g) * cwts.float().to(dev)).sum()).backward()
  for a, b_, name in zip(t32, t64, ('tex', 'mask', 'disp')):
    got = a.grad.cpu().double().numpy()
    want = b_.grad.numpy()
    scale = np.abs(want).max() + 1e-30
    bad = np.abs(got - want) > 2e-4 * scale + 1e-3 * np.abs(want)
    assert bad.mean() < 0.005, (name, bad.mean(), np.abs(got - want).max() / scale)


@pytest.mark.parametrize('compose', [True, False])
@pytest.mark.parametrize('seed', range(4))
def test_backward_under_general_projections(seed, compose, dev):
  """lsi_splat_bwd vs fp64 autograd for random projective matrices (rotation,
  shear, a normaliser that varies over the image, part of the image leaving the
  target), a mask input, disparities below 0 (zero weight) -- and, for the odd
  seeds, NaN / Inf disparities, whose pixels are dropped: their gradients must
  be exactly 0 and nothing else may turn non-finite."""
  import lsi_torch_ref as TR
  from lsi.geometry import ldi
  rs = np.random.RandomState(900 + seed)
  nl, b, h, w = 2, 1, 20, 28
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = rs.uniform(-0.05, 0.45, (nl, b, h, w, 1)).astype(np.float32)
  mask = rs.uniform(0.3, 1.0, (nl, b, h, w, 1)).astype(np.float32)
  m = np.eye(4) + rs.normal(0, 0.12, (4, 4))
  m[0, 2] += rs.uniform(-3, 3); m[1, 2] += rs.uniform(-3, 3)
  m[0, 3] = rs.uniform(-12, 12); m[1, 3] = rs.uniform(-8, 8)
  m[2] = [rs.normal(0, 0.004), rs.normal(0, 0.004), rs.uniform(0.8, 1.2),
          rs.normal(0, 0.2)]
  m[3] = [0, 0, 0, 1]
  mat = m.astype(np.float32)[None]
  dropped = np.zeros(disp.shape, bool)
  if seed % 2:
    dropped = rs.rand(*disp.shape) < 0.03
    disp[dropped] = np.where(rs.rand(int(dropped.sum())) < 0.5, np.nan, np.inf)
  s, bg, md, zb = 0.5, 1e-3, 0.4, 50.0
  clean = np.where(dropped, 0.0, disp)        # fp64 graph: dropped = masked out
  kill = np.where(dropped, 0.0, 1.0)
  t64 = [torch.tensor(x, dtype=torch.float64, requires_grad=True)
         for x in (tex, mask, clean)]
  img, wts, _ = TR.forward_splat(t64[0], t64[1] * torch.tensor(kill), t64[2],
                                 torch.tensor(mat, dtype=torch.float64), s, bg,
                                 md, zb, compose)
  gen = torch.Generator().manual_seed(seed)
  cimg = torch.rand(img.shape, generator=gen, dtype=torch.float64)
  cwts = torch.rand(wts.shape, generator=gen, dtype=torch.float64) * 1e-3
  ((img * cimg).sum() + (torch.log(wts) * cwts).sum()).backward()
  t32 = [torch.tensor(x, device=dev, requires_grad=True)
         for x in (tex, mask, disp)]
  img_g, wts_g = ldi.forward_splat_matrix(
      t32, torch.tensor(mat), compose_layers=compose, trg_downsampling=s,
      bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  ((img_g * cimg.float().to(dev)).sum() +
   (torch.log(wts_g) * cwts.float().to(dev)).sum()).backward()
  for a, b_, name in zip(t32, t64, ('tex', 'mask', 'disp')):
    got = a.grad.cpu().double().numpy()
    assert np.isfinite(got).all(), name
    assert not got[np.broadcast_to(dropped, got.shape)].any(), name
    want = np.where(np.broadcast_to(dropped, got.shape), 0.0, b_.grad.numpy())
    scale = np.abs(want).max() + 1e-30
    bad = np.abs(got - want) > 2e-4 * scale + 1e-3 * np.abs(want)
    assert bad.mean() < 0.01, (name, bad.mean(), np.abs(got - want).max() / scale)


# ---------------------------------------------------------------------------
# LSI_PATH_STREAM specifics: every internal route (monotone RMW, ranked RMW,
# exact slow path, window overflow) against the C oracle.
# ---------------------------------------------------------------------------
def _stream_case(rs, nl, b, h, w, kind, max_disp=0.4):
  tex, disp, mat = _synth(rs, nl, b, h, w)
  if kind == 'smooth':          # monotone lanes: plain RMW
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 0.2 + 0.1 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
    disp = np.broadcast_to(base[None, None, :, :, None],
                           (nl, b, h, w, 1)).astype(np.float32).copy()
  elif kind == 'iid':           # fold-overs everywhere: ranked RMW
    pass
  elif kind == 'outside':       # disparities beyond [0, max_disp]: slow path
    disp = rs.uniform(-0.5, 3.0, disp.shape).astype(np.float32) * max_disp
  elif kind == 'constant':      # long runs of identical cells at s = 0.5
    disp[:] = 0.123
  elif kind == 'pitch':         # camera pitch: normaliser varies per row
    k = np.array([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1]],
                 np.float32)
    k = np.broadcast_to(k, (b, 3, 3)).copy()
    k_t = k.copy()
    k_t[:, 0, 0] *= 1.1                                  # different focal
    rot = np.stack([_rot(0.04 * (i + 1), 0, 0) for i in range(b)]).astype(
        np.float32)
    t = np.broadcast_to(np.array([[-0.4], [0], [0]], np.float32),
                        (b, 3, 1)).copy()
    mat = O.forward_projection_matrix(k, k_t, rot, t)
    mat[:, 1, 0] = 0; mat[:, 2, 0] = 0; mat[:, 1, 3] = 0; mat[:, 2, 3] = 0
  return tex, disp.astype(np.float32), mat


@pytest.mark.parametrize('kind', ['smooth', 'iid', 'outside', 'constant',
                                  'pitch'])
@pytest.mark.parametrize('shape', [(2, 2, 64, 256, 0.5), (1, 1, 33, 260, 0.5),
                                   (3, 1, 16, 40, 1), (2, 1, 40, 516, 0.5)])
@pytest.mark.parametrize('compose', [True, False])
def test_stream_path_routes(kind, shape, compose, dev, ref_cpu):
  from lsi.geometry import ldi
  nl, b, h, w, s = shape
  if (h * s) != int(h * s):
    h += 1
  rs = np.random.RandomState(nl * 1000 + w)
  tex, disp, mat = _stream_case(rs, nl, b, h, w, kind)
  mask = (rs.rand(nl, b, h, w, 1) > 0.1).astype(np.float32) * \
      rs.rand(nl, b, h, w, 1).astype(np.float32)
  want = ref_cpu.forward_splat(tex, mask, disp, mat, s, 1e-3, 0.4, 50, compose)
  ldi_src = [torch.tensor(x, device=dev) for x in (tex, mask, disp)]
  for rows in (0, 1, 4):
    img, wts = ldi.forward_splat_matrix(
        ldi_src, torch.tensor(mat), compose_layers=compose, trg_downsampling=s,
        bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50, path='stream',
        band_rows=rows)
    np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                               atol=IMG_ATOL)
    np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)


GENERAL_STREAM = 0x40000000  # LsiSplatDesc.reserved: not the compact instance


@pytest.mark.parametrize('kind', ['smooth', 'iid', 'outside', 'constant',
                                  'clampy', 'nonfinite', 'zoom_out', 'zoom_in'])
@pytest.mark.parametrize('shape', [(2, 2, 64, 256), (4, 1, 24, 768),
                                   (1, 3, 10, 512), (3, 1, 130, 256),
                                   (9, 1, 12, 256), (2, 2, 20, 384),
                                   (3, 1, 16, 132), (2, 1, 8, 640)])
def test_stream_compact_instance_routes(kind, shape, dev, ref_cpu):
  """The compact STREAM instance (csrc/lsi_splat_stream2.hip: compose mode, no
  mask, unit normaliser; rows of 256-pixel segments, the last one possibly
  partial: W = 384, 132, 640) on every internal
  route, against the C oracle and against the general stream kernel, for
  several band heights (incl. bands that do not divide the image) and both
  merge exclusions."""
  from lsi.geometry import ldi
  nl, b, h, w = shape
  rs = np.random.RandomState(nl * 1000 + w + h)
  tex, disp, mat = _stream_case(rs, nl, b, h, w,
                                kind if kind in ('smooth', 'iid', 'outside',
                                                 'constant') else 'smooth')
  if kind == 'clampy':
    # side weights within a few ulps of the 1e-3 clamp thresholds of the row
    # weights 0.25 / 0.75: target x-coordinates k + 0.004 * (1 +- tiny)
    m03 = float(mat[0, 0, 3])
    xx = np.arange(w, dtype=np.float32)
    frac = rs.choice(np.array([0.004, 0.0040001, 0.0039999, 0.001333, 0.0013334,
                               0.0013333, 0.996, 0.9960001, 0.5, 0.25],
                              np.float32), size=(nl, b, h, w))
    # X = (x + .5 + d * m03) * .5 - .5  ->  d for X = round(x / 2) + frac
    want_x = np.floor(xx / 2)[None, None, None, :] + frac - 3.0
    disp = (((want_x + 0.5) * 2.0 - xx - 0.5) / m03)[..., None].astype(np.float32)
    disp = np.clip(disp, 1e-4, 0.5).astype(np.float32)
  if kind in ('zoom_out', 'zoom_in'):
    # target camera with other focal lengths / principal point: several (or a
    # fraction of a) source row per target row -- far more (fewer) source rows
    # per band than the launch planner assumes; still a unit normaliser
    k_s = np.array([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1]],
                   np.float32)
    k_t = k_s.copy()
    k_t[0, 0] *= 0.9
    k_t[1, 1] *= 0.23 if kind == 'zoom_out' else 2.1
    k_t[1, 2] += 3.3
    k_s = np.broadcast_to(k_s, (b, 3, 3)).copy()
    k_t = np.broadcast_to(k_t, (b, 3, 3)).copy()
    rot = np.broadcast_to(np.eye(3, dtype=np.float32), (b, 3, 3)).copy()
    t = np.broadcast_to(np.array([[-0.4], [0], [0]], np.float32), (b, 3, 1)).copy()
    mat = O.forward_projection_matrix(k_s, k_t, rot, t)
    assert np.all(mat[:, 2] == np.array([0, 0, 1, 0], np.float32))
  if kind == 'nonfinite':
    bad = rs.rand(*disp.shape) < 0.02
    disp[bad] = rs.choice(np.array([np.nan, np.inf, -np.inf, -1.0, 1e30],
                                   np.float32), size=int(bad.sum()))
  want = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50, True)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]

  def run(**kw):
    return ldi.forward_splat_matrix(
        ldi_src, torch.tensor(mat), compose_layers=True, trg_downsampling=0.5,
        bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50, path='stream', **kw)

  gen_img, gen_wts = run(experiment=GENERAL_STREAM)
  np.testing.assert_allclose(gen_img.cpu().numpy(), want['img'], rtol=0,
                             atol=IMG_ATOL)
  for rows in (0, 1, 2, 4, 16):
    for locks in (1, 2):  # reserved bits 18-19: 1 row locks, 2 cell locks
      img, wts = run(band_rows=rows, experiment=locks << 18)
      np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                                 atol=IMG_ATOL)
      np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
      assert float((img - gen_img).abs().max()) <= IMG_ATOL
      torch.testing.assert_close(wts, gen_wts, rtol=WTS_RTOL, atol=0)
  # units handed out 1 / 2 layers per ticket (reserved bits 12-15; what deep
  # LDIs get by default)
  for sub in (1, 2):
    img, wts = run(band_rows=4, experiment=(sub << 12) | (1 << 18))
    np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                               atol=IMG_ATOL)
    np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
  # both builds of the kernel by request: tune_threads <= 768 -> 12 waves x two
  # register sets, > 768 -> 16 waves x one (by itself the planner takes the
  # second for bands with fewer than two units per wave)
  for threads in (768, 1024, 320, 960):
    for rows in (0, 8):
      img, wts = run(band_rows=rows, threads=threads)
      np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                                 atol=IMG_ATOL)
      np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)


@pytest.mark.parametrize('kind', ['smooth', 'iid', 'outside', 'clampy'])
@pytest.mark.parametrize('shape', [(2, 2, 64, 256), (4, 1, 24, 768), (3, 1, 130, 256)])
def test_stream_compact_instance_both_outputs(kind, shape, dev, ref_cpu):
  """lsi_splat_fwd_both on the compact STREAM instance (one tile per layer in
  LDS, every item merged into its layer's tile, L + 1 views written by the
  epilogue): per-layer and composed views against the C oracle and against the
  general stream kernel, several band heights, both merge exclusions."""
  from lsi.geometry import ldi
  nl, b, h, w = shape
  rs = np.random.RandomState(nl * 100 + w + h)
  tex, disp, mat = _stream_case(rs, nl, b, h, w, kind if kind != 'clampy' else 'smooth')
  if kind == 'clampy':
    m03 = float(mat[0, 0, 3])
    xx = np.arange(w, dtype=np.float32)
    frac = rs.choice(np.array([0.004, 0.0040001, 0.0039999, 0.001333, 0.0013334,
                               0.996, 0.5], np.float32), size=(nl, b, h, w))
    want_x = np.floor(xx / 2)[None, None, None, :] + frac - 3.0
    disp = np.clip((((want_x + 0.5) * 2.0 - xx - 0.5) / m03)[..., None], 1e-4,
                   0.5).astype(np.float32)
  want_i = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50, False)
  want_c = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50, True)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]

  def run(**kw):
    return ldi.forward_splat_both(ldi_src, torch.tensor(mat), trg_downsampling=0.5,
                                  bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50,
                                  path='stream', **kw)

  gen = run(experiment=GENERAL_STREAM)
  for rows in (0, 1, 2, 4):
    for locks in (1, 2):
      img, wts, img_c, wts_c = run(band_rows=rows, experiment=locks << 18)
      np.testing.assert_allclose(img.cpu().numpy(), want_i['img'], rtol=0, atol=IMG_ATOL)
      np.testing.assert_allclose(wts.cpu().numpy(), want_i['wts'], rtol=WTS_RTOL)
      np.testing.assert_allclose(img_c.cpu().numpy(), want_c['img'], rtol=0, atol=IMG_ATOL)
      np.testing.assert_allclose(wts_c.cpu().numpy(), want_c['wts'], rtol=WTS_RTOL)
      assert float((img - gen[0]).abs().max()) <= IMG_ATOL
      assert float((img_c - gen[2]).abs().max()) <= IMG_ATOL


def test_stream_compact_instance_queue_overflow(dev, ref_cpu):
  """Every pixel outside its task's window (disparities far beyond max_disp):
  all corners go through the per-wave queue, which overflows many times per
  task and is emptied by the out-of-line flush."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(77)
  tex, disp, mat = _synth(rs, 2, 1, 16, 512)
  disp = rs.uniform(1.0, 1.6, disp.shape).astype(np.float32)  # 2.5x .. 4x max_disp
  want = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50, True)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  for locks in (1, 2):
    img, wts = ldi.forward_splat_matrix(
        ldi_src, torch.tensor(mat), trg_downsampling=0.5, bg_layer_disp=1e-3,
        max_disp=0.4, zbuf_scale=50, path='stream', experiment=locks << 18)
    np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                               atol=IMG_ATOL)
    np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)


@pytest.mark.parametrize('kind', ['smooth', 'iid'])
def test_stream_path_is_run_to_run_stable(kind, dev):
  """Default mode: windows are merged into the tile in arrival order, so runs
  agree to summation-order noise of a handful of terms.  LSI_DETERMINISTIC:
  merges in ticket order -- runs are bitwise identical."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(21)
  tex, disp, mat = _stream_case(rs, 2, 2, 128, 512, kind)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  def run(det):
    return ldi.forward_splat_matrix(ldi_src, torch.tensor(mat),
                                    trg_downsampling=0.5, bg_layer_disp=1e-3,
                                    max_disp=0.4, zbuf_scale=50, path='stream',
                                    deterministic=det)
  outs = [run(False) for _ in range(3)]
  for img, wts in outs[1:]:
    assert float((img - outs[0][0]).abs().max()) <= 1e-6
    torch.testing.assert_close(wts, outs[0][1], rtol=1e-6, atol=0)
  dets = [run(True) for _ in range(4)]
  for img, wts in dets[1:]:
    assert torch.equal(img, dets[0][0]) and torch.equal(wts, dets[0][1])
  # (the deterministic mode runs the general stream kernel -- compensated exp --
  # the default one the compact instance -- exp2 of a fused argument)
  torch.testing.assert_close(dets[0][0], outs[0][0], rtol=0, atol=4e-6)


def test_stream_path_rejects_what_it_cannot_render(dev):
  from lsi.geometry import ldi
  rs = np.random.RandomState(22)
  tex, disp, mat = _synth(rs, 1, 1, 16, 32)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  # (the disparity output: with the composed view only)
  with pytest.raises(RuntimeError, match='precondition'):
    ldi.forward_splat_matrix(ldi_src, torch.tensor(mat), compose_layers=False,
                             compute_trg_disp=True, path='stream')
  tex2, disp2, mat2 = _synth(rs, 1, 1, 16, 30)   # W % 4 != 0
  with pytest.raises(RuntimeError, match='precondition'):
    ldi.forward_splat_matrix(
        [torch.tensor(tex2, device=dev), None, torch.tensor(disp2, device=dev)],
        torch.tensor(mat2), path='stream')


@pytest.mark.gpu
@pytest.mark.parametrize('locks', [1, 2])         # 1: row locks, 2: cell locks
@pytest.mark.parametrize('mode', [1, 2])          # 1: halo bands, 2: exchange
@pytest.mark.parametrize('band_rows', [1, 4, 16])
def test_stream_bands_that_do_not_divide_the_image(band_rows, mode, locks, dev):
  """66 target rows in bands of 4 or 16: the last band is partial, in both band
  decompositions and with both merge exclusions (row locks / per-cell locks
  where the planner allows them), forced through the experiments field."""
  from lsi.geometry import ldi, projection
  gen = torch.Generator(device='cpu').manual_seed(29)
  nl, b, h, w = 2, 2, 132, 260
  tex = torch.rand(nl, b, h, w, 3, generator=gen).to(dev)
  disp = (0.4 * torch.rand(nl, b, h, w, 1, generator=gen)).to(dev)
  k = torch.tensor([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1.0]])
  k = k.expand(b, 3, 3)
  mat = projection.forward_projection_matrix(
      k, k, torch.eye(3).expand(b, 3, 3),
      torch.tensor([[-0.532], [0], [0]]).expand(b, 3, 1))
  kw = dict(trg_downsampling=0.5, bg_layer_disp=1e-3, max_disp=0.4,
            zbuf_scale=50)
  for compose in (True, False):
    ref_img, ref_wts = ldi.forward_splat_matrix(
        [tex, None, disp], mat, compose_layers=compose, path='atomic', **kw)
    for _ in range(2):
      img, wts = ldi.forward_splat_matrix(
          [tex, None, disp], mat, compose_layers=compose, path='stream',
          band_rows=band_rows, experiment=(mode << 16) | (locks << 18), **kw)
      torch.testing.assert_close(img, ref_img, rtol=0, atol=IMG_ATOL)
      torch.testing.assert_close(wts, ref_wts, rtol=WTS_RTOL, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize('band_rows', [0, 1, 2, 4, 8, 16])
def test_stream_band_exchange(band_rows, dev):
  """STREAM path: target rows shared by two row bands are combined through the
  workspace by whichever band finishes second.  Every band height must give the
  ATOMIC path's image, and repeated calls (the arrival counters are left zero
  by the kernel, not cleared per call) must give the same bits every time."""
  from lsi.geometry import ldi, projection
  gen = torch.Generator(device='cpu').manual_seed(23)
  nl, b, h, w = 2, 3, 128, 512
  tex = torch.rand(nl, b, h, w, 3, generator=gen).to(dev)
  disp = (0.4 * torch.rand(nl, b, h, w, 1, generator=gen)).to(dev)
  k = torch.tensor([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1.0]])
  k = k.expand(b, 3, 3)
  eye = torch.eye(3).expand(b, 3, 3)
  t = torch.tensor([[-0.532], [0], [0]]).expand(b, 3, 1)
  mat = projection.forward_projection_matrix(k, k, eye, t)
  kw = dict(trg_downsampling=0.5, bg_layer_disp=1e-3, max_disp=0.4,
            zbuf_scale=50)
  for compose in (True, False):
    ref_img, ref_wts = ldi.forward_splat_matrix(
        [tex, None, disp], mat, compose_layers=compose, path='atomic', **kw)
    first = None
    for rep in range(6):
      img, wts = ldi.forward_splat_matrix(
          [tex, None, disp], mat, compose_layers=compose, path='stream',
          band_rows=band_rows, experiment=2 << 16, **kw)  # exchange bands
      torch.testing.assert_close(img, ref_img, rtol=0, atol=IMG_ATOL)
      torch.testing.assert_close(wts, ref_wts, rtol=WTS_RTOL, atol=0)
      if first is None:
        first = (img.clone(), wts.clone())
      else:  # a + b == b + a: the order of arrival does not change the bits
        diff = float((img - first[0]).abs().max())
        assert diff <= 1e-6, (rep, diff)


@pytest.mark.gpu
def test_stream_workspace_contract_through_the_c_abi(dev):
  """lsi_hip.h: without LSI_WS_KEEP the library clears the arrival counters
  itself, so a dirty workspace is fine; with it, a workspace zero-filled once is
  reused by every later call; a too-small workspace is refused."""
  import ctypes
  from lsi import _C
  from lsi.geometry import ldi, projection
  gen = torch.Generator(device='cpu').manual_seed(5)
  nl, b, h, w = 2, 2, 64, 256
  tex = torch.rand(nl, b, h, w, 3, generator=gen).to(dev)
  disp = (0.4 * torch.rand(nl, b, h, w, 1, generator=gen)).to(dev)
  k = torch.tensor([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1.0]])
  k = k.expand(b, 3, 3)
  mat_host = projection.forward_projection_matrix(
      k, k, torch.eye(3).expand(b, 3, 3),
      torch.tensor([[-0.532], [0], [0]]).expand(b, 3, 1)).contiguous()
  mat = mat_host.to(dev)
  ref_img, ref_wts = ldi.forward_splat_matrix(
      [tex, None, disp], mat, trg_downsampling=0.5, bg_layer_disp=1e-3,
      max_disp=0.4, zbuf_scale=50, path='atomic')
  lib = _C.lib()
  ht, wt = h // 2, w // 2
  bg = _C.bg_weight(1e-3, 0.4, 50)

  def run(flags_extra, ws):
    d = ldi._desc(tex, None, disp, ht, wt, 0.5, 0.4, 50.0, bg,
                  _C.LSI_COMPOSE | flags_extra, 0, 4)
    ldi.select_path(d, mat_host, 'stream')
    d.reserved = 2 << 16            # experiments field: force the exchange bands
    img = torch.full((1, b, ht, wt, 3), -7.0, device=dev)
    wts = torch.full((1, b, ht, wt, 1), -7.0, device=dev)
    rc = lib.lsi_splat_fwd(ctypes.byref(d), _C.ptr(tex), _C.ptr(disp), None,
                           _C.ptr(mat), _C.ptr(img), _C.ptr(wts), None,
                           _C.ptr(ws), ws.numel() if ws is not None else 0,
                           _C.stream_ptr(dev))
    return rc, img, wts, d

  d0 = ldi._desc(tex, None, disp, ht, wt, 0.5, 0.4, 50.0, bg, _C.LSI_COMPOSE, 0)
  need = int(lib.lsi_splat_workspace_bytes(ctypes.byref(d0)))
  dirty = torch.full((need,), 0x5A, dtype=torch.uint8, device=dev)
  for _ in range(3):                 # library clears the counters per call
    rc, img, wts, _d = run(0, dirty)
    assert rc == 0
    torch.testing.assert_close(img, ref_img, rtol=0, atol=IMG_ATOL)
    torch.testing.assert_close(wts, ref_wts, rtol=WTS_RTOL, atol=0)
  kept = torch.zeros((need,), dtype=torch.uint8, device=dev)
  for _ in range(3):                 # caller keeps a zero-filled workspace
    rc, img, wts, _d = run(_C.LSI_WS_KEEP, kept)
    assert rc == 0
    torch.testing.assert_close(img, ref_img, rtol=0, atol=IMG_ATOL)
  torch.cuda.synchronize()
  # the counters are back to zero after every call
  ncount = 1 * b * ht * 4
  assert int(kept[:ncount].to(torch.int64).sum()) == 0
  rc, _, _, _ = run(0, torch.zeros((64,), dtype=torch.uint8, device=dev))
  assert rc == -3                    # LSI_EWORKSPACE
  rc, _, _, _ = run(0, None)
  assert rc == -2                    # LSI_ENULL


@pytest.mark.gpu
@pytest.mark.parametrize('path', ['tile', 'atomic'])
@pytest.mark.parametrize('seed', range(8))
def test_any_pose_paths_under_hostile_projections(seed, path, dev, ref_cpu):
  """Random projective matrices, far outside anything a camera pair produces:
  strong rotation and shear, the normaliser changing sign inside the image
  (points behind the camera), disparities outside [0, max_disp], NaN / Inf
  pixels.  The tile path bounds which source rows can reach a tile; that bound
  must stay a superset whatever M is.  Checked against the C oracle."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(100 + seed)
  nl, b, h, w = 2, 2, 40, 72
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = rs.uniform(-0.2, 1.3, (nl, b, h, w, 1)).astype(np.float32)
  mask = rs.rand(nl, b, h, w, 1).astype(np.float32)
  if seed % 2:
    bad = rs.rand(nl, b, h, w, 1) < 0.01
    disp[bad] = np.nan
    disp[rs.rand(nl, b, h, w, 1) < 0.005] = np.inf
  mat = np.zeros((b, 4, 4), np.float32)
  for i in range(b):
    m = np.eye(4) + rs.normal(0, 0.35, (4, 4))
    m[0, 2] += rs.uniform(-20, 20); m[1, 2] += rs.uniform(-10, 10)
    m[0, 3] = rs.uniform(-30, 30); m[1, 3] = rs.uniform(-30, 30)
    # normaliser n = m20 x + m21 y + m22 + m23 d: sign change inside the image
    m[2] = [rs.normal(0, 0.02), rs.normal(0, 0.03), rs.uniform(-0.5, 1.0),
            rs.normal(0, 0.8)]
    m[3] = [0, 0, 0, 1] if seed < 4 else m[3]
    mat[i] = m.astype(np.float32)
  s = 0.5
  for compose in (True, False):
    want = ref_cpu.forward_splat(tex, mask, disp, mat, s, 0.2, 1.0, 50, compose)
    ldi_src = [torch.tensor(x, device=dev) for x in (tex, mask, disp)]
    img, wts, dsp = ldi.forward_splat_matrix(
        ldi_src, torch.tensor(mat), compose_layers=compose,
        compute_trg_disp=True, trg_downsampling=s, bg_layer_disp=0.2,
        max_disp=1.0, zbuf_scale=50, path=path)
    np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL,
                               atol=1e-30)
    np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                               atol=IMG_ATOL)
    np.testing.assert_allclose(dsp.cpu().numpy(), want['disp'], rtol=DSP_RTOL,
                               atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(10))
def test_stream_path_under_hostile_row_uniform_projections(seed, dev, ref_cpu):
  """Random matrices that satisfy the stream precondition but nothing else a
  stereo rig would: mirrored or sheared x, target rows decreasing with the
  source row (the analytic row range must fall back to the scan), a normaliser
  that varies with the row, huge disparity shifts (windows overflow into the
  exact slow path), disparities outside [0, max_disp], NaN / Inf pixels."""
  from lsi import _C
  from lsi.geometry import ldi
  import ctypes
  rs = np.random.RandomState(500 + seed)
  nl, b, h, w = 3, 2, 48, 256
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = rs.uniform(-0.1, 0.6, (nl, b, h, w, 1)).astype(np.float32)
  mask = rs.rand(nl, b, h, w, 1).astype(np.float32)
  if seed % 3 == 0:
    disp[rs.rand(nl, b, h, w, 1) < 0.01] = np.nan
    disp[rs.rand(nl, b, h, w, 1) < 0.005] = np.inf
    disp[rs.rand(nl, b, h, w, 1) < 0.005] = -np.inf
  mat = np.zeros((b, 4, 4), np.float32)
  for i in range(b):
    sx = rs.choice([-1.0, 1.0]) if seed % 2 else 1.0
    sy = -1.0 if seed % 5 == 4 else 1.0
    m = np.zeros((4, 4))
    m[0] = [sx * rs.uniform(0.4, 1.8), rs.normal(0, 0.3), rs.uniform(-30, 30),
            rs.choice([-1, 1]) * rs.uniform(0, 150)]
    m[1] = [0, sy * rs.uniform(0.5, 1.6), rs.uniform(-8, 8) + (h if sy < 0 else 0), 0]
    m[2] = [0, rs.normal(0, 0.004), rs.uniform(0.7, 1.4), 0]
    m[3] = [0, 0, 0, 1] if seed < 5 else [rs.normal(0, 0.01), rs.normal(0, 0.01),
                                          rs.normal(0, 0.1), rs.uniform(0.5, 1.5)]
    mat[i] = m.astype(np.float32)
  s = 0.5
  probe = ldi._desc(torch.empty((nl, b, h, w, 3), device='meta'), None,
                    torch.empty((nl, b, h, w, 1), device='meta'), h // 2, w // 2,
                    s, 0.4, 50.0, 0.0, 0, 0)
  mat_t = torch.tensor(mat)  # (kept alive while its pointer is in use)
  ok = _C.lib().lsi_stream_ok(ctypes.byref(probe),
                              ctypes.c_void_p(mat_t.data_ptr()))
  assert ok, 'the generator must produce matrices the stream path accepts'
  for compose in (True, False):
    want = ref_cpu.forward_splat(tex, mask, disp, mat, s, 1e-3, 0.4, 50, compose)
    ldi_src = [torch.tensor(x, device=dev) for x in (tex, mask, disp)]
    for mode in (1, 2, 0):   # stream: halo bands, exchange bands; 0: rowband
      img, wts = ldi.forward_splat_matrix(
          ldi_src, torch.tensor(mat), compose_layers=compose, trg_downsampling=s,
          bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50,
          path='stream' if mode else 'rowband', experiment=mode << 16)
      np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL,
                                 atol=1e-30)
      np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                                 atol=IMG_ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('compose', [True, False])
def test_stream_task_table_is_refilled_in_chunks(compose, mode, dev, ref_cpu):
  """8-row bands at trg_downsampling 0.25 over 2304-pixel rows: 36 source rows
  x 9 segments (x layer groups) tasks per band against a task table held to
  64 entries (experiments field, bits 20+) -- the table is refilled mid-band,
  and again at the start of every later pass."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(77)
  nl, b, h, w = 2, 1, 96, 2304
  tex, disp, mat = _synth(rs, nl, b, h, w)
  s = 0.25
  want = ref_cpu.forward_splat(tex, np.ones_like(disp), disp, mat, s, 1e-3, 0.4,
                               50, compose)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  img, wts = ldi.forward_splat_matrix(
      ldi_src, torch.tensor(mat), compose_layers=compose, trg_downsampling=s,
      bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50, path='stream',
      band_rows=8, experiment=(mode << 16) | (4 << 20))
  np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
  np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                             atol=IMG_ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(1, 37, 36, 64), (2, 5, 52, 128), (1, 201, 8, 64)])
def test_stream_workgroup_placement_covers_every_band_once(shape, dev, ref_cpu):
  """Workgroup -> (band, batch element) goes through an XCD-aware bijective
  remap with reciprocal divisions: grids whose size is not a multiple of 8, one
  band per image, hundreds of batch elements -- every output must be written."""
  from lsi.geometry import ldi
  nl, b, h, w = shape
  rs = np.random.RandomState(b)
  tex, disp, mat = _synth(rs, nl, b, h, w)
  want = ref_cpu.forward_splat(tex, np.ones_like(disp), disp, mat, 0.5, 1e-3, 0.4,
                               50, True)
  ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  for rows in (0, 1, 2):
    img = torch.full((1, b, h // 2, w // 2, 3), float('nan'), device=dev)
    img, wts = ldi.forward_splat_matrix(
        ldi_src, torch.tensor(mat), trg_downsampling=0.5, bg_layer_disp=1e-3,
        max_disp=0.4, zbuf_scale=50, path='stream', band_rows=rows)
    np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)
    np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                               atol=IMG_ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['fs_kitti_L2_s05.npz', 'fs_general_L3_s05.npz',
                                  'fs_edge_L2_s05.npz'])
def test_both_outputs_from_one_sweep(case, dev):
  """lsi_splat_fwd_both: the per-layer and the composed rendering of one call
  (STREAM: second LDS tile; other paths: per-layer pass + layer sum) against
  the reference's goldens for compose_layers=False / True; its backward against
  the sum of the two separate backwards."""
  from lsi.geometry import ldi
  g = golden(case)
  s, bg, md, zb = _params(g)
  kw = dict(trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)

  def leaves():
    return [torch.tensor(g['tex'], device=dev, requires_grad=True),
            torch.tensor(g['mask'], device=dev, requires_grad=True),
            torch.tensor(g['disp'], device=dev, requires_grad=True)]

  la = leaves()
  img_i, wts_i, img_c, wts_c = ldi.forward_splat_both(la, torch.tensor(g['M']), **kw)
  np.testing.assert_allclose(img_i.detach().cpu().numpy(), g['indep_img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts_i.detach().cpu().numpy(), g['indep_wts'],
                             rtol=WTS_RTOL)
  np.testing.assert_allclose(img_c.detach().cpu().numpy(), g['compose_img'], rtol=0,
                             atol=IMG_ATOL)
  np.testing.assert_allclose(wts_c.detach().cpu().numpy(), g['compose_wts'],
                             rtol=WTS_RTOL)
  rs = np.random.RandomState(9)
  ci = torch.tensor(rs.rand(*img_i.shape).astype(np.float32), device=dev)
  cc = torch.tensor(rs.rand(*img_c.shape).astype(np.float32), device=dev)
  cw = torch.tensor(rs.rand(*wts_c.shape).astype(np.float32) * 1e-3, device=dev)
  ((img_i * ci).sum() + (img_c * cc).sum() + (wts_c * cw).sum()).backward()

  lb = leaves()
  a_i, _ = ldi.forward_splat_matrix(lb, torch.tensor(g['M']), compose_layers=False, **kw)
  a_c, w_c = ldi.forward_splat_matrix(lb, torch.tensor(g['M']), compose_layers=True, **kw)
  ((a_i * ci).sum() + (a_c * cc).sum() + (w_c * cw).sum()).backward()
  for x, y in zip(la, lb):
    scale = float(y.grad.abs().max()) + 1e-30
    assert float((x.grad - y.grad).abs().max()) <= 2e-5 * scale


@pytest.mark.gpu
def test_both_outputs_at_config3_shard_size(dev, ref_cpu):
  """4-layer 256x768 batch 4 (STREAM, second tile in LDS) vs the C oracle."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(15)
  tex, disp, mat = _synth(rs, 4, 4, 256, 768)
  src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  img_i, wts_i, img_c, wts_c = ldi.forward_splat_both(
      src, torch.tensor(mat), trg_downsampling=0.5, bg_layer_disp=1e-3,
      max_disp=0.4, zbuf_scale=50)
  for compose, (img, wts) in ((False, (img_i, wts_i)), (True, (img_c, wts_c))):
    want = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50,
                                 compose, want_disp=False)
    np.testing.assert_allclose(img.cpu().numpy(), want['img'], rtol=0,
                               atol=IMG_ATOL)
    np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=WTS_RTOL)


def _rectified_case(seed, nl, b, h, w, row_scale=1.0, row_shift=0.0, x_shift=0.0,
                    odd_values=False):
  """Inputs the STREAM path accepts with the SIMPLE bit: M rows 2, 3 =
  (0,0,1,0), (0,0,0,1), no x / disparity term in row 1."""
  rs = np.random.RandomState(seed)
  tex = rs.rand(nl, b, h, w, 3).astype(np.float32)
  disp = rs.uniform(-0.03, 0.45, (nl, b, h, w, 1)).astype(np.float32)
  dropped = np.zeros(disp.shape, bool)
  if odd_values:
    dropped = rs.rand(*disp.shape) < 0.02
    disp[dropped] = np.where(rs.rand(int(dropped.sum())) < 0.5, np.nan, np.inf)
  mats = []
  for _ in range(b):
    m = np.eye(4)
    m[0, 0] = rs.uniform(0.9, 1.1); m[0, 1] = rs.normal(0, 0.02)
    m[0, 2] = x_shift + rs.uniform(-2, 2); m[0, 3] = rs.uniform(-40, 40)
    m[1, 1] = row_scale; m[1, 2] = row_shift + rs.uniform(-1.5, 1.5)
    mats.append(m)
  return tex, disp, dropped, np.stack(mats).astype(np.float32)


@pytest.mark.parametrize('both', [False, True])
@pytest.mark.parametrize('kind', ['plain', 'odd', 'zoom_in', 'flipped',
                                  'leaving', 'rows8'])
def test_streamed_backward_matches_autograd_and_the_gather_kernel(kind, both, dev,
                                                                  monkeypatch):
  """lsi_splat_bwd[_both] on rectified pairs runs the streamed kernel
  (lsi_splat_bwd_stream.hip).  Checked against fp64 autograd of the reference's
  op graph and against the one-thread-per-pixel gather kernel
  (LSI_BWD_STREAM=0), for compose / per-layer / both outputs, NaN and Inf
  disparities (zero gradient), a vertical zoom whose canvas rows do not fit the
  LDS tile (global gathers), decreasing rows, and most of the image leaving
  the target."""
  import lsi_torch_ref as TR
  from lsi.geometry import ldi
  kw = dict(plain={}, odd=dict(odd_values=True),
            zoom_in=dict(row_scale=9.0, row_shift=-40.0),
            flipped=dict(row_scale=-1.0, row_shift=24.0),
            leaving=dict(x_shift=-300.0), rows8={})[kind]
  nl, b, h, w = 3, 2, 24, 256
  tex, disp, dropped, mat = _rectified_case(40 + len(kind), nl, b, h, w, **kw)
  if kind == 'rows8':
    monkeypatch.setenv('LSI_BWD_STREAM_ROWS', '8')
  s, bg, md, zb = 0.5, 1e-3, 0.4, 50.0
  gen = torch.Generator().manual_seed(7)

  def run(stream):
    monkeypatch.setenv('LSI_BWD_STREAM', '1' if stream else '0')
    t32 = [torch.tensor(x, device=dev, requires_grad=True) for x in (tex, disp)]
    g = torch.Generator().manual_seed(7)
    if both:
      img, wts, img_c, wts_c = ldi.forward_splat_both(
          [t32[0], None, t32[1]], torch.tensor(mat), trg_downsampling=s,
          bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
      outs = [img, wts, img_c, wts_c]
    else:
      outs = list(ldi.forward_splat_matrix(
          [t32[0], None, t32[1]], torch.tensor(mat), compose_layers=(kind != 'odd'),
          trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb))
    loss = 0
    coefs = []
    for o in outs:
      c = torch.rand(o.shape, generator=g, dtype=torch.float64)
      coefs.append(c)
      if o.shape[-1] == 3:
        loss = loss + (o * c.float().to(dev)).sum()
      else:
        loss = loss + (torch.log(o) * (1e-3 * c).float().to(dev)).sum()
    loss.backward()
    return [t.grad.cpu().double().numpy() for t in t32], coefs

  got, coefs = run(True)
  old, _ = run(False)
  for a, o, name in zip(got, old, ('tex', 'disp')):
    assert np.isfinite(a).all(), name
    scale = np.abs(o).max() + 1e-30
    assert np.abs(a - o).max() <= 2e-5 * scale, (name, np.abs(a - o).max() / scale)
  assert (got[1][dropped] == 0).all() and (got[0][dropped[..., 0]] == 0).all()

  # fp64 autograd of the op graph (dropped pixels: masked out)
  clean = np.where(dropped, 0.0, disp)
  kill = torch.tensor(np.where(dropped, 0.0, 1.0))
  t64 = [torch.tensor(x, dtype=torch.float64, requires_grad=True)
         for x in (tex, clean)]
  m64 = torch.tensor(mat, dtype=torch.float64)
  loss = 0
  modes = [False, True] if both else [kind != 'odd']
  k = 0
  for compose in modes:
    img, wts, _ = TR.forward_splat(t64[0], kill, t64[1], m64, s, bg, md, zb, compose)
    loss = loss + (img * coefs[k]).sum() + (torch.log(wts) * 1e-3 * coefs[k + 1]).sum()
    k += 2
  loss.backward()
  for a, b_, name in zip(got, t64, ('tex', 'disp')):
    want = b_.grad.numpy()
    scale = np.abs(want).max() + 1e-30
    bad = np.abs(a - want) > 2e-4 * scale + 1e-3 * np.abs(want)
    assert bad.mean() < 0.005, (name, bad.mean(), np.abs(a - want).max() / scale)


@pytest.mark.parametrize('w', [256, 512, 384, 68])
def test_packed_rgbd_pixels_render_like_separate_tensors(w, dev):
  """Colour and disparity given as views of ONE [L,B,H,W,4] buffer (what a
  channels-last conv head writes, sliced): the descriptor carries
  LSI_PACKED_RGBD, the compact STREAM instance and the streamed backward read
  whole RGBD pixels with 16-byte loads (also rows whose last 256-pixel segment
  is partial) -- outputs and gradients as for separate contiguous tensors."""
  from lsi import _C
  from lsi.geometry import ldi
  nl, b, h = 3, 2, 24
  rs = np.random.RandomState(77 + w)
  pred_np = rs.rand(nl, b, h, w, 4).astype(np.float32)
  pred_np[..., 3] = pred_np[..., 3] * 0.5 - 0.03
  _, _, _, mat = _rectified_case(5, nl, b, h, w)
  s, bg, md, zb = 0.5, 1e-3, 0.4, 50.0
  kw = dict(trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)

  def run(packed, both):
    pred = torch.tensor(pred_np, device=dev, requires_grad=True)
    tex, disp = pred[..., 0:3], pred[..., 3:4]
    if not packed:
      tex, disp = tex.contiguous(), disp.contiguous()
    d = ldi._desc(tex, None, disp, h // 2, w // 2, s, md, zb, 0.1, 0, 0)
    assert bool(d.flags & _C.LSI_PACKED_RGBD) == packed
    if both:
      outs = list(ldi.forward_splat_both([tex, None, disp], torch.tensor(mat), **kw))
    else:
      outs = list(ldi.forward_splat_matrix([tex, None, disp], torch.tensor(mat),
                                           compose_layers=True, **kw))
    g = torch.Generator().manual_seed(1)
    loss = 0
    for o in outs:
      c = torch.rand(o.shape, generator=g).to(dev)
      loss = loss + ((o if o.shape[-1] == 3 else torch.log(o) * 1e-3) * c).sum()
    loss.backward()
    return [o.detach().cpu().numpy() for o in outs], pred.grad.cpu().numpy()

  for both in (False, True):
    outs_p, g_p = run(True, both)
    outs_c, g_c = run(False, both)
    for a, c in zip(outs_p, outs_c):
      if a.shape[-1] == 3:
        assert np.abs(a - c).max() <= 4e-6
      else:
        np.testing.assert_allclose(a, c, rtol=1e-5, atol=0)   # (summation order)
    scale = np.abs(g_c).max()
    assert np.abs(g_p - g_c).max() <= 2e-5 * scale


def test_packed_rgbd_flag_is_verified_by_the_entry_points(dev):
  """LSI_PACKED_RGBD is the caller's statement about its pointers; a descriptor
  carrying it with tensors that are not one RGBD buffer is LSI_EINVAL."""
  import ctypes
  from lsi import _C
  from lsi.geometry import ldi
  nl, b, h, w = 1, 1, 8, 256
  pred = torch.rand(nl, b, h, w, 4, device=dev)
  tex, disp = pred[..., 0:3], pred[..., 3:4]
  d = ldi._desc(tex, None, disp, h // 2, w // 2, 0.5, 0.4, 50.0, 0.1,
                _C.LSI_COMPOSE, _C.LSI_PATH_TILE)
  assert d.flags & _C.LSI_PACKED_RGBD
  other = torch.rand(nl, b, h, w, 4, device=dev)[..., 3:4]   # same strides, other buffer
  img = torch.empty(1, b, h // 2, w // 2, 3, device=dev)
  wts = torch.empty(1, b, h // 2, w // 2, 1, device=dev)
  mat = torch.eye(4, device=dev).expand(b, 4, 4).contiguous()
  lib = _C.lib()
  ws_bytes = int(lib.lsi_splat_workspace_bytes(ctypes.byref(d)))
  ws = torch.zeros(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
  args = lambda dp: (ctypes.byref(d), _C.ptr(tex), _C.ptr(dp), None, _C.ptr(mat),
                     _C.ptr(img), _C.ptr(wts), None, _C.ptr(ws), ws_bytes,
                     _C.stream_ptr(dev))
  assert lib.lsi_splat_fwd(*args(other)) == -1   # LSI_EINVAL
  assert lib.lsi_splat_fwd(*args(disp)) == 0
  torch.cuda.synchronize()


@pytest.mark.parametrize('packed', [False, True])
@pytest.mark.parametrize('w', [256, 512, 384, 140])
def test_stream_renders_the_target_disparity_of_rectified_pairs(w, packed, dev):
  """forward_splat(compose_layers=True, compute_trg_disp=True) -- what the
  evaluation script asks for (ldi_pred_eval.py:340-353) -- on a rectified pair:
  the compact STREAM instance renders the composed view and, in a second
  launch with one tile per layer, each layer's splatted disparity over its own
  weight, maximum over the layers (ldi.py:147-180).  Against the NumPy oracle
  and the any-pose path."""
  from lsi import _C
  from lsi.geometry import ldi
  nl, b, h = 3, 2, 24
  rs = np.random.RandomState(31 + w)
  pred_np = rs.rand(nl, b, h, w, 4).astype(np.float32)
  pred_np[..., 3] = pred_np[..., 3] * 0.5 - 0.03
  bad = rs.rand(nl, b, h, w) < 0.01   # dropped pixels: NaN / Inf disparities
  pred_np[..., 3][bad] = np.where(rs.rand(int(bad.sum())) < 0.5, np.nan, np.inf)
  _, _, _, mat = _rectified_case(9, nl, b, h, w)
  s, bg, md, zb = 0.5, 1e-3, 0.4, 50.0
  pred = torch.tensor(pred_np, device=dev)
  tex, disp = pred[..., 0:3], pred[..., 3:4]
  if not packed:
    tex, disp = tex.contiguous(), disp.contiguous()
  d = ldi._desc(tex, None, disp, h // 2, w // 2, s, md, zb, 0.1,
                _C.LSI_COMPOSE | _C.LSI_WANT_DISP, 0)
  assert ldi.select_path(d, torch.tensor(mat), 'auto') == _C.LSI_PATH_STREAM
  kw = dict(compose_layers=True, compute_trg_disp=True, trg_downsampling=s,
            bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)
  got = [t.cpu().numpy() for t in
         ldi.forward_splat_matrix([tex, None, disp], torch.tensor(mat), **kw)]
  ref = [t.cpu().numpy() for t in
         ldi.forward_splat_matrix([tex, None, disp], torch.tensor(mat),
                                  path='tile', **kw)]
  # (oracle: the dropped pixels masked out, their disparities made finite)
  want = O.forward_splat(pred_np[..., 0:3], (~bad)[..., None].astype(np.float32),
                         np.where(bad[..., None], 0.1, pred_np[..., 3:4]), mat,
                         trg_downsampling=s,
                         bg_layer_disp=bg, max_disp=md, zbuf_scale=zb,
                         compose_layers=True)
  for other in (ref, (want['img'], want['wts'], want['disp'])):
    np.testing.assert_allclose(got[0], other[0], rtol=0, atol=IMG_ATOL)
    np.testing.assert_allclose(got[1], other[1], rtol=WTS_RTOL, atol=0)
    np.testing.assert_allclose(got[2], other[2], rtol=DSP_RTOL, atol=1e-7)


@pytest.mark.parametrize('shape', [(1, 1, 37, 256, 1.0), (2, 3, 18, 768, 0.5),
                                   (4, 1, 40, 512, 0.25), (2, 2, 8, 256, 0.5),
                                   (2, 2, 12, 384, 0.5), (3, 1, 10, 132, 0.5),
                                   (2, 1, 6, 640, 1.0)])
def test_streamed_backward_shapes_and_scales(shape, dev, monkeypatch):
  """The streamed backward against the gather kernel over target scales 1 /
  0.5 / 0.25, one to three row segments, band remainders (H not a multiple of
  the band height), one layer, several batch elements with their own matrices;
  compose and both-output modes."""
  from lsi.geometry import ldi
  nl, b, h, w, s = shape
  tex, disp, _, mat = _rectified_case(200 + h, nl, b, h, w)
  bg, md, zb = 1e-3, 0.4, 50.0
  kw = dict(trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)

  def run(stream, both):
    monkeypatch.setenv('LSI_BWD_STREAM', '1' if stream else '0')
    t32 = [torch.tensor(x, device=dev, requires_grad=True) for x in (tex, disp)]
    g = torch.Generator().manual_seed(3)
    if both:
      outs = list(ldi.forward_splat_both([t32[0], None, t32[1]], torch.tensor(mat), **kw))
    else:
      outs = list(ldi.forward_splat_matrix([t32[0], None, t32[1]], torch.tensor(mat),
                                           compose_layers=True, **kw))
    loss = 0
    for o in outs:
      c = torch.rand(o.shape, generator=g).to(dev)
      loss = loss + ((o if o.shape[-1] == 3 else torch.log(o) * 1e-3) * c).sum()
    loss.backward()
    return [t.grad.cpu().double().numpy() for t in t32]

  for both in (False, True):
    got, old = run(True, both), run(False, both)
    for a, o, name in zip(got, old, ('tex', 'disp')):
      assert np.isfinite(a).all(), name
      scale = np.abs(o).max() + 1e-30
      assert np.abs(a - o).max() <= 2e-5 * scale, (name, both, np.abs(a - o).max() / scale)


def test_differential_fuzz_of_the_round3_kernels(dev):
  """tools/fuzz_compact.py, 60 random rectified configurations: compact STREAM
  instance (compose / both / compose + disparity; separate tensors or RGBD
  pixels; folds, NaN / Inf, flipped rows, shifts past the window) against the
  any-pose path, streamed backward against the gather kernel."""
  import subprocess
  import sys
  from conftest import ROOT
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_compact.py'),
                        '60', '1234'], capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
  assert 'fuzz_compact: 60 cases' in out.stdout


def test_workspace_covers_every_path_on_tiny_targets(dev):
  """lsi_splat_workspace_bytes is enough for whichever path renders: a target
  of one cell needs more room for the any-pose path's disparity ranges than
  for canvases (found by tools/fuzz_tile.py)."""
  from lsi.geometry import ldi
  rs = np.random.RandomState(5)
  tex = torch.tensor(rs.rand(3, 2, 4, 4, 3).astype(np.float32), device=dev)
  disp = torch.tensor((0.1 + 0.5 * rs.rand(3, 2, 4, 4, 1)).astype(np.float32), device=dev)
  mat = torch.eye(4).expand(2, 4, 4).contiguous()
  outs = {}
  for path in ('tile', 'atomic'):
    outs[path] = ldi.forward_splat_matrix([tex, None, disp], mat, compose_layers=True,
                                          trg_downsampling=0.25, bg_layer_disp=0.05,
                                          max_disp=1.0, zbuf_scale=10.0, path=path)
  torch.testing.assert_close(outs['tile'][0], outs['atomic'][0], rtol=0, atol=IMG_ATOL)
  torch.testing.assert_close(outs['tile'][1], outs['atomic'][1], rtol=WTS_RTOL, atol=0)


@pytest.mark.parametrize('shape', [(2, 4, 24, 256), (3, 2, 10, 384), (1, 1, 6, 132),
                                   (15, 1, 8, 256)])
def test_per_layer_outputs_alone_on_the_compact_instance(shape, dev):
  """forward_splat(compose_layers=False) of a small rectified LDI (B * L <= 16)
  runs the tile-per-layer instance of the compact STREAM kernel without its
  composed view: against the NumPy oracle and the general stream kernel."""
  from lsi.geometry import ldi
  nl, b, h, w = shape
  tex, disp, _, mat = _rectified_case(300 + w + nl, nl, b, h, w)
  s, bg, md, zb = 0.5, 1e-3, 0.4, 50.0
  kw = dict(compose_layers=False, trg_downsampling=s, bg_layer_disp=bg, max_disp=md,
            zbuf_scale=zb)
  src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
  got = [t.cpu().numpy() for t in ldi.forward_splat_matrix(src, torch.tensor(mat), **kw)]
  gen = [t.cpu().numpy() for t in ldi.forward_splat_matrix(
      src, torch.tensor(mat), experiment=GENERAL_STREAM, **kw)]
  want = O.forward_splat(tex, np.ones_like(disp), disp, mat, trg_downsampling=s,
                         bg_layer_disp=bg, max_disp=md, zbuf_scale=zb,
                         compose_layers=False)
  assert got[0].shape == (nl, b, h // 2, w // 2, 3)
  for other in (gen, (want['img'], want['wts'])):
    np.testing.assert_allclose(got[0], other[0], rtol=0, atol=IMG_ATOL)
    np.testing.assert_allclose(got[1], other[1], rtol=WTS_RTOL, atol=0)


@pytest.mark.parametrize('both', [False, True])
@pytest.mark.parametrize('w', [256, 384])
def test_streamed_backward_with_a_mask(w, both, dev, monkeypatch):
  """A caller that passes the reference's mask tensor (ldi.py:145-146: pixel
  weight = soft z-buffer weight * mask) keeps the streamed backward: gradients
  w.r.t. textures, masks and disparities against fp64 autograd of the op graph
  and against the gather kernel."""
  import lsi_torch_ref as TR
  from lsi.geometry import ldi
  nl, b, h = 3, 2, 20
  tex, disp, _, mat = _rectified_case(77 + w, nl, b, h, w)
  rs = np.random.RandomState(w)
  mask = rs.uniform(0.2, 1.0, (nl, b, h, w, 1)).astype(np.float32)
  s, bg, md, zb = 0.5, 1e-3, 0.4, 50.0
  kw = dict(trg_downsampling=s, bg_layer_disp=bg, max_disp=md, zbuf_scale=zb)

  def run(stream):
    monkeypatch.setenv('LSI_BWD_STREAM', '1' if stream else '0')
    t32 = [torch.tensor(x, device=dev, requires_grad=True) for x in (tex, mask, disp)]
    g = torch.Generator().manual_seed(5)
    if both:
      outs = list(ldi.forward_splat_both(t32, torch.tensor(mat), **kw))
    else:
      outs = list(ldi.forward_splat_matrix(t32, torch.tensor(mat), compose_layers=True, **kw))
    loss, coefs = 0, []
    for o in outs:
      c = torch.rand(o.shape, generator=g, dtype=torch.float64)
      coefs.append(c)
      loss = loss + ((o if o.shape[-1] == 3 else torch.log(o) * 1e-3) * c.float().to(dev)).sum()
    loss.backward()
    return [t.grad.cpu().double().numpy() for t in t32], coefs

  got, coefs = run(True)
  old, _ = run(False)
  for a, o, name in zip(got, old, ('tex', 'mask', 'disp')):
    scale = np.abs(o).max() + 1e-30
    assert np.abs(a - o).max() <= 2e-5 * scale, (name, np.abs(a - o).max() / scale)
  t64 = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (tex, mask, disp)]
  m64 = torch.tensor(mat, dtype=torch.float64)
  loss, k = 0, 0
  for compose in ([False, True] if both else [True]):
    img, wts, _ = TR.forward_splat(t64[0], t64[1], t64[2], m64, s, bg, md, zb, compose)
    loss = loss + (img * coefs[k]).sum() + (torch.log(wts) * 1e-3 * coefs[k + 1]).sum()
    k += 2
  loss.backward()
  for a, b_, name in zip(got, t64, ('tex', 'mask', 'disp')):
    want = b_.grad.numpy()
    scale = np.abs(want).max() + 1e-30
    bad = np.abs(a - want) > 2e-4 * scale + 1e-3 * np.abs(want)
    assert bad.mean() < 0.005, (name, bad.mean(), np.abs(a - want).max() / scale)


def test_the_library_picks_the_kernel_build_from_the_field(dev, ref_cpu):
  """LsiSplatDesc.adapt / lsi_stream_adapt_state (include/lsi_hip.h): on a
  geometry where the planner alone takes 12 waves x two register sets, a folded
  (i.i.d.) disparity field moves the call to the 16-wave build after a few
  calls, a smooth field leaves it on 12 -- read from the kernel's own route
  counts into the CALLER's record (lsi.geometry.ldi owns one per device, stream
  and geometry; the library keeps no state); the rendering is the same either
  way."""
  import ctypes
  from lsi import _C
  from lsi.geometry import ldi
  nl, b, h, w = 2, 8, 256, 768
  states = {}
  for kind, md in (('smooth', 0.4), ('iid', 0.41)):    # (max_disp is part of the geometry key)
    rs = np.random.RandomState(5)
    tex, disp, mat = _stream_case(rs, nl, b, h, w, kind)
    t_tex, t_disp = torch.tensor(tex, device=dev), torch.tensor(disp, device=dev)
    mat_t = torch.tensor(mat)
    for _ in range(6):
      img, wts = ldi.forward_splat_matrix([t_tex, None, t_disp], mat_t, trg_downsampling=0.5,
                                          bg_layer_disp=1e-3, max_disp=md, zbuf_scale=50)
      torch.cuda.synchronize()
    desc = ldi._desc(t_tex, None, t_disp, h // 2, w // 2, 0.5, md, 50.0,
                     _C.bg_weight(1e-3, md, 50.0), _C.LSI_COMPOSE, 0)
    ldi.select_path(desc, mat_t, 'auto')
    assert desc.path == _C.LSI_PATH_STREAM
    states[kind] = ldi.stream_adapt(desc, dev).state()
    want = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, md, 50, True)
    assert float(np.abs(img.cpu().numpy() - want['img']).max()) <= 2e-5
    np.testing.assert_allclose(wts.cpu().numpy(), want['wts'], rtol=1e-4)
  assert states == {'smooth': 1, 'iid': 2}, states
