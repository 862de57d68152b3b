"""KITTI loader mirror and TF-checkpoint name map on CPU:
a miniature directory tree with the data set's layout, and a round trip of the
network weights through TF variable names / layouts."""
import os
import types

import numpy as np
import pytest
import torch


def _write_png(path, h, w, seed):
  from PIL import Image
  os.makedirs(os.path.dirname(path), exist_ok=True)
  rs = np.random.RandomState(seed)
  Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(path)


def _calib_text(fx, fy, cx, cy, baseline2, baseline3):
  def p(b):
    return '%f 0 %f %f 0 %f %f 0 0 0 1 0' % (fx, cx, -fx * b, fy, cy)
  return ('calib_time: 09-Jan-2012 13:57:47\n'
          'P_rect_02: %s\nP_rect_03: %s\n' % (p(baseline2), p(baseline3)))


def test_raw_city_split_is_the_seeded_70_15_15_of_28_sequences(tmp_path):
  from lsi.data.kitti import data
  names = data.raw_city_sequences()
  assert len(names) == 28 and len(set(names)) == 28
  got = {}
  for split in ('train', 'val', 'test'):
    opts = types.SimpleNamespace(batch_size=1, kitti_data_root=str(tmp_path),
                                 kitti_dataset_variant='raw_city',
                                 data_split=split, img_height=8, img_width=16)
    got[split] = data.DataLoader(opts).split_sequences()
  assert [len(got[s]) for s in ('train', 'val', 'test')] == [20, 4, 4]
  assert sorted(got['train'] + got['val'] + got['test']) == sorted(names)
  # the reference's shuffle: RandomState(0).shuffle of the ordered list
  ref = list(names)
  np.random.RandomState(0).shuffle(ref)
  assert got['train'] == ref[:20] and got['val'] == ref[20:24]


def test_loader_on_a_miniature_raw_city_tree(tmp_path):
  from lsi.data.kitti import data
  root = tmp_path / 'kitti_raw'
  opts = types.SimpleNamespace(batch_size=2, kitti_data_root=str(tmp_path),
                               kitti_dataset_variant='raw_city',
                               data_split='train', img_height=16, img_width=48,
                               kitti_dl_disparities=False)
  probe = data.DataLoader(opts)
  seq = probe.split_sequences()[0]
  date = seq[:10]
  for cam in ('image_02', 'image_03'):
    for i in range(3):
      _write_png(str(root / date / (seq + '_sync') / cam / 'data' /
                     ('%010d.png' % i)), 32, 96, 10 * i + (cam == 'image_03'))
  (root / date / 'calib_cam_to_cam.txt').write_text(
      _calib_text(700.0, 710.0, 48.0, 16.0, 0.06, 0.59))
  dl = data.DataLoader(opts)
  assert len(dl.img_list_src) == 3
  assert all('image_03' in t and 'image_02' not in t for t in dl.img_list_trg)
  img_s, img_t, k_s, k_t, rot, trans = dl.forward(2)
  assert img_s.shape == (2, 16, 48, 3) and img_t.shape == (2, 16, 48, 3)
  assert img_s.dtype == np.float32 and 0.0 <= img_s.min() and img_s.max() <= 1.0
  # 96 x 32 -> 48 x 16: intrinsics scaled by 0.5 in both axes
  np.testing.assert_allclose(k_s[0], [[350.0, 0, 24.0], [0, 355.0, 8.0], [0, 0, 1]])
  np.testing.assert_allclose(rot[0], np.eye(3))
  # translation between the rectified cameras: -(0.59 - 0.06) along x
  np.testing.assert_allclose(trans[0].ravel(), [-0.53, 0, 0], atol=1e-9)
  # AREA resize by 2 = exact 2x2 box mean (computed in float32: no rounding to
  # 1/255 steps)
  from PIL import Image
  raw = np.asarray(Image.open(dl.src_image_names[0]), np.float32) / 255
  box = raw.reshape(16, 2, 48, 2, 3).mean(axis=(1, 3))
  assert np.abs(img_s[0] - box).max() <= 1e-6
  # every sample of an epoch exactly once
  seen = [dl._next_index() for _ in range(1)] + []
  assert len(set(seen)) == 1


def test_tf_checkpoint_names_and_round_trip():
  import ldi_enc_dec as script
  from lsi.nnutils import tf_checkpoint
  argv = ['--dataset', 'kitti', '--n_layers', '2', '--img_height', '128',
          '--img_width', '128', '--batch_size', '1']
  opts = script.apply_dataset_overrides(script.build_parser().parse_args(argv))
  torch.manual_seed(0)
  model = script.LdiNet(opts)
  names = dict((tf, (key, kind)) for tf, key, kind in
               tf_checkpoint.variable_map(model))
  # slim scopes of the reference (nets.py:244-348, 73-161)
  for want in ('encoder_decoder_unet/cnv1/weights',
               'encoder_decoder_unet/cnv7b/BatchNorm/beta',
               'encoder_decoder_unet/upcnv7/weights',
               'encoder_decoder_unet/icnv4/BatchNorm/moving_variance',
               'ldi_tex_disp/pixelwise_pred/upsample_1/decoder/upcnv3b/weights',
               'ldi_tex_disp/pixelwise_pred/upsample_0/pred_0/weights',
               'ldi_tex_disp/pixelwise_pred/upsample_0/pred_0/biases'):
    assert want in names, want
  # every parameter of the live model is mapped exactly once
  mapped = set(k for k, _ in names.values())
  params = set(k for k, _ in model.named_parameters())
  assert params <= mapped
  tf_vars = tf_checkpoint.export_tf_variables(model)
  assert tf_vars['encoder_decoder_unet/cnv1/weights'].shape == (7, 7, 3, 32)
  assert tf_vars['encoder_decoder_unet/upcnv7/weights'].shape == (4, 4, 512, 512)
  assert tf_vars['ldi_tex_disp/pixelwise_pred/upsample_0/pred_0/weights'].shape == (3, 3, 32, 4)
  torch.manual_seed(1)
  other = script.LdiNet(opts)
  # one variable missing, one with a wrong shape: skipped, like optimistic_restorer
  broken = dict(tf_vars)
  del broken['encoder_decoder_unet/cnv2/weights']
  broken['encoder_decoder_unet/cnv3/weights'] = np.zeros((3, 3, 1, 1), np.float32)
  loaded, skipped = tf_checkpoint.load_tf_variables(other, broken)
  assert sorted(skipped) == ['encoder_decoder_unet/cnv2/weights',
                             'encoder_decoder_unet/cnv3/weights']
  tf_checkpoint.load_tf_variables(other, tf_vars, strict=True)
  x = torch.rand(1, 128, 128, 3)
  with torch.no_grad():
    a, b = model.predict(x), other.predict(x)
  for u, v in zip(a, b):
    if u is not None:
      assert torch.equal(u, v)


def test_scene_generator_geometry_matches_the_reference():
  """tests/golden/scene_geometry.npz: outputs of the reference's own NumPy
  helpers (syntheticPlanes/utils.py:29-201, executed in place by
  oracle/make_goldens.py) and of its view sampler (data.py:29-52) with numpy's
  generator seeded: plane intrinsics, plane centres, canonical transforms, the
  five box planes, look-at rotations and sampled views."""
  import numpy as np
  from conftest import golden
  from lsi.data import synthetic_planes as sp
  g = golden('scene_geometry.npz')
  close = lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
  for row, want in zip(g['kmat_in'], g['kmat_out']):
    close(sp.dims2kmat(*row), want)
  for k, (sx, sy), want in zip(g['kmat_out'], g['rsz_scale'], g['rsz_out']):
    close(sp.resize_instrinsic(k, sx, sy), want)
  n = len(g['gc_pt'])
  for i in range(n):
    args = (g['gc_pt'][i], g['gc_x'][i], g['gc_y'][i], g['gc_wh'][i, 0],
            g['gc_wh'][i, 1])
    close(sp.get_centre(*args, g['gc_off'][i, 0], g['gc_off'][i, 1]), g['gc_out'][i])
    close(sp.get_centre(*args), g['gc_out_default'][i])
    rot, trans = sp.canonical_transform(g['gc_out'][i], g['gc_x'][i], g['gc_y'][i])
    close(rot, g['ct_rot'][i]); close(trans, g['ct_trans'][i])
    rot, trans = sp.canonical_transform(g['gc_out'][i], g['gc_x'][i], g['gc_y'][i],
                                        g['ct_init'][i])
    close(rot, g['ct_rot_init'][i]); close(trans, g['ct_trans_init'][i])
  for e, ext in enumerate(g['box_extent']):
    planes = sp.box_planes(ext)
    assert len(planes) == 5
    for k in ('pt', 'x_dir', 'y_dir'):
      close(np.stack([np.asarray(p[k], np.float64) for p in planes]),
            g['box%d_%s' % (e, k)])
    close(np.array([[p['w'], p['h'], p['off_x'], p['off_y']] for p in planes]),
          g['box%d_whoff' % e])
  for d, want in zip(g['lookat_delta'], g['lookat_rot']):
    close(sp.lookat_rotation(d), want)
  views = sp.sample_views(5, np.random.RandomState(int(g['views_seed'])))
  close(np.stack([v[0] for v in views]), g['views_rot'])
  close(np.stack([v[1] for v in views]), g['views_trans'])


def test_kitti_decode_and_area_resize_follow_tensorflow(tmp_path):
  """Reference kitti/data.py:247-266: tf.image.decode_image gives uint8 -- a
  16-bit PNG (the SPS-stereo disparity maps store disp * 256) keeps its HIGH
  byte, it is not saturated; the AREA resize weights every source pixel by the
  fraction of it inside the output cell (non-integer factors such as
  1242 x 375 -> 768 x 256)."""
  import numpy as np
  from PIL import Image
  from lsi.data.kitti import data
  rs = np.random.RandomState(0)
  raw16 = rs.randint(0, 65536, (30, 50)).astype(np.uint16)
  raw16[0, :4] = [300, 10240, 255, 256]
  path = str(tmp_path / 'disp16.png')
  Image.fromarray(raw16).save(path)
  dec = data.decode_png(path)
  assert dec.shape == (30, 50, 1)
  np.testing.assert_array_equal(dec[..., 0], (raw16 >> 8).astype(np.float32))
  assert list(dec[0, :4, 0]) == [1.0, 40.0, 0.0, 1.0]      # not 255, 255, ...
  # non-integer AREA factor against a brute-force coverage integral
  img = rs.rand(7, 10, 2).astype(np.float32)
  got = data.area_resize(img, 3, 4)

  def cover(n_in, n_out):
    m = np.zeros((n_out, n_in))
    sub = 1000                        # sub-samples per source pixel
    pos = (np.arange(n_in * sub) + 0.5) / sub
    cell = np.minimum((pos * n_out / n_in).astype(int), n_out - 1)
    for p, c in zip(pos, cell):
      m[c, int(p)] += 1
    return m / m.sum(1, keepdims=True)

  want = np.einsum('ai,bj,ijc->abc', cover(7, 3), cover(10, 4), img.astype(np.float64))
  np.testing.assert_allclose(got, want, atol=2e-3)
  assert abs(float(got.mean()) - float(img.mean())) < 1e-6   # mass preserved
  img8, orig = data._load_image(path, 15, 25, 1)              # disparity channel
  assert orig == (30, 50, 1) and img8.shape == (15, 25, 1)
  assert float(img8.max()) <= 1.0


def test_kitti_dataset_without_files_fails_loudly(tmp_path):
  """--dataset=kitti reads KITTI through lsi/data/kitti; a missing data root is
  an error unless procedural pairs are asked for explicitly."""
  import pytest
  import ldi_enc_dec as script
  argv = ['--dataset', 'kitti', '--n_layers', '2', '--img_height', '128',
          '--img_width', '128', '--batch_size', '1', '--cpu', 'true',
          '--kitti_data_root', str(tmp_path / 'nope'), '--checkpoint_dir',
          str(tmp_path)]
  opts = script.apply_dataset_overrides(script.build_parser().parse_args(argv))
  tr = script.Trainer(opts)
  with pytest.raises(FileNotFoundError, match='kitti_data_root'):
    tr.setup()
