"""KITTI stereo-pair loader (mirror of the reference's lsi/data/kitti/data.py,
raw_city / mview / odom variants) on NumPy + Pillow, no TF queues.

What is reproduced (reference lines in parentheses): the raw_city sequence list
(:39-75), the seeded 70/15/15 sequence split (:151-174), the file lists incl.
the excluded frame (:152, :168-172) and the SPS-stereo disparity names
(:181-192), calibration parsing (:227-245), the camera model of a pair --
intrinsics of P_rect_02 / P_rect_03, translation from the projection matrices'
fourth column, identity rotation, intrinsics rescaled to img_width x img_height
(:303-342) -- decoding to uint8 (16-bit PNGs keep their high byte, as
tf.image.decode_image does) and AREA resizing in float32 with exact fractional
pixel coverage (:247-266).

What is not: the order in which TF's three independently seeded shuffle queues
emit samples (:250-279) -- here one seeded permutation per epoch pairs each left
image with its own right image by construction.

The dataset itself is not available in the build container; the loader is
exercised on a miniature directory tree with the same layout
(tests/test_data_cpu.py).
"""
import fnmatch
import os

import numpy as np


def resize_instrinsic(intrinsic, scale_x, scale_y):
  """Intrinsics of an image resized by (scale_x, scale_y): the first row of K
  scales with x, the second with y (reference :32-36)."""
  return np.diag([scale_x, scale_y, 1.0]) @ np.asarray(intrinsic, np.float64)


# drive numbers of the city sequences of the KITTI raw data set, by date
_RAW_CITY = {
    '2011_09_26': (1, 2, 5, 9, 11, 13, 14, 17, 18, 48, 51, 56, 57, 59, 60, 84, 91,
                   93, 95, 96, 104, 106, 113, 117),
    '2011_09_28': (1, 2),
    '2011_09_29': (26, 71),
}


def raw_city_sequences():
  """Names of the city sequences of KITTI raw (reference :39-75), in the
  reference's order (by date, then drive number)."""
  return ['%s_drive_%04d' % (date, n)
          for date in sorted(_RAW_CITY) for n in _RAW_CITY[date]]


def _numbers(text):
  """The float array a calibration value spells, or None when any token is not
  a number (e.g. the `calib_time` stamp)."""
  try:
    return np.array([float(tok) for tok in text.split()], np.float64)
  except ValueError:
    return None


def read_calib_file(file_path):
  """KITTI calibration file -> {key: float array, or the text when the value is
  not numeric} (reference :227-245)."""
  data = {}
  with open(file_path, 'r') as f:
    for line in f:
      key, sep, value = line.partition(':')
      if not sep:
        continue
      value = value.strip()
      arr = _numbers(value) if value else None
      data[key] = value if arr is None else arr
  return data


def pair_cameras(calib_data, src_shape, trg_shape, h, w):
  """Camera model of a rectified pair (reference forward_instance :303-342).

  P_rect_0x = K [I | c] with K c in the fourth column: the camera centre offset
  is c = K^-1 P[:, 3] (K is upper triangular with K[2] = (0, 0, 1), so the
  reference's two explicit back-substitutions are exactly this solve).  Returns
  k_s, k_t (3x3, rescaled to w x h), rot (identity) and trans (3x1), the
  translation from the source (cam 2) to the target (cam 3) frame."""
  p2 = np.asarray(calib_data['P_rect_02'], np.float64).reshape(3, 4)
  p3 = np.asarray(calib_data['P_rect_03'], np.float64).reshape(3, 4)

  def offset(p):
    k, kt = p[:, :3], p[:, 3]
    z = kt[2]
    return np.array([(kt[0] - k[0, 2] * z) / k[0, 0],
                     (kt[1] - k[1, 2] * z) / k[1, 1], z])

  trans = offset(p3) - offset(p2)
  k_s = resize_instrinsic(p2[:, :3], w / src_shape[1], h / src_shape[0])
  k_t = resize_instrinsic(p3[:, :3], w / trg_shape[1], h / trg_shape[0])
  return k_s, k_t, np.eye(3), trans.reshape(3, 1)


def _area_matrix(n_in, n_out):
  """n_out x n_in weights of TF's AREA resize along one axis: output cell i is
  the mean of the input over [i * n_in / n_out, (i + 1) * n_in / n_out), every
  input pixel weighted by the fraction of it that lies inside."""
  scale = n_in / float(n_out)
  m = np.zeros((n_out, n_in), np.float64)
  for i in range(n_out):
    lo, hi = i * scale, (i + 1) * scale
    j0, j1 = int(np.floor(lo)), min(int(np.ceil(hi)), n_in)
    for j in range(j0, j1):
      m[i, j] = min(hi, j + 1) - max(lo, j)
    m[i] /= scale
  return m.astype(np.float32)


def area_resize(img, h, w):
  """tf.image.resize_images(AREA) of an H x W x C float image (exact
  fractional-coverage box filter, separable)."""
  img = np.asarray(img, np.float32)
  if img.shape[0] != h:
    img = np.tensordot(_area_matrix(img.shape[0], h), img, axes=(1, 0))
  if img.shape[1] != w:
    img = np.tensordot(_area_matrix(img.shape[1], w), img,
                       axes=(1, 1)).transpose(1, 0, 2)
  return np.ascontiguousarray(img, np.float32)


def decode_png(path):
  """tf.image.decode_image(...) with its default uint8 output, as float32 in
  [0, 255], H x W x C.  16-bit PNGs (the SPS-stereo disparity maps are
  disp * 256) keep their HIGH byte, as TF's conversion to uint8 does -- Pillow's
  convert('L') would saturate them at 255."""
  from PIL import Image  # pylint: disable=g-import-not-at-top
  with Image.open(path) as im:
    if im.mode in ('I;16', 'I;16B', 'I;16L', 'I'):
      arr = (np.asarray(im).astype(np.int64) >> 8).astype(np.float32)
    elif im.mode in ('L', 'P', '1', 'LA'):
      arr = np.asarray(im.convert('L'), np.float32)
    else:
      arr = np.asarray(im.convert('RGB'), np.float32)
  if arr.ndim == 2:
    arr = arr[:, :, None]
  return arr


def _load_image(path, h, w, nc=3):
  """Decoded image scaled to [0, 1], first nc channels, AREA-resized to h x w in
  float32 (reference img_queue_loader :247-266).  Returns (image, original
  shape)."""
  arr = decode_png(path) * np.float32(1.0 / 255)
  if arr.shape[2] < nc:
    raise ValueError('%s has %d channels, %d needed' % (path, arr.shape[2], nc))
  arr = arr[:, :, :nc]
  orig = (arr.shape[0], arr.shape[1], nc)
  return area_resize(arr, h, w), orig


class DataLoader(object):
  """KITTI data loading class (reference :78-385).

  opts needs: batch_size, kitti_data_root, kitti_dataset_variant ('raw_city',
  'mview' or 'odom'), data_split ('train' | 'val' | 'test'), img_height,
  img_width[, kitti_dl_disparities]."""

  def __init__(self, opts):
    self.opts = opts
    self.batch_size = opts.batch_size
    self.dataset_variant = opts.kitti_dataset_variant
    self.output_disparities = (
        self.dataset_variant == 'raw_city' and
        bool(getattr(opts, 'kitti_dl_disparities', False)) and
        opts.data_split != 'train')
    self.root_dir = opts.kitti_data_root
    if self.dataset_variant == 'odom':
      self.root_dir = os.path.join(self.root_dir, 'odometry', 'dataset',
                                   'sequences')
    elif self.dataset_variant == 'mview':
      self.root_dir = os.path.join(self.root_dir, 'stereo_multiview_2015')
      self.root_dir += '/training' if opts.data_split == 'train' else '/testing'
    elif self.dataset_variant == 'raw_city':
      self.root_dir = os.path.join(self.root_dir, 'kitti_raw')
    else:
      raise ValueError('unknown kitti_dataset_variant %r' % self.dataset_variant)
    self.h = opts.img_height
    self.w = opts.img_width
    self.init_img_names_seq_list()
    self.cam_calibration = None
    self._order, self._cursor = None, 0
    self._rng = np.random.RandomState(0)
    self.src_image_names = []

  # -- file lists -------------------------------------------------------------
  @staticmethod
  def _pngs(top):
    out = []
    for root, _, filenames in os.walk(top):
      for filename in fnmatch.filter(filenames, '*.png'):
        out.append(os.path.join(root, filename))
    return out

  def split_sequences(self):
    """raw_city: the seeded shuffle and 70 / 15 / 15 split of the sequence
    names (reference :151-166)."""
    seq_names = raw_city_sequences()
    rng = np.random.RandomState(0)
    rng.shuffle(seq_names)
    n_all = len(seq_names)
    n_train = int(round(0.7 * n_all))
    n_val = int(round(0.15 * n_all))
    split = self.opts.data_split
    if split == 'train':
      return seq_names[0:n_train]
    if split == 'val':
      return seq_names[n_train:(n_train + n_val)]
    if split == 'test':
      return seq_names[(n_train + n_val):n_all]
    return seq_names

  def init_img_names_seq_list(self):
    opts = self.opts
    self.img_list_src, self.img_list_trg, self.seq_id_list = [], [], []
    if self.dataset_variant == 'mview':
      self.img_list_src = sorted(self._pngs(os.path.join(self.root_dir, 'image_2')))
      for img_name in self.img_list_src:
        self.seq_id_list.append(int(img_name.split('/')[-1].split('_')[0]))
    elif self.dataset_variant == 'odom':
      data_seq = {'train': list(range(0, 7)) + list(range(12, 21)),
                  'val': list(range(7, 9)), 'test': list(range(9, 11))}[
                      opts.data_split]
      for seq_id in data_seq:
        seq_dir = os.path.join(self.root_dir, '{:02d}'.format(seq_id))
        for name in self._pngs(os.path.join(seq_dir, 'image_2')):
          self.img_list_src.append(name)
          self.seq_id_list.append(seq_id)
    else:  # raw_city
      exclude_img = '2011_09_26_drive_0117_sync/image_02/data/0000000074.png'
      for seq_id in self.split_sequences():
        seq_date = seq_id[0:10]
        seq_dir = os.path.join(self.root_dir, seq_date, '{}_sync'.format(seq_id))
        for name in self._pngs(os.path.join(seq_dir, 'image_02')):
          if exclude_img not in name:
            self.img_list_src.append(name)
            self.seq_id_list.append(seq_date)
    if self.dataset_variant == 'raw_city':
      self.img_list_trg = [f.replace('image_02', 'image_03')
                           for f in self.img_list_src]
      if self.output_disparities:
        self.img_list_disp_src = []
        for im_name in self.img_list_src:
          parts = im_name.split('/')
          self.img_list_disp_src.append(os.path.join(
              self.root_dir, 'spss_stereo_results', parts[-4],
              parts[-1][:-4] + '_left_initial_disparity.png'))
        self.img_list_disp_trg = [f.replace('left', 'right')
                                  for f in self.img_list_disp_src]
    else:
      self.img_list_trg = [f.replace('image_2', 'image_3')
                           for f in self.img_list_src]

  # -- calibration ------------------------------------------------------------
  def preload_calib_files(self):
    """One calibration per sequence / date (reference :194-225)."""
    self.cam_calibration = {}
    if self.dataset_variant == 'mview':
      top = os.path.join(self.root_dir, 'calib_cam_to_cam')
      for root, _, filenames in os.walk(top):
        for filename in fnmatch.filter(filenames, '*.txt'):
          seq_id = int(filename.split('.txt')[0])
          self.cam_calibration[seq_id] = read_calib_file(
              os.path.join(root, filename))
    elif self.dataset_variant == 'odom':
      for seq_id in sorted(set(self.seq_id_list)):
        calib = read_calib_file(os.path.join(
            self.root_dir, '{:02d}'.format(seq_id), 'calib.txt'))
        for key in ['P_rect_00', 'P_rect_01', 'P_rect_02', 'P_rect_03']:
          calib[key] = np.copy(calib[key.replace('_rect_0', '')])
        self.cam_calibration[seq_id] = calib
    else:
      for seq_date in sorted(set(self.seq_id_list)):
        self.cam_calibration[seq_date] = read_calib_file(os.path.join(
            self.root_dir, seq_date, 'calib_cam_to_cam.txt'))

  # -- batches ----------------------------------------------------------------
  def forward_instance(self, img_src, img_trg, src_shape, trg_shape, calib_data):
    """(img_s, img_t, k_s, k_t, rot, trans) of one pair (reference :303-342)."""
    k_s, k_t, rot, trans = pair_cameras(calib_data, src_shape, trg_shape,
                                        self.h, self.w)
    return img_src, img_trg, k_s, k_t, rot, trans

  def _next_index(self):
    n = len(self.img_list_src)
    if n == 0:
      raise RuntimeError('no KITTI images under %s' % self.root_dir)
    if self._order is None or self._cursor >= n:
      self._order, self._cursor = self._rng.permutation(n), 0
    i = int(self._order[self._cursor])
    self._cursor += 1
    return i

  def forward(self, bs):
    """bs instances: [img_s, img_t, k_s, k_t, rot, trans(, disp_s, disp_t)],
    each stacked along the batch axis (reference :344-385)."""
    if self.cam_calibration is None:
      self.preload_calib_files()
    ids = [self._next_index() for _ in range(bs)]
    self.src_image_names = [self.img_list_src[i] for i in ids]
    instances, disps_s, disps_t = [], [], []
    for i in ids:
      img_src, src_shape = _load_image(self.img_list_src[i], self.h, self.w)
      img_trg, trg_shape = _load_image(self.img_list_trg[i], self.h, self.w)
      instances.append(self.forward_instance(
          img_src, img_trg, src_shape, trg_shape,
          self.cam_calibration[self.seq_id_list[i]]))
      if self.output_disparities:
        disps_s.append(_load_image(self.img_list_disp_src[i], self.h, self.w, 1)[0])
        disps_t.append(_load_image(self.img_list_disp_trg[i], self.h, self.w, 1)[0])
    out = [np.stack([inst[k] for inst in instances]) for k in range(6)]
    if self.output_disparities:
      out.append(np.stack(disps_s))
      out.append(np.stack(disps_t))
    return out
