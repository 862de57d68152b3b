"""Procedural planar worlds rendered through the planar-warp path: counterpart of
the reference's lsi/data/syntheticPlanes/{data,utils}.py.

Reproduced from the reference (its lines in parentheses):
  * plane / camera math: `dims2kmat`, `get_centre`, `canonical_transform`,
    `box_planes`, `lookat_rotation` (utils.py:36-201), `sample_views`
    (data.py:29-52);
  * `WorldGenerator` (data.py:55-290): a box room x in [-.7,.7], y in [-.5,.5],
    z in [2,3.5] of which the first `n_box_planes` of (front, floor, ceiling,
    left wall, right wall) are instantiated, and `n_obj_min..n_obj_max`
    billboard objects standing on the floor (`random_obj_plane`), the unused
    object slots being fully transparent planes (`dummy_obj_plane`); every plane
    is a unit-distance fronto-parallel plane in its own canonical frame
    (n = (0,0,1), a = -1) with a rotation / translation into the world and an
    intrinsic matrix that maps the plane's extent onto its texture;
  * `Renderer` (data.py:293-513): every plane is warped into the view by its
    plane-induced homography (`homography.transform_plane_imgs`, HIP bilinear
    gather), the per-plane disparity maps come from `homography.trg_disp_maps`,
    and the layers are composed with `layers.compose(soft=False, min_disp=0.2,
    depth_softmax_temp=0.4)` / `layers.compose_depth` (HIP);
  * `DataLoader` (data.py:516-673): source camera = world frame, target camera
    from `sample_views(1)`, K = [[W,0,W/2],[0,H,H/2],[0,0,1]], optional AREA
    down-sampling by `synth_ds_factor`, optional ground truth (foreground /
    background disparities and background-only renderings).
Not reproducible here: the SUN / PASCAL-VOC texture images (absent; no network)
-- textures and object silhouettes are procedural.
"""
import math

import numpy as np
import torch

from lsi.geometry import homography, layers
from lsi.nnutils import helpers as nn_helpers


# ---------------------------------------------------------------------------
# plane and camera math (reference syntheticPlanes/utils.py:36-201)
# ---------------------------------------------------------------------------
def resize_instrinsic(intrinsic, scale_x, scale_y):
  out = np.copy(intrinsic)
  out[0, :] *= scale_x
  out[1, :] *= scale_y
  return out


def dims2kmat(w_plane, h_plane, w_tex, h_tex):
  """Intrinsics that map a w_plane x h_plane plane at Z = 1 onto a w_tex x h_tex
  texture (utils.py:36-51)."""
  dz = 1.0
  return np.array([[w_tex * dz / w_plane, 0, w_tex / 2],
                   [0, h_tex * dz / h_plane, h_tex / 2], [0, 0, 1]])


def _unit(v):
  v = np.reshape(np.asarray(v, np.float64), (3, 1))
  return v / np.linalg.norm(v)


def get_centre(pt, x_dir, y_dir, w, h, off_x=0.5, off_y=0.5):
  """Centre of a plane given a point at (off_x*w, off_y*h) from its top-left
  corner (utils.py:54-78)."""
  return (np.reshape(np.asarray(pt, np.float64), (3, 1)) +
          w * _unit(x_dir) * (0.5 - off_x) + h * _unit(y_dir) * (0.5 - off_y))


def canonical_transform(centre_s, x_dir, y_dir, trans_init=None):
  """(rot, trans) that take the canonical plane (centre (0,0,1), axes x, y) to
  centre_s with the given in-plane directions (utils.py:81-111)."""
  x_dir, y_dir = _unit(x_dir), _unit(y_dir)
  if trans_init is None:
    trans_init = np.array([0, 0, 1])
  trans_init = np.reshape(np.asarray(trans_init, np.float64), (3, 1))
  z_dir = np.cross(x_dir, y_dir, axisa=0, axisb=0, axisc=0)
  rot = np.concatenate([x_dir, y_dir, z_dir], axis=1)
  return rot, np.reshape(centre_s, (3, 1)) - np.matmul(rot, trans_init)


def box_planes(extent):
  """front, floor, ceiling, left wall, right wall of the box (x0, y0, z0, x1,
  y1, z1), each with a corner point, in-plane axes and size (utils.py:114-178)."""
  x0, y0, z0, x1, y1, z1 = extent
  ex, ey, ez = np.array([1, 0, 0]), np.array([0, 1, 0]), np.array([0, 0, 1])

  def plane(pt, xd, yd, w, h):
    return {'pt': np.array(pt, np.float64), 'x_dir': xd, 'y_dir': yd, 'w': w,
            'h': h, 'off_x': 0, 'off_y': 0}

  front = plane([x0, y0, z1], ex, ey, x1 - x0, y1 - y0)
  ceil = plane([x0, y0, z1], ex, -1 * ez, x1 - x0, z1 - z0)
  floor = plane([x0, y1, z1], ex, -1 * ez, x1 - x0, z1 - z0)
  wall_l = plane([x0, y0, z0], ez, ey, z1 - z0, y1 - y0)
  wall_r = plane([x1, y0, z0], ez, ey, z1 - z0, y1 - y0)
  return [front, floor, ceil, wall_l, wall_r]


def lookat_rotation(delta):
  """R with R*delta = (0, 0, z) (utils.py:186-201)."""
  delta = np.reshape(delta, 3)
  theta = np.arctan2(delta[0], delta[2])
  phi = np.arcsin(delta[1] / np.linalg.norm(delta))
  rot_y = np.array([[math.cos(-theta), 0, math.sin(-theta)], [0, 1, 0],
                    [-math.sin(-theta), 0, math.cos(-theta)]])
  rot_x = np.array([[1, 0, 0], [0, math.cos(phi), -math.sin(phi)],
                    [0, math.sin(phi), math.cos(phi)]])
  return np.matmul(rot_x, rot_y)


def sample_views(nviews, rs):
  """Look-at cameras (data.py:29-52): position x,y ~ U[-.5,.5], z = 0; look-at
  x,y ~ U[-.5,.5], z ~ U[3,3.5].  Returns [(rot, trans)] world -> camera."""
  out = []
  for _ in range(nviews):
    cam = np.array([rs.uniform(-0.5, 0.5), rs.uniform(-0.5, 0.5), 0.0])
    at = np.array([rs.uniform(-0.5, 0.5), rs.uniform(-0.5, 0.5),
                   rs.uniform(3.0, 3.5)])
    rot = lookat_rotation(at - cam)
    out.append((rot, -np.matmul(rot, cam.reshape(3, 1))))
  return out


# ---------------------------------------------------------------------------
# procedural stand-ins for the SUN / PASCAL textures
# ---------------------------------------------------------------------------
def _texture(gen, h, w):
  """Smooth colour field + checker pattern, h x w x 3 in [0, 1]."""
  lo = torch.rand((1, 3, h // 16 + 2, w // 16 + 2), generator=gen)
  tex = torch.nn.functional.interpolate(lo, size=(h, w), mode='bicubic',
                                        align_corners=False).clamp(0, 1)
  yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
  checker = (((yy // 16) + (xx // 16)) % 2).float() * 0.15
  return (tex[0].permute(1, 2, 0) * 0.85 + checker[..., None]).numpy()


def _silhouette(rs, h, w):
  """Object alpha mask (the reference's PASCAL object crops fill their texture
  with an alpha channel): a union of two ellipses touching the bottom edge."""
  yy, xx = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing='ij')
  m = np.zeros((h, w), np.float32)
  for _ in range(2):
    cx, cy = rs.uniform(0.35, 0.65) * w, rs.uniform(0.45, 0.7) * h
    rx, ry = rs.uniform(0.2, 0.42) * w, rs.uniform(0.28, 0.45) * h
    m = np.maximum(m, ((((xx - cx) / rx)**2 + ((yy - cy) / ry)**2) < 1)
                   .astype(np.float32))
  m[int(0.8 * h):, int(0.35 * w):int(0.65 * w)] = 1.0   # stands on the floor
  return m


# ---------------------------------------------------------------------------
# world generator (reference data.py:55-290)
# ---------------------------------------------------------------------------
EXTENT = [-0.7, -0.5, 2.0, 0.7, 0.5, 3.5]   # data.py:236-243


class WorldGenerator(object):
  """Random box-like worlds with textured side planes and a few (almost)
  fronto-parallel foreground objects standing on the floor."""

  def __init__(self, h=400, w=400, n_obj_max=4, n_obj_min=1, n_box_planes=5,
               seed=0):
    self.h, self.w = h, w
    self.n_obj_min, self.n_obj_max = n_obj_min, n_obj_max
    self.n_box_planes = n_box_planes
    self.bs = n_box_planes + n_obj_max
    assert self.bs > 0  # at least box planes or billboards
    self.rs = np.random.RandomState(seed)
    self.gen = torch.Generator(device='cpu').manual_seed(seed)

  def dummy_obj_plane(self, z_max):
    """A default plane at z = z_max for fully transparent slots (data.py:118-142)."""
    return {'pt': np.array([0, 0, z_max]), 'x_dir': np.array([1, 0, 0]),
            'y_dir': np.array([0, 1, 0]), 'w': 1, 'h': 1, 'off_x': 0.5,
            'off_y': 0.5}

  def random_obj_plane(self, extent, aspect, fixed_plane=None):
    """A plane inside the box for an object of (perturbed) aspect ratio h / w,
    bottom edge on the floor, frontal orientation (data.py:144-202)."""
    rs = self.rs
    aspect = np.exp(np.log(aspect) + rs.uniform(-0.2, 0.2))
    h_box, w_box = extent[4] - extent[1], extent[3] - extent[0]
    d_box = extent[5] - extent[2]
    if aspect < h_box / w_box:  # width is the bottleneck
      w_obj = rs.uniform(0.4, 0.6) * w_box
      h_obj = w_obj * aspect
    else:                       # height is the bottleneck
      h_obj = rs.uniform(0.4, 0.6) * h_box
      w_obj = h_obj / aspect
    w_frac = w_obj / w_box
    if fixed_plane is not None:
      centre_x = extent[0] + 0.25 * w_box + 0.25 * fixed_plane * w_box
      centre_z = extent[2] + 0.2 * fixed_plane * d_box
    else:
      centre_x = (extent[0] + w_box * rs.uniform(0.1, 0.9 - w_frac) +
                  0.5 * w_obj)
      centre_z = extent[2] + 0.5 * rs.uniform(0, extent[5] - extent[2])
    return {'pt': np.array([centre_x, extent[4], centre_z]),
            'x_dir': np.array([1, 0, 0]), 'y_dir': np.array([0, 1, 0]),
            'w': w_obj, 'h': h_obj, 'off_x': 0.5, 'off_y': 1}

  def forward(self):
    """rot_w2s, t_w2s [bs,3,3 / bs,3,1] (canonical plane frame -> world), k_w
    [bs,3,3], n_hat_w [bs,1,3], a_w [bs,1,1], imgs_w [bs,h,w,3], masks_w
    [bs,h,w,1] (data.py:204-290)."""
    bs, h, w, nb = self.bs, self.h, self.w, self.n_box_planes
    imgs_w = np.ones((bs, h, w, 3), np.float32)
    masks_w = np.zeros((bs, h, w, 1), np.float32)
    n_hat_w = np.tile(np.array([[[0.0, 0.0, 1.0]]]), (bs, 1, 1))
    a_w = np.tile(np.array([[[-1.0]]]), (bs, 1, 1))
    planes = box_planes(EXTENT)[0:nb]
    for i in range(len(planes)):
      masks_w[i] = 1
      imgs_w[i] = _texture(self.gen, h, w)
    n_obj = self.rs.randint(self.n_obj_min, self.n_obj_max + 1)
    for ix in range(self.n_obj_max):
      if ix < n_obj:
        imgs_w[ix + nb] = _texture(self.gen, h, w)
        masks_w[ix + nb, :, :, 0] = _silhouette(self.rs, h, w)
        aspect_tex = self.rs.uniform(0.7, 1.6)   # crop height / width
        planes.append(self.random_obj_plane(EXTENT, aspect_tex, fixed_plane=ix))
      else:
        planes.append(self.dummy_obj_plane(EXTENT[5]))
    t_w2s = np.zeros((bs, 3, 1))
    rot_w2s = np.zeros((bs, 3, 3))
    k_w = np.zeros((bs, 3, 3))
    for ix, pl in enumerate(planes):
      centre = get_centre(pl['pt'], pl['x_dir'], pl['y_dir'], pl['w'], pl['h'],
                          off_x=pl['off_x'], off_y=pl['off_y'])
      rot_w2s[ix], t_w2s[ix] = canonical_transform(centre, pl['x_dir'],
                                                   pl['y_dir'])
      # (the reference passes (h, w) for (w_tex, h_tex), data.py:288 -- the
      # same thing for its square textures; the texture here may not be square)
      k_w[ix] = dims2kmat(pl['w'], pl['h'], w, h)
    return rot_w2s, t_w2s, k_w, n_hat_w, a_w, imgs_w, masks_w


# ---------------------------------------------------------------------------
# renderer (reference data.py:293-513)
# ---------------------------------------------------------------------------
MIN_DISP, SOFTMAX_TEMP = 2e-1, 0.4     # data.py:363-365


class Renderer(object):
  """Renders a planar world into a camera: homography warp of every plane
  (HIP bilinear gather), analytic per-plane disparities, hard composition."""

  def __init__(self, n_imgs, h=400, w=400, ds_factor=1, device='cuda'):
    self.n_imgs, self.h, self.w, self.ds = n_imgs, h, w, ds_factor
    self.device = torch.device(device)
    self.pixel_coords = nn_helpers.pixel_coords(1, h, w, device=self.device)[0]
    self.world = None
    self.k_s = self.k_t = None

  def set_cameras(self, k_s, k_t):
    self.k_s, self.k_t = [self._t(k) for k in (k_s, k_t)]

  def _t(self, a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32,
                           device=self.device)

  def set_world(self, rot_w2s, t_w2s, k_w, n_hat_w, a_w, imgs_w, masks_w):
    self.world = [self._t(x) for x in (rot_w2s, t_w2s, k_w, n_hat_w, a_w,
                                       imgs_w, masks_w)]

  def _warp(self, rot_v, trans_v, k_v):
    """Per-plane images, masks and disparity maps in the view (rot_v, trans_v):
    world -> view composed with every plane's canonical -> world transform
    (`w2t_rendering`, data.py:368-383)."""
    rot_w2s, t_w2s, k_w, n_hat_w, a_w, imgs_w, masks_w = self.world
    n = self.n_imgs
    rv = self._t(rot_v)[None].expand(n, 3, 3)
    tv = self._t(trans_v)[None].expand(n, 3, 1)
    rot_w2t = nn_helpers.seq_matmul(rv, rot_w2s)
    t_w2t = tv + nn_helpers.seq_matmul(rv, t_w2s)
    pc = self.pixel_coords[None].expand(n, self.h, self.w, 3)
    k_vv = k_v[None].expand(n, 3, 3)
    imgs = homography.transform_plane_imgs(imgs_w, pc, k_w, k_vv, rot_w2t, t_w2t,
                                           n_hat_w, a_w)
    masks = homography.transform_plane_imgs(masks_w, pc, k_w, k_vv, rot_w2t,
                                            t_w2t, n_hat_w, a_w)
    dmaps = homography.trg_disp_maps(pc, k_vv, rot_w2t, t_w2t, n_hat_w, a_w)
    return imgs, masks, dmaps, rot_w2t, t_w2t

  def _down(self, x):
    if self.ds == 1:
      return x
    from lsi.loss import loss  # pylint: disable=g-import-not-at-top
    return loss.area_downsample(x[None], self.h // self.ds, self.w // self.ds)[0]

  def render_planes(self, rot_v, trans_v, k_v=None):
    """The composed image of the view, H/ds x W/ds x 3 (data.py:385-391, 474-485)."""
    imgs, masks, dmaps, _, _ = self._warp(rot_v, trans_v,
                                          self.k_t if k_v is None else k_v)
    return self._down(layers.compose(imgs, masks, dmaps, soft=False,
                                     min_disp=MIN_DISP,
                                     depth_softmax_temp=SOFTMAX_TEMP))

  def render_disps(self, rot_v, trans_v, k_v=None):
    """(foreground, background) disparity maps of the view (data.py:393-405,
    487-499): compose_depth without / with bg_layer."""
    _, masks, dmaps, _, _ = self._warp(rot_v, trans_v,
                                       self.k_t if k_v is None else k_v)
    fg = layers.compose_depth(masks, dmaps, bg_layer=False, min_disp=MIN_DISP,
                              depth_softmax_temp=SOFTMAX_TEMP)
    bg = layers.compose_depth(masks, dmaps, bg_layer=True, min_disp=MIN_DISP,
                              depth_softmax_temp=SOFTMAX_TEMP)
    return self._down(fg), self._down(bg)

  def plane_geometry(self, rot_v, trans_v):
    """Plane normals and offsets in the view's frame (data.py:380-381, 501-513)."""
    rot_w2s, t_w2s, _, n_hat_w, a_w, _, _ = self.world
    n = self.n_imgs
    rv = self._t(rot_v)[None].expand(n, 3, 3)
    tv = self._t(trans_v)[None].expand(n, 3, 1)
    rot_w2t = nn_helpers.seq_matmul(rv, rot_w2s)
    t_w2t = tv + nn_helpers.seq_matmul(rv, t_w2s)
    return homography.transform_plane_eqns(rot_w2t, t_w2t, n_hat_w, a_w)


# ---------------------------------------------------------------------------
# data loader (reference data.py:516-673)
# ---------------------------------------------------------------------------
class DataLoader(object):
  """WorldGenerator + Renderer.  opts: img_height, img_width, n_obj_max,
  n_obj_min, n_box_planes, synth_ds_factor, synth_dl_eval_data."""

  def __init__(self, opts, device='cuda', seed=0):
    self.opts = opts
    self.output_gt = bool(getattr(opts, 'synth_dl_eval_data', False))
    ds = int(getattr(opts, 'synth_ds_factor', 1))
    self.ds = ds
    img_width, img_height = opts.img_width * ds, opts.img_height * ds
    self.n_box_planes = getattr(opts, 'n_box_planes', 5)
    n_obj_max = getattr(opts, 'n_obj_max', 4)
    self.generator = WorldGenerator(h=img_height, w=img_width,
                                    n_obj_max=n_obj_max,
                                    n_obj_min=getattr(opts, 'n_obj_min', 1),
                                    n_box_planes=self.n_box_planes, seed=seed)
    self.renderer = Renderer(self.n_box_planes + n_obj_max, h=img_height,
                             w=img_width, ds_factor=ds, device=device)
    f_x, f_y = float(img_width), float(img_height)
    self.k_s = np.array([[f_x, 0, f_x / 2.0], [0, f_y, f_y / 2.0], [0, 0, 1]])
    self.k_t = np.copy(self.k_s)
    self.renderer.set_cameras(self.k_s, self.k_t)
    self.rs = np.random.RandomState(seed + 1)

  def forward_instance(self):
    gen, ren, nb = self.generator, self.renderer, self.n_box_planes
    world = gen.forward()
    ren.set_world(*world)
    rot_src, trans_src = np.eye(3), np.zeros((3, 1))
    rot_trg, trans_trg = sample_views(1, self.rs)[0]
    img_s = ren.render_planes(rot_src, trans_src, ren.k_s)
    img_t = ren.render_planes(rot_trg, trans_trg, ren.k_t)
    rot = np.matmul(rot_trg, rot_src.T)
    trans = trans_trg - np.matmul(rot, trans_src)
    k_s = resize_instrinsic(self.k_s, 1.0 / self.ds, 1.0 / self.ds)
    k_t = resize_instrinsic(self.k_t, 1.0 / self.ds, 1.0 / self.ds)
    out = [img_s, img_t, torch.tensor(k_s, dtype=torch.float32),
           torch.tensor(k_t, dtype=torch.float32),
           torch.tensor(rot, dtype=torch.float32),
           torch.tensor(trans, dtype=torch.float32)]
    if self.output_gt:
      disp_s_fg, _ = ren.render_disps(rot_src, trans_src, ren.k_s)
      disp_t_fg, _ = ren.render_disps(rot_trg, trans_trg, ren.k_t)
      n_hat, a = ren.plane_geometry(rot_src, trans_src)
      # the room alone: object planes made transparent (data.py:606-619)
      masks_bg = np.copy(world[6])
      masks_bg[nb:] = 0
      ren.set_world(*(list(world[:6]) + [masks_bg]))
      img_s_bg = ren.render_planes(rot_src, trans_src, ren.k_s)
      disp_s_bg, _ = ren.render_disps(rot_src, trans_src, ren.k_s)
      img_t_bg = ren.render_planes(rot_trg, trans_trg, ren.k_t)
      disp_t_bg, _ = ren.render_disps(rot_trg, trans_trg, ren.k_t)
      out += [n_hat, a, disp_s_fg, disp_s_bg, disp_t_fg, disp_t_bg, img_s_bg,
              img_t_bg]
    return out

  def forward(self, bs):
    """bs instances, every output stacked along the batch axis: img_s, img_t,
    k_s, k_t, rot, trans[, n_hat, a, disp_s_fg, disp_s_bg, disp_t_fg, disp_t_bg,
    img_s_bg, img_t_bg]."""
    inst = [self.forward_instance() for _ in range(bs)]
    return [torch.stack([inst[b][i] for b in range(bs)])
            for i in range(len(inst[0]))]


class SceneGenerator(object):
  """Convenience wrapper: (img_src, img_trg, k_s, k_t, rot, t, disp_src,
  disp_trg) with foreground ground-truth disparities."""

  def __init__(self, img_height, img_width, n_obj=2, device='cuda', seed=0):
    import types  # pylint: disable=g-import-not-at-top
    opts = types.SimpleNamespace(img_height=img_height, img_width=img_width,
                                 n_obj_max=n_obj, n_obj_min=min(1, n_obj),
                                 n_box_planes=5, synth_ds_factor=1,
                                 synth_dl_eval_data=True)
    self.loader = DataLoader(opts, device=device, seed=seed)

  def forward(self, bs):
    out = self.loader.forward(bs)
    return (out[0], out[1], out[2], out[3], out[4], out[5], out[8], out[10])
