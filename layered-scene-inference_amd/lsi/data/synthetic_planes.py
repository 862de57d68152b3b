"""Procedural layered scenes rendered through the planar-warp path (a compact
counterpart of the reference's lsi/data/syntheticPlanes/{data,utils}.py).

The reference textures a box room and billboard objects with SUN / PASCAL
images (absent here) and renders the two views with
`homography.transform_plane_imgs` + `layers.compose` in a private TF session
(syntheticPlanes/data.py:309-415).  This generator keeps that rendering path --
`layers.planar_transform` (HIP bilinear gather) + `layers.compose` /
`compose_depth` -- and the reference's camera model (`sample_views`,
data.py:29-52; K = [[W,0,W/2],[0,H,H/2],[0,0,1]], data.py:548-557), with
procedural textures on fronto-parallel planes of a world camera.
"""
import math

import numpy as np
import torch

from lsi.geometry import layers
from lsi.nnutils import helpers as nn_helpers


def lookat_rotation(delta):
  """R with R*delta = (0, 0, z) (reference syntheticPlanes/utils.py:186-201)."""
  delta = np.reshape(delta, 3)
  theta = np.arctan2(delta[0], delta[2])
  phi = np.arcsin(delta[1] / np.linalg.norm(delta))
  rot_y = np.array([[math.cos(-theta), 0, math.sin(-theta)], [0, 1, 0],
                    [-math.sin(-theta), 0, math.cos(-theta)]])
  rot_x = np.array([[1, 0, 0], [0, math.cos(phi), -math.sin(phi)],
                    [0, math.sin(phi), math.cos(phi)]])
  return np.matmul(rot_x, rot_y)


def sample_views(nviews, rs):
  """Look-at cameras (reference syntheticPlanes/data.py:29-52): position
  x,y ~ U[-.5,.5], z = 0; look-at x,y ~ U[-.5,.5], z ~ U[3,3.5]."""
  out = []
  for _ in range(nviews):
    cam = np.array([rs.uniform(-0.5, 0.5), rs.uniform(-0.5, 0.5), 0.0])
    at = np.array([rs.uniform(-0.5, 0.5), rs.uniform(-0.5, 0.5),
                   rs.uniform(3.0, 3.5)])
    rot = lookat_rotation(at - cam)
    out.append((rot, -np.matmul(rot, cam.reshape(3, 1))))
  return out


def _texture(gen, h, w, device):
  lo = torch.rand((1, 3, h // 16 + 2, w // 16 + 2), generator=gen)
  tex = torch.nn.functional.interpolate(lo, size=(h, w), mode='bicubic',
                                        align_corners=False).clamp(0, 1)
  yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
  checker = (((yy // 16) + (xx // 16)) % 2).float() * 0.15
  return (tex[0].permute(1, 2, 0) * 0.85 + checker[..., None]).to(device)


class SceneGenerator(object):
  """n_obj billboard layers in front of a background plane, in a world camera
  at the origin; `forward(bs)` returns (img_src, img_trg, k_s, k_t, rot, t,
  disp_src, disp_trg) like the reference loader with synth_dl_eval_data."""

  def __init__(self, img_height, img_width, n_obj=2, device='cuda', seed=0):
    self.h, self.w, self.n_obj = img_height, img_width, n_obj
    self.device = torch.device(device)
    self.rs = np.random.RandomState(seed)
    self.gen = torch.Generator(device='cpu').manual_seed(seed)

  def _world_layers(self):
    h, w, dev = self.h, self.w, self.device
    imgs, masks, depths = [], [], []
    for i in range(self.n_obj):  # foreground billboards, nearest first
      z = self.rs.uniform(2.0 + 0.4 * i, 2.3 + 0.4 * i)
      cx, cy = self.rs.uniform(0.25, 0.75) * w, self.rs.uniform(0.3, 0.7) * h
      rx, ry = self.rs.uniform(0.1, 0.25) * w, self.rs.uniform(0.15, 0.3) * h
      yy, xx = torch.meshgrid(torch.arange(h, device=dev),
                              torch.arange(w, device=dev), indexing='ij')
      m = ((((xx - cx) / rx)**2 + ((yy - cy) / ry)**2) < 1).float()[..., None]
      imgs.append(_texture(self.gen, h, w, dev))
      masks.append(m)
      depths.append(z)
    imgs.append(_texture(self.gen, h, w, dev))       # back wall of the room
    masks.append(torch.ones((h, w, 1), device=dev))
    depths.append(3.5)
    return torch.stack(imgs), torch.stack(masks), depths

  def forward(self, bs):
    h, w, dev = self.h, self.w, self.device
    k = torch.tensor([[float(w), 0, w / 2.0], [0, float(h), h / 2.0],
                      [0, 0, 1.0]], device=dev)
    outs = {n: [] for n in ('src', 'trg', 'rot', 't', 'dsrc', 'dtrg')}
    pc = nn_helpers.pixel_coords(1, h, w, device=dev)
    for _ in range(bs):
      imgs, masks, depths = self._world_layers()
      nl = imgs.shape[0]
      n_hat = torch.tensor([[0.0, 0.0, 1.0]], device=dev).expand(nl, 1, 1, 3)
      a = torch.tensor(depths, device=dev).view(nl, 1, 1, 1) * -1.0
      (r_s, t_s), (r_t, t_t) = sample_views(2, self.rs)
      views = []
      for r_v, t_v in ((r_s, t_s), (r_t, t_t)):
        rv = torch.tensor(r_v, dtype=torch.float32, device=dev)[None]
        tv = torch.tensor(t_v, dtype=torch.float32, device=dev)[None]
        li, lm, ld = layers.planar_transform(
            imgs[:, None], masks[:, None], pc, k[None], k[None], rv, tv, n_hat,
            a)
        img = layers.compose(li, lm, ld, soft=False, min_disp=1e-6,
                             depth_softmax_temp=0.4)
        dsp = layers.compose_depth(lm, ld, min_disp=1e-6,
                                   depth_softmax_temp=0.4)
        views.append((img[0], dsp[0]))
      rot = np.matmul(r_t, r_s.T)                    # src -> trg
      t = t_t - np.matmul(rot, t_s)
      outs['src'].append(views[0][0]); outs['dsrc'].append(views[0][1])
      outs['trg'].append(views[1][0]); outs['dtrg'].append(views[1][1])
      outs['rot'].append(torch.tensor(rot, dtype=torch.float32))
      outs['t'].append(torch.tensor(t, dtype=torch.float32))
    kk = k.cpu().expand(bs, 3, 3).contiguous()
    return (torch.stack(outs['src']), torch.stack(outs['trg']), kk, kk.clone(),
            torch.stack(outs['rot']), torch.stack(outs['t']),
            torch.stack(outs['dsrc']), torch.stack(outs['dtrg']))
