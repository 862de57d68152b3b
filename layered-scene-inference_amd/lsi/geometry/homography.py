"""Image transformations via plane-induced homographies (mirror of the
reference's lsi/geometry/homography.py).  The 3x3 algebra is host-side torch;
the warp itself is the HIP bilinear gather (lsi.geometry.sampling)."""
import torch

from lsi.geometry import sampling
from lsi.geometry.projection import _inv3
from lsi.nnutils import helpers as nn_helpers

# 3x3 products in the oracle's sequential-k order (helpers.seq_matmul): the
# warped pixel coordinates feed floor() in the bilinear sampler
_mm = nn_helpers.seq_matmul


def inv_homography(k_s, k_t, rot, t, n_hat, a):
  """Inverse homography K_s (R^T + R^T t n R^T / (a - n R^T t)) K_t^-1
  (reference homography.py:28-51).  n_hat: [...] x 1 x 3, a: [...] x 1 x 1."""
  rot_t = nn_helpers.transpose(rot)
  k_t_inv = _inv3(k_t)
  denom = a - _mm(_mm(n_hat, rot_t), t)
  numerator = _mm(_mm(_mm(rot_t, t), n_hat), rot_t)
  return _mm(
      _mm(k_s, rot_t + nn_helpers.divide_safe(numerator, denom)),
      k_t_inv)


def inv_homography_dmat(k_t, rot, t, n_hat, a):
  """M with M (u, v, 1) = d_t (reference homography.py:54-73)."""
  rot_t = nn_helpers.transpose(rot)
  k_t_inv = _inv3(k_t)
  denom = a - _mm(_mm(n_hat, rot_t), t)
  return nn_helpers.divide_safe(
      -1 * _mm(_mm(n_hat, rot_t), k_t_inv), denom)


def normalize_homogeneous(pts_coords):
  """Homogeneous -> regular coordinates (reference homography.py:76-92)."""
  return nn_helpers.divide_safe(pts_coords[..., :-1], pts_coords[..., -1:])


def transform_plane_imgs(imgs, pixel_coords_trg, k_s, k_t, rot, t, n_hat, a):
  """Warps imgs by the per-plane homographies (reference homography.py:95-117).
  imgs: [...] x H_s x W_s x C; pixel_coords_trg: [...] x H_t x W_t x 3."""
  hom_t2s_planes = inv_homography(k_s, k_t, rot, t, n_hat, a)
  pixel_coords_t2s = nn_helpers.transform_pts(pixel_coords_trg, hom_t2s_planes)
  pixel_coords_t2s = normalize_homogeneous(pixel_coords_t2s)
  return sampling.bilinear_wrapper(imgs, pixel_coords_t2s)


def transform_plane_eqns(rot, t, n_hat, a):
  """Plane equations in the target frame (reference homography.py:120-136)."""
  rot_t = nn_helpers.transpose(rot)
  n_hat_t = _mm(n_hat, rot_t)
  a_t = a - _mm(n_hat, _mm(rot_t, t))
  return n_hat_t, a_t


def trg_disp_maps(pixel_coords_trg, k_t, rot, t, n_hat, a):
  """Per-pixel inverse depth of the planes in the target view (reference
  homography.py:139-156).  Returns [...] x H_t x W_t x 1."""
  dmats_t = inv_homography_dmat(k_t, rot, t, n_hat, a)  # [...] x 1 x 3
  prod = dmats_t.unsqueeze(-2) * pixel_coords_trg
  # three terms, added left to right (the oracle's reduce_sum over 3 elements)
  return (prod[..., 0:1] + prod[..., 1:2]) + prod[..., 2:3]
