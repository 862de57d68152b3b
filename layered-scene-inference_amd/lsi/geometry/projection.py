"""Perspective projection utilities (mirror of lsi/geometry/projection.py)."""
import torch

from lsi.nnutils import helpers as nn_helpers


# Small-matrix product in the order pinned by the parity oracle: a projection
# matrix computed here is bit-identical to the oracle's.
_seq_matmul = nn_helpers.seq_matmul


def _inv3(k_mat):
  """3x3 inverse evaluated in fp64 and rounded once to the input dtype (the
  fp32 LU of LAPACK / Eigen / rocSOLVER differ from each other in the last
  ulp; a correctly rounded inverse is the reproducible choice)."""
  return torch.linalg.inv(k_mat.to(torch.float64)).to(k_mat.dtype)


def pad_intrinsic(k_mat):
  """[...] x 3 x 3 intrinsics -> [...] x 4 x 4 (reference projection.py:27-46)."""
  lead = tuple(k_mat.shape[:-2])
  out = torch.zeros(lead + (4, 4), dtype=k_mat.dtype, device=k_mat.device)
  out[..., :3, :3] = k_mat
  out[..., 3, 3] = 1
  return out


def pad_extrinsic(rot_mat, trans_mat):
  """[R t; 0 1] as [...] x 4 x 4 (reference projection.py:49-68)."""
  lead = tuple(rot_mat.shape[:-2])
  out = torch.zeros(lead + (4, 4), dtype=rot_mat.dtype, device=rot_mat.device)
  out[..., :3, :3] = rot_mat
  out[..., :3, 3:4] = trans_mat
  out[..., 3, 3] = 1
  return out


def forward_projection_matrix(k_s, k_t, rot, t):
  """src pixel (u, v, 1, disp) -> trg frame: pad(K_t) [R t; 0 1] pad(K_s^-1)
  (reference projection.py:71-86)."""
  k_s_inv = _inv3(k_s)
  return _seq_matmul(pad_intrinsic(k_t),
                     _seq_matmul(pad_extrinsic(rot, t), pad_intrinsic(k_s_inv)))


def inverse_projection_matrix(k_s, k_t, rot, t):
  """trg pixel -> src frame (reference projection.py:89-106)."""
  k_t_inv = _inv3(k_t)
  rot_inv = nn_helpers.transpose(rot)
  t_inv = -1 * _seq_matmul(rot_inv, t)
  return _seq_matmul(pad_intrinsic(k_s),
                     _seq_matmul(pad_extrinsic(rot_inv, t_inv),
                                 pad_intrinsic(k_t_inv)))


def disocclusion_mask(disps_src, disps_trg, pixel_coords_src, src2trg_mat,
                      thresh=1e-2):
  """1 where a source pixel is dis-occluded in the target view (reference
  projection.py:109-150).  disps: B x H x W x 1; returns B x H x W x 1."""
  from lsi.geometry import sampling  # pylint: disable=g-import-not-at-top
  _, h_t, w_t, _ = disps_trg.shape
  coords_src = torch.cat([pixel_coords_src, disps_src], dim=-1)
  coords_trg = nn_helpers.transform_pts(coords_src, src2trg_mat)
  uv, normalizer, disps_src2trg = torch.split(coords_trg, [2, 1, 1], dim=-1)
  uv = nn_helpers.divide_safe(uv, normalizer)
  disps_src2trg = nn_helpers.divide_safe(disps_src2trg, normalizer)
  u, v = uv[..., 0:1], uv[..., 1:2]
  trunc = ((u > w_t).float() + (v > h_t).float() + (u < 0).float() +
           (v < 0).float())
  trunc = (trunc > 0).float()
  sampled = sampling.bilinear_wrapper(disps_trg, uv, compose=True)
  disocc = (torch.abs(disps_src2trg - sampled) > thresh).float()
  return (1 - trunc) * disocc
