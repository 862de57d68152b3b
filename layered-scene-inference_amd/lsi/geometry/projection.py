"""Perspective projection utilities (mirror of lsi/geometry/projection.py)."""
import torch

from lsi.nnutils import helpers as nn_helpers


# Small-matrix product in the order pinned by the parity oracle: a projection
# matrix computed here is bit-identical to the oracle's.
_seq_matmul = nn_helpers.seq_matmul


def _inv3(k_mat):
  """3x3 inverse evaluated in fp64 (adjugate / determinant, every operation
  rounded in fp64) and rounded once to the input dtype: the fp32 LU of LAPACK /
  Eigen / rocSOLVER differ from each other in the last ulp, a correctly rounded
  inverse is the reproducible choice.  Same operation order as
  lsi_projection_matrices (csrc/lsi_host.hip): the two agree bit for bit."""
  a = k_mat.to(torch.float64)
  a0, a1, a2 = a[..., 0, 0], a[..., 0, 1], a[..., 0, 2]
  a3, a4, a5 = a[..., 1, 0], a[..., 1, 1], a[..., 1, 2]
  a6, a7, a8 = a[..., 2, 0], a[..., 2, 1], a[..., 2, 2]
  c00 = a4 * a8 - a5 * a7
  c01 = a5 * a6 - a3 * a8
  c02 = a3 * a7 - a4 * a6
  det = a0 * c00 + a1 * c01 + a2 * c02
  r = 1.0 / det
  inv = torch.stack([
      c00 * r, (a2 * a7 - a1 * a8) * r, (a1 * a5 - a2 * a4) * r,
      c01 * r, (a0 * a8 - a2 * a6) * r, (a2 * a3 - a0 * a5) * r,
      c02 * r, (a1 * a6 - a0 * a7) * r, (a0 * a4 - a1 * a3) * r], dim=-1)
  return inv.reshape(a.shape).to(k_mat.dtype)


def _host_matrices(k_s, k_t, rot, t, inverse):
  """lsi_projection_matrices (csrc/lsi_host.hip): the whole batch in one call,
  for B x 3 x 3 / B x 3 x 1 fp32 cameras on the host -- the same values as the
  torch ops below, without a dozen small-tensor dispatches.  None when the
  arguments are not of that form."""
  args = (k_s, k_t, rot, t)
  if not all(isinstance(x, torch.Tensor) and x.device.type == 'cpu' and
             x.dtype == torch.float32 and x.dim() == 3 for x in args):
    return None
  b = k_s.shape[0]
  if (tuple(k_s.shape) != (b, 3, 3) or tuple(k_t.shape) != (b, 3, 3) or
      tuple(rot.shape) != (b, 3, 3) or tuple(t.shape) != (b, 3, 1)):
    return None
  from lsi import _C  # pylint: disable=g-import-not-at-top
  try:
    lib = _C.lib()
  except (OSError, RuntimeError):
    return None          # (library not built: the torch ops serve)
  k_s, k_t, rot, t = [x.detach().contiguous() for x in args]
  out = torch.empty((b, 4, 4), dtype=torch.float32)
  rc = lib.lsi_projection_matrices(b, k_s.data_ptr(), k_t.data_ptr(),
                                   rot.data_ptr(), t.data_ptr(), int(inverse),
                                   out.data_ptr())
  _C.check(rc, 'lsi_projection_matrices')
  return out


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
  fast = _host_matrices(k_s, k_t, rot, t, False)
  if fast is not None:
    return fast
  k_s_inv = _inv3(k_s)
  return _seq_matmul(pad_intrinsic(k_t),
                     _seq_matmul(pad_extrinsic(rot, t), pad_intrinsic(k_s_inv)))


def inverse_projection_matrix(k_s, k_t, rot, t):
  """trg pixel -> src frame (reference projection.py:89-106)."""
  fast = _host_matrices(k_s, k_t, rot, t, True)
  if fast is not None:
    return fast
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
