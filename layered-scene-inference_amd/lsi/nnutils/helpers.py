"""Misc helper functions (mirror of the reference's lsi/nnutils/helpers.py).

Small host-side / elementwise helpers on torch.Tensors; the renderer itself is
in lsi.geometry.ldi / lsi.geometry.sampling (HIP kernels).
"""
import torch


def transpose(rot):
  """Transposes the last two dimensions (reference helpers.py:65-79)."""
  return rot.transpose(-1, -2)


def divide_safe(num, den, name=None):
  """num / (den + 1e-8 * [den == 0]) (reference helpers.py:82-85)."""
  del name
  eps = 1e-8
  den = torch.as_tensor(den, dtype=torch.float32)
  den = den + eps * (den == 0).to(torch.float32)
  return num / den


def pixel_coords(bs, h, w, device=None):
  """bs x h x w x 3 tensor with (x, y, 1) at each pixel centre, 0.5-indexed
  (reference helpers.py:88-113)."""
  ys = torch.arange(1, h + 1, dtype=torch.float32, device=device) - 0.5
  xs = torch.arange(1, w + 1, dtype=torch.float32, device=device) - 0.5
  grid = torch.stack([xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w),
                      torch.ones((h, w), dtype=torch.float32, device=device)],
                     dim=-1)
  return grid.unsqueeze(0).expand(bs, h, w, 3).contiguous()


def seq_matmul(a, b):
  """Matrix product accumulated sequentially over k, every multiply and add a
  separately rounded fp32 elementwise op (no FMA, no blocked BLAS order): the
  evaluation order the parity oracle pins for TF's small matmuls
  (oracle/lsi_oracle.py: matmul_seq).  Coordinates, masks and thresholds
  derived from it are bit-identical to the oracle's."""
  out = a[..., :, 0:1] * b[..., 0:1, :]
  for k in range(1, a.shape[-1]):
    out = out + a[..., :, k:k + 1] * b[..., k:k + 1, :]
  return out


def transform_pts(pts_coords_init, tform_mat):
  """Per-pixel D x D matrix transform, [...] x H x W x D (reference
  helpers.py:116-137).  The product is the sequential-k one (seq_matmul):
  the pixel coordinates it yields feed floor / threshold decisions."""
  shape = pts_coords_init.shape
  d = tform_mat.shape[-1]
  lead = tuple(tform_mat.shape[:-2])
  flat = pts_coords_init.reshape(lead + (-1, d))
  out = seq_matmul(flat, tform_mat.transpose(-1, -2))
  return out.reshape(shape)


def soft_z_buffering(layer_masks, layer_disps, depth_softmax_temp=1):
  """Pixelwise probability of belonging to each layer, L x [...] x 1
  (reference helpers.py:140-160)."""
  eps = 1e-8
  layer_disps = torch.relu(layer_disps)
  layer_depths = divide_safe(1.0, layer_disps)
  log_depth_probs = -layer_depths / depth_softmax_temp
  log_layer_probs = torch.log(layer_masks + eps) + log_depth_probs
  log_layer_probs = log_layer_probs - log_layer_probs.max(dim=0, keepdim=True)[0]
  layer_probs = torch.exp(log_layer_probs)
  return layer_probs / layer_probs.sum(dim=0, keepdim=True)


def enforce_bg_occupied(ldi_masks):
  """Last layer's mask set to all ones (reference helpers.py:163-177)."""
  n_layers = ldi_masks.shape[0]
  if n_layers == 1:
    return ldi_masks * 0 + 1
  masks_fg, masks_bg = ldi_masks[:n_layers - 1], ldi_masks[n_layers - 1:]
  return torch.cat([masks_fg, masks_bg * 0 + 1], dim=0)


def zbuffer_weights(disps, scale=50):
  """exp((clip(d, 0, 1) - 0.5) * scale) * [d > 0] (reference helpers.py:180-193)."""
  disps = torch.as_tensor(disps, dtype=torch.float32)
  pos_disps = (disps > 0).to(torch.float32)
  disps = torch.clamp(disps, 0, 1) - 0.5
  return torch.exp(disps * scale) * pos_disps
