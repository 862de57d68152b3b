"""Layer composition and planar transforms (mirror of the reference's
lsi/geometry/layers.py)."""
import torch

from lsi.geometry import homography
from lsi.nnutils import helpers as nn_helpers


def _one_hot_argmax(selection_mask, depth):
  idx = torch.argmax(selection_mask, dim=0)
  return torch.moveaxis(
      torch.nn.functional.one_hot(idx, depth).to(selection_mask.dtype), -1, 0)


def compose(imgs, masks, dmaps, soft=False, min_disp=1e-6,
            depth_softmax_temp=1):
  """Composes layer images into one image with a white background layer at
  min_disp (reference layers.py:29-70).  imgs: L x [...] x C, masks/dmaps:
  L x [...] x 1.  Returns [...] x C.  On a ROCm device: one HIP pass
  (lsi_compose_fwd, forward only)."""
  if imgs.is_cuda:
    from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
    return _hip.compose(imgs, masks, dmaps, soft, min_disp, depth_softmax_temp)
  n_layers = imgs.shape[0]
  dmaps = torch.relu(dmaps)
  imgs = torch.cat([imgs, torch.ones_like(imgs[:1])], 0)
  masks = torch.cat([masks, torch.ones_like(masks[:1])], 0)
  dmaps = torch.cat([dmaps, torch.ones_like(dmaps[:1]) * min_disp], 0)
  selection_mask = nn_helpers.soft_z_buffering(
      masks, dmaps, depth_softmax_temp=depth_softmax_temp)
  if not soft:
    selection_mask = _one_hot_argmax(selection_mask, n_layers + 1)
  return torch.sum(selection_mask * imgs, dim=0)


def compose_depth(masks, dmaps, bg_layer=False, min_disp=1e-6,
                  depth_softmax_temp=1):
  """Composes layer disparities into one map (reference layers.py:73-115).  On
  a ROCm device: lsi_compose_depth_fwd."""
  if masks.is_cuda:
    from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
    return _hip.compose_depth(masks, dmaps, bg_layer, min_disp,
                              depth_softmax_temp)
  n_layers = masks.shape[0]
  dmaps = torch.relu(dmaps)
  bg_disp = torch.ones_like(dmaps[:1]) * min_disp
  masks = torch.cat([masks, torch.ones_like(masks[:1])], 0)
  dmaps = torch.cat([dmaps, bg_disp], 0)
  if bg_layer:
    dmaps_selection = torch.max(dmaps) - dmaps[0:n_layers]
    dmaps_selection = torch.cat([dmaps_selection, bg_disp], 0)
  else:
    dmaps_selection = dmaps
  selection_mask = nn_helpers.soft_z_buffering(
      masks, dmaps_selection, depth_softmax_temp=depth_softmax_temp)
  selection_mask = _one_hot_argmax(selection_mask, n_layers + 1)
  return torch.sum(selection_mask * dmaps, dim=0)


def planar_transform(imgs, masks, pixel_coords_trg, k_s, k_t, rot, t, n_hat, a):
  """Warps L planar layers (images + masks) into the target view and computes
  their target disparity maps (reference layers.py:118-162)."""
  n_layers = imgs.shape[0]

  def rep(x):
    return x.unsqueeze(0).expand((n_layers,) + tuple(x.shape))

  k_s, k_t, t, rot = rep(k_s), rep(k_t), rep(t), rep(rot)
  pixel_coords_trg = rep(pixel_coords_trg)
  imgs_masks = torch.cat([imgs, masks], dim=-1)
  imgs_masks_trg = homography.transform_plane_imgs(
      imgs_masks, pixel_coords_trg, k_s, k_t, rot, t, n_hat, a)
  imgs_trg, masks_trg = imgs_masks_trg[..., :3], imgs_masks_trg[..., 3:4]
  dmaps_trg = homography.trg_disp_maps(pixel_coords_trg, k_t, rot, t, n_hat, a)
  return imgs_trg, masks_trg, dmaps_trg
