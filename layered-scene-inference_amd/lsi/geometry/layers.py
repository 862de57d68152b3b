"""Layer composition and planar transforms (mirror of the reference's
lsi/geometry/layers.py)."""
import torch

from lsi.geometry import homography


def compose(imgs, masks, dmaps, soft=False, min_disp=1e-6,
            depth_softmax_temp=1):
  """Composes layer images into one image with a white background layer at
  min_disp (reference layers.py:29-70).  imgs: L x [...] x C, masks/dmaps:
  L x [...] x 1.  Returns [...] x C.  One HIP pass (lsi_compose_fwd, forward
  only: data generation and evaluation); CPU tensors raise."""
  from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
  return _hip.compose(imgs, masks, dmaps, soft, min_disp, depth_softmax_temp)


def compose_depth(masks, dmaps, bg_layer=False, min_disp=1e-6,
                  depth_softmax_temp=1):
  """Composes layer disparities into one map (reference layers.py:73-115):
  lsi_compose_depth_fwd (forward only); CPU tensors raise."""
  from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
  return _hip.compose_depth(masks, dmaps, bg_layer, min_disp,
                            depth_softmax_temp)


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
