"""Loss functions on LDIs (mirror of the reference's lsi/loss/loss.py) plus the
view-synthesis loss the reference computes inline in ldi_enc_dec.py:337-357.

The three losses are fused HIP kernels (csrc/lsi_loss.hip, forward and
backward); tensors that do not live on a ROCm device raise, as for the
renderer -- there is no second implementation (the torch restatements the
tests check gradients against live in oracle/lsi_torch_ref.py)."""
import math

import torch



def event_prob(layer_masks):
  """Per-pixel ordered-multiplication layer probabilities (reference
  loss.py:27-45; dead code there, kept for surface parity)."""
  eps = 1e-6
  layer_masks = torch.clamp(layer_masks, eps, 1 - eps)
  log_inv_m = torch.log(1 - layer_masks)
  log_prob = torch.cumsum(log_inv_m, dim=0) - log_inv_m + torch.log(layer_masks)
  layer_probs = torch.exp(log_prob)
  escape_probs = 1 - torch.sum(layer_probs, dim=0, keepdim=True)
  return layer_probs, escape_probs


def decreasing_disp_loss(layer_disps):
  """Penalises disparities that increase from one layer to the next, with the
  nearer layer detached (reference loss.py:48-63).  On a ROCm device: the fused
  HIP kernel (lsi_disp_reg_loss_fwd)."""
  n_layers = layer_disps.shape[0]
  if n_layers == 1:
    return 0
  from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
  return _hip.disp_regularisers(layer_disps)[1]


def zbuffer_composition_loss(layer_imgs, layer_masks, layer_disps, trg_imgs,
                             bg_layer_disp=0, max_disp=1, zbuf_scale=10):
  """Depth+mask weighted self-consistency loss with a white background layer
  (reference loss.py:66-115).  On a ROCm device: one fused HIP pass
  (lsi_zbuf_comp_loss_fwd / _bwd)."""
  from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
  return _hip.zbuffer_composition_loss(layer_imgs, layer_masks, layer_disps,
                                       trg_imgs, bg_layer_disp, max_disp,
                                       zbuf_scale)


def area_downsample(img, ht, wt):
  """tf.image.resize_images(..., AREA) for integer reduction factors: exact box
  mean.  img: B x H x W x C."""
  b, h, w, c = img.shape
  fy, fx = h // ht, w // wt
  if fy * ht != h or fx * wt != w:
    raise ValueError('AREA resize implemented for integer factors only')
  if fy == 1 and fx == 1:
    return img
  return img.reshape(b, ht, fy, wt, fx, c).mean(dim=(2, 4))


def _py2_round(x):
  return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def view_synthesis_loss(recons_splat, to_recons_img, splat_bdry_ignore=0.05):
  """L1 view-synthesis loss of the reference's training script
  (ldi_enc_dec.py:337-357): AREA-downsample the target to the splat's size,
  mean |diff| over channels, min over layers, crop the border, mean."""
  _, _, ht, wt, _ = recons_splat.shape
  from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
  return _hip.view_synthesis_loss(recons_splat, to_recons_img,
                                  _py2_round(wt * splat_bdry_ignore),
                                  _py2_round(ht * splat_bdry_ignore))
