"""View-synthesis evaluation metrics (the arithmetic of the reference's
`ldi_pred_eval.py:297-548` `define_metrics`, without the TF session / HTML
plumbing): masked L1 view-synthesis error, its dis-occlusion-restricted
variant, rendered-disparity error, and PSNR.  Metrics are returned as
(sum, normaliser) pairs; datasets are aggregated as sum(metric)/sum(norm) like
`test_utils.py:247-255`.
"""
import torch

from lsi.geometry import ldi as ldi_utils
from lsi.loss import loss as loss_utils


def _centre_mask(b, h, w, frac, device):
  x_min = loss_utils._py2_round(w * frac)
  y_min = loss_utils._py2_round(h * frac)
  m = torch.zeros((b, h, w), device=device)
  m[:, y_min:h - y_min, x_min:w - x_min] = 1.0
  return m


def view_synthesis_metrics(ldi_src, pixel_coords, k_s, k_t, rot, t, imgs_trg,
                           opts, valid_mask=None, disocc_mask=None,
                           gt_disp_trg=None):
  """Renders `ldi_src` into the target camera (compose_layers=True,
  compute_trg_disp=True: ldi_pred_eval.py:340-353) and accumulates the masked
  errors against `imgs_trg` (B x H x W x 3).

  opts needs: trg_splat_downsampling, zbuf_scale, bg_layer_disp, max_disp,
  splat_bdry_ignore.  valid_mask / disocc_mask / gt_disp_trg: B x H x W x 1 or
  None.  Returns a dict name -> (sum, norm) of 0-d tensors.
  """
  recons, _, recons_disp = ldi_utils.forward_splat(
      ldi_src, pixel_coords, k_s, k_t, rot, t, compose_layers=True,
      compute_trg_disp=True, trg_downsampling=opts.trg_splat_downsampling,
      zbuf_scale=opts.zbuf_scale, bg_layer_disp=opts.bg_layer_disp,
      max_disp=opts.max_disp)
  _, b, ht, wt, _ = recons.shape
  dev = recons.device
  target = loss_utils.area_downsample(imgs_trg, ht, wt)
  if valid_mask is None:
    valid = torch.ones((b, ht, wt), device=dev)
  else:  # "ignore pixels that might have aliasing": > 0.95 after AREA resize
    valid = (loss_utils.area_downsample(valid_mask, ht, wt)[..., 0] >
             0.95).float()
  centre = _centre_mask(b, ht, wt, opts.splat_bdry_ignore, dev) * valid
  pw = torch.min(torch.mean(torch.abs(target.unsqueeze(0) - recons), dim=4),
                 dim=0)[0] * centre
  out = {'compose_splat_loss': (pw.sum(), centre.sum())}
  mse = (((target - recons[0])**2).mean(dim=3) * centre).sum() / centre.sum()
  out['psnr'] = (10.0 * torch.log10(1.0 / mse.clamp_min(1e-20)), torch.ones(()))
  if disocc_mask is not None:
    # the reference uses the un-thresholded down-sampled mask here
    # (ldi_pred_eval.py:403-409)
    dm = loss_utils.area_downsample(disocc_mask.float(), ht, wt)[..., 0]
    out['compose_splat_loss_disocc'] = ((pw * dm).sum(), (centre * dm).sum())
  if gt_disp_trg is not None:
    gt = loss_utils.area_downsample(gt_disp_trg, ht, wt)
    pd = torch.min(torch.mean(torch.abs(gt.unsqueeze(0) - recons_disp), dim=4),
                   dim=0)[0] * centre
    out['depth_splat_loss'] = (pd.sum(), centre.sum())
    if disocc_mask is not None:
      out['depth_splat_loss_disocc'] = ((pd * dm).sum(), (centre * dm).sum())
  return out


def aggregate(metric_dicts):
  """sum(metric) / sum(norm) over iterations (test_utils.py:247-255)."""
  totals = {}
  for d in metric_dicts:
    for k, (s, n) in d.items():
      a, c = totals.get(k, (0.0, 0.0))
      totals[k] = (a + float(s), c + float(n))
  return {k: a / c for k, (a, c) in totals.items() if c > 0}


def layer_prediction_metrics(ldi_src, ldi_trg, imgs_src, imgs_trg, gt, opts):
  """Per-layer texture / disparity errors of the predicted LDIs against ground
  truth (reference ldi_pred_eval.py:455-521; synthetic data only).

  ldi_*: [tex L x B x H x W x 3, masks, disps L x B x H x W x 1].
  gt: dict with src_gt_disp, trg_gt_disp (foreground, B x H x W x 1) and,
      for the background metrics, src/trg_gt_disp_bg and src/trg_gt_tex_bg.
  Foreground (layer 0): valid where the gt disparity exceeds bg_layer_disp;
  background (last layer): valid where the foreground hides the background
  (gt_disp > gt_disp_bg).  Texture errors are summed over the pixels and divided
  by 3 (the channels).  Returns name -> (sum, norm)."""
  n_layers = ldi_src[0].shape[0]
  out = {}
  v_s = (gt['src_gt_disp'] > opts.bg_layer_disp).float()
  v_t = (gt['trg_gt_disp'] > opts.bg_layer_disp).float()
  fg_tex = ((torch.abs(ldi_src[0][0] - imgs_src) * v_s).sum() / 3 +
            (torch.abs(ldi_trg[0][0] - imgs_trg) * v_t).sum() / 3)
  fg_disp = ((torch.abs(ldi_src[2][0] - gt['src_gt_disp']) * v_s).sum() +
             (torch.abs(ldi_trg[2][0] - gt['trg_gt_disp']) * v_t).sum())
  n_fg = (v_s + v_t).sum()
  out['fg_tex_error'] = (fg_tex, n_fg)
  out['fg_disp_error'] = (fg_disp, n_fg)
  if 'src_gt_disp_bg' in gt:
    b_s = (gt['src_gt_disp'] > gt['src_gt_disp_bg']).float()
    b_t = (gt['trg_gt_disp'] > gt['trg_gt_disp_bg']).float()
    bg_tex = ((torch.abs(ldi_src[0][n_layers - 1] - gt['src_gt_tex_bg']) *
               b_s).sum() / 3 +
              (torch.abs(ldi_trg[0][n_layers - 1] - gt['trg_gt_tex_bg']) *
               b_t).sum() / 3)
    bg_disp = ((torch.abs(ldi_src[2][n_layers - 1] - gt['src_gt_disp_bg']) *
                b_s).sum() +
               (torch.abs(ldi_trg[2][n_layers - 1] - gt['trg_gt_disp_bg']) *
                b_t).sum())
    n_bg = (b_s + b_t).sum()
    out['bg_tex_error'] = (bg_tex, n_bg)
    out['bg_disp_error'] = (bg_disp, n_bg)
  return out
