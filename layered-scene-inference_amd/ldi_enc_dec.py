"""Script for training an LDI predictor via the view-synthesis loss: the eager,
MI355X counterpart of the reference's `ldi_enc_dec.py` (same flag names and
defaults, same dataset-dependent overrides, same six loss terms).

  python layered-scene-inference_amd/ldi_enc_dec.py --dataset=kitti \\
      --batch_size=4 --n_layers=2 --img_height=256 --img_width=768 ...
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
      layered-scene-inference_amd/ldi_enc_dec.py ...      # DDP over RCCL

`--dataset=kitti` reads stereo pairs through lsi/data/kitti (--kitti_data_root,
--kitti_dataset_variant, --data_split, as the reference).  The data set is not
available in the build container (no network): `--kitti_procedural=true` keeps
the KITTI camera model and the dataset-dependent constants
(ldi_enc_dec.py:415-427) and feeds procedural pairs instead (a smooth random
texture seen from the two cameras through one fronto-parallel plane, so that
the view-synthesis loss has something consistent to explain); the SUN / PASCAL
textures of the synthetic scenes are procedural as well.
"""
import argparse
import math
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
  sys.path.insert(0, _HERE)

from lsi.geometry import ldi as ldi_utils  # noqa: E402
from lsi.geometry import projection  # noqa: E402
from lsi.loss import loss  # noqa: E402
from lsi.nnutils import helpers as nn_helpers  # noqa: E402
from lsi.nnutils import nets  # noqa: E402
from lsi.nnutils import train_utils  # noqa: E402


def _bool(v):
  return str(v).lower() in ('1', 'true', 'yes')


def build_parser():
  p = argparse.ArgumentParser(description=__doc__)
  train_utils.define_default_flags(p)
  a = p.add_argument
  # experiment flags: names and defaults of reference ldi_enc_dec.py:39-123
  a('--exp_name', default='synth_ldi_pred_encdec')
  a('--n_layers', type=int, default=2)
  a('--pred_ldi_masks', type=_bool, default=False)
  a('--dataset', default='synthetic', choices=['synthetic', 'kitti'])
  a('--data_split', default='train', choices=['all', 'train', 'val', 'test'])
  # KITTI (reference ldi_enc_dec.py:76-83)
  a('--kitti_data_root', default='/datasets/kitti')
  a('--kitti_dataset_variant', default='mview',
    choices=['odom', 'mview', 'raw_city'])
  a('--kitti_dl_disparities', type=_bool, default=False)
  a('--kitti_procedural', type=_bool, default=False,
    help='no KITTI files: procedural stereo pairs seen through the KITTI camera '
    'model and constants (throughput runs, tests).  Without this switch a '
    'missing --kitti_data_root is an error')
  a('--self_cons_wt', type=float, default=1.0)
  a('--l0_self_cons', type=_bool, default=False)
  a('--indep_splat_wt', type=float, default=1.0)
  a('--compose_splat_wt', type=float, default=1.0)
  a('--splat_bdry_ignore', type=float, default=0.1)
  a('--zbuf_scale', type=float, default=50)
  a('--trg_splat_downsampling', type=float, default=0.5)
  a('--disp_smoothness_wt', type=float, default=0.1)
  a('--incr_depth_wt', type=float, default=10.0)
  a('--use_unet', type=_bool, default=True)
  a('--n_layerwise_steps', type=int, default=3)
  a('--bg_layer_disp', type=float, default=1e-6)
  a('--depth_softmax_temp', type=float, default=1e-6)
  a('--max_disp', type=float, default=0)
  # build-specific switches
  # synthetic planar worlds (reference ldi_enc_dec.py:52-55, 86-123): procedural
  # textures replace the SUN / PASCAL images
  a('--synth_scene', default='pairs', choices=['pairs', 'planes'],
    help="synthetic inputs: 'pairs' = shifted smooth images generated on the "
    "device (throughput runs); 'planes' = box room + billboard objects "
    'rendered through planar_transform + compose (lsi/data/synthetic_planes)')
  a('--debug_synth_texture', type=_bool, default=False,
    help='feed the ground-truth fg / bg disparities in place of the predicted '
    'ones (n_layers = 2, --synth_scene planes): the renderer must then '
    'reconstruct the other view')
  a('--synth_ds_factor', type=int, default=1)
  a('--n_obj_min', type=int, default=1)
  a('--n_obj_max', type=int, default=4)
  a('--n_box_planes', type=int, default=5)
  a('--bf16', type=_bool, default=True,
    help='the network in bf16 (torch.autocast) with fp32 accumulation -- the '
    'product default on a ROCm device: every convolution then runs on this '
    "repo's MFMA kernels (csrc/lsi_conv*.hip), every batch norm on csrc/lsi_bn.hip "
    '(BASELINE config 4: "bf16 convs + fp32 splat"); the renderer and the losses '
    'are fp32 either way.  false = the reference\'s own arithmetic (fp32 '
    'convolutions): through the library (MIOpen), DESIGN.md 4.8')
  a('--batched_pairs', type=_bool, default=True,
    help='source and target images go through the network in one pass, every '
    'batch norm with separate statistics per view (same arithmetic as two passes)')
  a('--paired_splat', type=_bool, default=True,
    help='with --batched_pairs: the src -> trg and trg -> src renderings of a step '
    'are one forward_splat_both call on the 2 B LDIs the network pass produced')
  a('--fused_adam', type=_bool, default=True,
    help='torch.optim.Adam(fused=True) on the GPU')
  a('--miopen_tune', type=_bool, default=False,
    help='let MIOpen tune its convolution solvers for this run (one-time 50-100 s; '
    'MIOPEN_FIND_ENFORCE=3, cached in ~/.config/miopen)')
  a('--flat_grads', type=_bool, default=False,
    help='data parallel without the DDP wrapper: one flat gradient buffer, one '
    'all-reduce per step (implied by --hip_graph with more than one rank)')
  a('--channels_last', type=_bool, default=True)
  a('--cpu', type=_bool, default=False,
    help='keep the model on the CPU (plumbing tests only: the renderer and the '
    'losses are HIP kernels, compute_losses raises without a ROCm device)')
  a('--miopen_search', type=_bool, default=False,
    help='exhaustive MIOpen solver search (torch.backends.cudnn.benchmark)')
  a('--tf_checkpoint_complete', type=_bool, default=False,
    help='also hold the variables the reference creates and checkpoints but '
    'never trains (the fc stack on the U-Net bottleneck, upcnv3 .. icnv1), '
    'frozen: the exact variable list of a TF checkpoint (lsi/nnutils/'
    'tf_checkpoint.py)')
  a('--hip_graph', type=_bool, default=False,
    help='capture the training step in a HIP graph (single process)')
  return p


def apply_dataset_overrides(opts):
  """ldi_enc_dec.py:415-427."""
  opts.checkpoint_dir = os.path.join(opts.checkpoint_dir, opts.exp_name)
  if opts.dataset == 'synthetic':
    opts.bg_layer_disp = 2e-1
    opts.depth_softmax_temp = 0.4
    if opts.max_disp == 0:
      opts.max_disp = 1.0
  elif opts.dataset == 'kitti':
    opts.bg_layer_disp = 1e-3
    opts.depth_softmax_temp = 0.4
    if opts.max_disp == 0:
      opts.max_disp = 0.4
  return opts


class LdiNet(torch.nn.Module):
  """encoder-decoder + per-layer LDI heads (define_pred_graph,
  ldi_enc_dec.py:175-228); the same weights process src and trg images."""

  def __init__(self, opts):
    super().__init__()
    if opts.use_unet:
      complete = bool(getattr(opts, 'tf_checkpoint_complete', False))
      self.enc_dec = nets.encoder_decoder_unet(
          nl_diff_enc_dec=opts.n_layerwise_steps, with_fc=complete,
          with_dead_decoder=complete,
          in_hw=(opts.img_height, opts.img_width))
    else:
      self.enc_dec = nets.encoder_decoder_simple(
          nl_diff_enc_dec=opts.n_layerwise_steps,
          in_hw=(opts.img_height, opts.img_width))
    self.ldi_tex_disp = nets.ldi_predictor(
        self.enc_dec.out_channels, n_layers=opts.n_layers,
        n_layerwise_steps=opts.n_layerwise_steps,
        skip_channels=self.enc_dec.skip_channels,
        pred_masks=opts.pred_ldi_masks)
    self.max_disp = opts.max_disp
    # (the `fc` stack of the complete TF parameter set sees the batch as one)
    self.batched_pairs = bool(getattr(opts, 'batched_pairs', True))

  def predict(self, imgs):
    _, feat_dec, skip_feat, _ = self.enc_dec(imgs)
    # float32 LDI, disparities scaled by max_disp (ldi_enc_dec.py:203-205)
    return self.ldi_tex_disp(feat_dec, skip_feat, disp_scale=self.max_disp)

  def forward(self, imgs_src, imgs_trg):
    self.pair_ldi = None
    if not self.batched_pairs or imgs_src.shape != imgs_trg.shape:
      return self.predict(imgs_src), self.predict(imgs_trg)
    # One pass over [src; trg]: every batch norm keeps separate statistics for
    # the two halves (nets.bn_groups), so the result is the reference's two
    # passes (ldi_enc_dec.py:175-228) with half the kernel launches and the
    # convolutions at twice the batch.  The halves come back as views.
    b = imgs_src.shape[0]
    with nets.bn_groups(2):
      tex, masks, disps = self.predict(torch.cat([imgs_src, imgs_trg], dim=0))
    half = lambda t, k: None if t is None else t[:, k * b:(k + 1) * b]
    # (the un-halved LDI: the trainer renders both directions with one launch)
    self.pair_ldi = [tex, masks, disps]
    return ([half(tex, 0), half(masks, 0), half(disps, 0)],
            [half(tex, 1), half(masks, 1), half(disps, 1)])


class SyntheticPairs(object):
  """Procedural stereo-like pairs with the dataset's camera model."""

  def __init__(self, opts, device, seed):
    self.opts, self.device = opts, device
    # images are generated where they are consumed (the CPU version of this
    # generator cost 12-20 ms per batch, a third of a training step)
    self.gen = torch.Generator(device=device).manual_seed(seed)

  def forward(self, bs):
    o = self.opts
    h, w = o.img_height, o.img_width
    lo = torch.rand((bs, 3, h // 16 + 2, w // 16 + 2), generator=self.gen,
                    device=self.device)
    img = torch.nn.functional.interpolate(lo, size=(h, w + 64), mode='bicubic',
                                          align_corners=False).clamp(0, 1)
    shift = 16  # pixels of parallax of the (single, fronto-parallel) plane
    src = img[..., 32:32 + w].permute(0, 2, 3, 1).contiguous()
    trg = img[..., 32 - shift:32 - shift + w].permute(0, 2, 3, 1).contiguous()
    if o.dataset == 'kitti':
      k = torch.tensor([[0.58 * w, 0, w / 2.0], [0, 0.58 * w, h / 2.0],
                        [0, 0, 1.0]])
      t = torch.tensor([[-0.532], [0.0], [0.0]])
    else:
      k = torch.tensor([[float(w), 0, w / 2.0], [0, float(h), h / 2.0],
                        [0, 0, 1.0]])
      t = torch.tensor([[-0.3], [0.0], [0.0]])
    k = k.expand(bs, 3, 3).contiguous()
    rot = torch.eye(3).expand(bs, 3, 3).contiguous()
    t = t.expand(bs, 3, 1).contiguous()
    return (src, trg, k, k.clone(), rot, t)


class KittiBatches(object):
  """lsi.data.kitti.DataLoader batches (NumPy) as float32 torch tensors:
  (img_s, img_t, k_s, k_t, rot, trans[, disp_s, disp_t])."""

  def __init__(self, loader, rank=0):
    self.loader = loader
    # every rank walks the data in its own order
    loader._rng = __import__('numpy').random.RandomState(rank)

  @property
  def src_image_names(self):
    return self.loader.src_image_names

  def forward(self, bs):
    return tuple(torch.as_tensor(a, dtype=torch.float32)
                 for a in self.loader.forward(bs))


class Trainer(train_utils.Trainer):
  """LDI prediction trainer (reference ldi_enc_dec.py:126-410)."""

  def define_data_loader(self):
    opts = self.opts
    if opts.dataset == 'synthetic' and (opts.synth_scene == 'planes' or
                                        opts.debug_synth_texture):
      from lsi.data import synthetic_planes  # pylint: disable=g-import-not-at-top
      if opts.debug_synth_texture and opts.n_layers != 2:
        raise ValueError('debug_synth_texture feeds (fg, bg) disparities: '
                         'n_layers must be 2')
      opts.synth_dl_eval_data = (bool(opts.debug_synth_texture) or
                                 bool(getattr(opts, 'synth_dl_eval_data', False)))
      self.data_loader = synthetic_planes.DataLoader(
          opts, device=self.device, seed=1234 + self.rank)
    elif opts.dataset == 'kitti' and not opts.kitti_procedural:
      # reference ldi_enc_dec.py:134-137
      from lsi.data.kitti import data as kitti_data  # pylint: disable=g-import-not-at-top
      if not os.path.isdir(opts.kitti_data_root):
        raise FileNotFoundError(
            '--dataset=kitti: no directory %r (--kitti_data_root); pass '
            '--kitti_procedural=true for procedural pairs with KITTI cameras'
            % opts.kitti_data_root)
      self.data_loader = KittiBatches(kitti_data.DataLoader(opts), self.rank)
    else:
      self.data_loader = SyntheticPairs(opts, self.device, 1234 + self.rank)
    bs = self.opts.batch_size
    self.pixel_coords = nn_helpers.pixel_coords(bs, self.opts.img_height,
                                                self.opts.img_width)

  def build_model(self):
    return LdiNet(self.opts)

  def feed(self):
    """One batch: (img_src, img_trg, k_s, k_t, rot, trans); with
    debug_synth_texture also the ground-truth LDI disparities (reference
    ldi_enc_dec.py:230-263)."""
    batch = self.data_loader.forward(self.opts.batch_size)
    self.gt_disps = None
    if self.opts.debug_synth_texture:
      (_, _, _, _, _, _, _, _, d_s_fg, d_s_bg, d_t_fg, d_t_bg, _, _) = batch
      self.gt_disps = (torch.stack([d_s_fg, d_s_bg], 0),
                       torch.stack([d_t_fg, d_t_bg], 0))
    return tuple(batch[:6])

  def stage(self, batch):
    """Images to the device; the two src->trg projection matrices are built on
    the host (tiny), kept there for the kernel-family choice and copied to the
    device.  plan = what the renderer would choose for these cameras."""
    imgs_src, imgs_trg, k_s, k_t, rot_mat, trans_mat = batch
    f32 = lambda t: t.detach().to('cpu', torch.float32)
    k_s, k_t, rot_mat, trans_mat = f32(k_s), f32(k_t), f32(rot_mat), f32(trans_mat)
    inv_rot_mat = nn_helpers.transpose(rot_mat)
    inv_trans_mat = -torch.matmul(inv_rot_mat, trans_mat)
    mat_trg = projection.forward_projection_matrix(k_s, k_t, rot_mat, trans_mat)
    mat_src = projection.forward_projection_matrix(k_t, k_s, inv_rot_mat,
                                                   inv_trans_mat)
    self.host_mats = {'trg': mat_trg.contiguous(), 'src': mat_src.contiguous()}
    plan = None
    if self.device.type == 'cuda':
      o = self.opts
      plan = tuple(
          ldi_utils.plan_key((o.n_layers, imgs_src.shape[0], o.img_height,
                              o.img_width), o.trg_splat_downsampling,
                             o.max_disp, self.host_mats[w])
          for w in ('trg', 'src'))
    dev = self.device

    def up(t):
      # A pageable host-to-device copy blocks the host until the GPU has drained
      # the kernels queued before it -- once per step that serialises the step's
      # launches with the previous step's kernels (eager training: 12 ms per step
      # instead of 10).  Through pinned memory the copy is asynchronous.
      if dev.type == 'cuda' and not t.is_cuda:
        return t.pin_memory().to(dev, non_blocking=True)
      return t.to(dev)
    staged = [up(imgs_src), up(imgs_trg), up(mat_trg), up(mat_src)]
    if getattr(self, 'gt_disps', None) is not None:
      # (part of the staged batch: a captured HIP graph replays on static copies)
      staged += [self.gt_disps[0].to(dev), self.gt_disps[1].to(dev)]
    return staged, plan

  def compute_losses(self, staged):
    opts = self.opts
    if self.device.type != 'cuda':
      raise RuntimeError('the renderer and the six loss terms are HIP kernels: '
                         'training needs a ROCm device')
    imgs_src, imgs_trg, mat_trg, mat_src = staged[:4]
    mats = {'trg': mat_trg, 'src': mat_src}
    amp = (torch.autocast('cuda', dtype=torch.bfloat16) if opts.bf16
           else _NullCtx())
    if hasattr(self, 'model'):
      self.model.pair_ldi = None
    with amp:
      ldi_src, ldi_trg = self.train_model(imgs_src, imgs_trg)
    if opts.debug_synth_texture and len(staged) >= 6:
      # ldi_enc_dec.py:223-225: keep the graph, substitute the values
      ldi_src[2] = 0 * ldi_src[2] + staged[4]
      ldi_trg[2] = 0 * ldi_trg[2] + staged[5]

    def ones_like_mask(l):
      return l[1] if l[1] is not None else torch.ones_like(l[2])

    # The network's output is one buffer of 2 B LDIs (source views, then target
    # views) whose halves `ldi_src` / `ldi_trg` are views of it.  A loss taken on
    # a half sends its gradient back through a slice: autograd fills a zero
    # tensor of the whole buffer per half and adds them up (texture, masks,
    # disparities: a dozen kernels over 25 - 100 MB each per step).  Every term
    # below is a mean over the batch, so the term of the pair is twice the mean
    # over the 2 B LDIs: one call per term on the whole buffer, no slices.
    pair = getattr(getattr(self, 'model', None), 'pair_ldi', None)
    paired = (pair is not None and getattr(opts, 'paired_splat', True) and
              not (opts.debug_synth_texture and len(staged) >= 6))

    # self-consistency (ldi_enc_dec.py:269-294)
    if paired and not opts.l0_self_cons:
      self_cons_loss = 2.0 * loss.zbuffer_composition_loss(
          pair[0], ones_like_mask(pair), pair[2],
          torch.cat([imgs_src, imgs_trg], dim=0),
          zbuf_scale=opts.zbuf_scale, bg_layer_disp=opts.bg_layer_disp,
          max_disp=opts.max_disp)
    elif opts.l0_self_cons:
      sc_src = torch.mean(torch.abs(imgs_src - ldi_src[0][0]))
      sc_trg = torch.mean(torch.abs(imgs_trg - ldi_trg[0][0]))
    else:
      sc_src = loss.zbuffer_composition_loss(
          ldi_src[0], ones_like_mask(ldi_src), ldi_src[2], imgs_src,
          zbuf_scale=opts.zbuf_scale, bg_layer_disp=opts.bg_layer_disp,
          max_disp=opts.max_disp)
      sc_trg = loss.zbuffer_composition_loss(
          ldi_trg[0], ones_like_mask(ldi_trg), ldi_trg[2], imgs_trg,
          zbuf_scale=opts.zbuf_scale, bg_layer_disp=opts.bg_layer_disp,
          max_disp=opts.max_disp)
    if not (paired and not opts.l0_self_cons):
      self_cons_loss = sc_src + sc_trg

    # view synthesis via forward splatting (ldi_enc_dec.py:296-357)
    zero = imgs_src.new_zeros(())
    indep_splat_loss, compose_splat_loss = zero, zero
    # One sweep per direction renders the per-layer AND the composed view
    # (the reference makes four forward_splat calls whose per-layer splats
    # are identical pairwise).
    if paired and (opts.indep_splat_wt > 0 or opts.compose_splat_wt > 0):
      # Both directions of the pair from ONE launch (and one in the backward):
      # the network's output is one buffer of 2 B LDIs -- the B source views'
      # then the B target views' -- so the src -> trg and the trg -> src
      # renderings (reference ldi_enc_dec.py:302-334) are a batch of 2 B
      # elements with their own matrices, compared with [imgs_trg; imgs_src].
      # A rank with 4 samples then issues 8-view launches instead of twice 4.
      # Each loss term is a mean over the batch: the mean over 2 B samples is
      # half the sum of the two directions' means.
      mat2 = torch.cat([mats['trg'], mats['src']], dim=0)
      host2 = torch.cat([self.host_mats['trg'], self.host_mats['src']], dim=0)
      target2 = torch.cat([imgs_trg, imgs_src], dim=0)
      img_i, _, img_c, _ = ldi_utils.forward_splat_both(
          pair, mat2, trg_downsampling=opts.trg_splat_downsampling,
          zbuf_scale=opts.zbuf_scale, bg_layer_disp=opts.bg_layer_disp,
          max_disp=opts.max_disp, mat_host=host2)
      if opts.indep_splat_wt > 0:
        indep_splat_loss = 2.0 * loss.view_synthesis_loss(
            img_i, target2, opts.splat_bdry_ignore)
      if opts.compose_splat_wt > 0:
        compose_splat_loss = 2.0 * loss.view_synthesis_loss(
            img_c, target2, opts.splat_bdry_ignore)
    elif opts.indep_splat_wt > 0 or opts.compose_splat_wt > 0:
      for which in ('trg', 'src'):
        if which == 'trg':
          target, l = imgs_trg, ldi_src
        else:
          target, l = imgs_src, ldi_trg
        img_i, _, img_c, _ = ldi_utils.forward_splat_both(
            l, mats[which], trg_downsampling=opts.trg_splat_downsampling,
            zbuf_scale=opts.zbuf_scale, bg_layer_disp=opts.bg_layer_disp,
            max_disp=opts.max_disp, mat_host=self.host_mats[which])
        if opts.indep_splat_wt > 0:
          indep_splat_loss = indep_splat_loss + loss.view_synthesis_loss(
              img_i, target, opts.splat_bdry_ignore)
        if opts.compose_splat_wt > 0:
          compose_splat_loss = compose_splat_loss + loss.view_synthesis_loss(
              img_c, target, opts.splat_bdry_ignore)

    # regularisers (ldi_enc_dec.py:388-396): both from one read of each
    # disparity tensor (fused HIP kernel lsi_disp_reg_loss_fwd)
    from lsi.loss import _hip as loss_hip  # pylint: disable=g-import-not-at-top
    if paired:
      sm_p, dc_p = loss_hip.disp_regularisers(pair[2])
      disp_smoothness_loss = 2.0 * sm_p
      incr_depth_loss = 2.0 * dc_p if opts.n_layers > 1 else zero
    else:
      sm_s, dc_s = loss_hip.disp_regularisers(ldi_src[2])
      sm_t, dc_t = loss_hip.disp_regularisers(ldi_trg[2])
      disp_smoothness_loss = sm_s + sm_t
      incr_depth_loss = (dc_s + dc_t) if opts.n_layers > 1 else zero

    total = zero
    if opts.self_cons_wt > 0:
      total = total + opts.self_cons_wt * self_cons_loss
    if opts.compose_splat_wt > 0:
      total = total + opts.compose_splat_wt * compose_splat_loss
    if opts.indep_splat_wt > 0:
      total = total + opts.indep_splat_wt * indep_splat_loss
    if opts.incr_depth_wt > 0:
      total = total + (opts.incr_depth_wt / opts.max_disp) * incr_depth_loss
    if opts.disp_smoothness_wt > 0:
      total = total + (opts.disp_smoothness_wt /
                       (opts.max_disp * opts.max_disp)) * disp_smoothness_loss
    scalars = {
        'self_cons_loss': self_cons_loss,
        'compose_splat_loss': compose_splat_loss,
        'indep_splat_loss': indep_splat_loss,
        'incr_depth_loss': incr_depth_loss,
        'disp_smoothness_loss': disp_smoothness_loss,
        'total_loss': total,
    }
    return total, scalars


class _NullCtx(object):

  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False


def main(argv=None):
  opts = apply_dataset_overrides(build_parser().parse_args(argv))
  trainer = Trainer(opts)
  trainer.setup()
  trainer.train()
  if trainer.dist is not None:
    trainer.dist.destroy_process_group()
  return trainer


if __name__ == '__main__':
  main()
