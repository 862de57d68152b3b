"""CNN mirror (lsi.nnutils.nets), eager trainer and DDP wiring on CPU.

The splat losses need the GPU (no CPU fallback), so the CPU runs switch them off
(--indep_splat_wt 0 --compose_splat_wt 0) and exercise everything else: the
U-Net + LDI heads, the self-consistency / smoothness / ordering losses, Adam,
checkpoint cadence + resume, and gradient all-reduce over two gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG

sys.path.insert(0, PKG)
from lsi.nnutils import nets  # noqa: E402


def test_parameter_counts_match_the_reference_graph():
  # live parameters 37.65 M (L=2) and 39.07 M (L=4)
  for nl, want in ((2, 37645736), (4, 39072304)):
    unet = nets.encoder_decoder_unet(nl_diff_enc_dec=3)
    head = nets.ldi_predictor(unet.out_channels, n_layers=nl,
                              n_layerwise_steps=3,
                              skip_channels=unet.skip_channels)
    assert nets.count_parameters(unet) + nets.count_parameters(head) == want
  assert unet.out_channels == 128          # icnv4 (nets.py:348 with nl_diff=3)


def test_tf_same_padding_and_slim_batch_norm():
  torch.manual_seed(0)
  # stride-2 7x7: TF SAME pads 2 before / 3 after on an even input
  conv = nets.SlimConv2d(1, 1, 7, 2, batch_norm=False, activation=None)
  x = torch.zeros(1, 1, 8, 8)
  x[0, 0, 0, 0] = 1.0
  y = conv(x)
  assert y.shape == (1, 1, 4, 4)
  w = conv.conv.weight[0, 0]
  # output (0,0) sees input (0,0) through kernel tap (2,2); output (1,1)
  # sees it through tap (0,0)
  assert torch.allclose(y[0, 0, 0, 0], w[2, 2] + conv.conv.bias[0])
  assert torch.allclose(y[0, 0, 1, 1], w[0, 0] + conv.conv.bias[0])
  assert nets._same_pad(8, 5, 2) == (1, 2) and nets._same_pad(8, 3, 2) == (0, 1)
  assert nets._same_pad(8, 3, 1) == (1, 1) and nets._same_pad(7, 3, 2) == (1, 1)
  # batch norm: batch statistics, beta only, eps 1e-3
  bn = nets.SlimBatchNorm(3)
  assert [n for n, _ in bn.named_parameters()] == ['beta']
  z = torch.randn(4, 3, 5, 5) * 3 + 2
  out = bn(z)
  assert abs(float(out.mean())) < 1e-5
  assert abs(float(out.var(unbiased=False)) - 1) < 1e-2
  up = nets.SlimConvTranspose2d(2, 3)
  assert up(torch.randn(1, 2, 4, 6)).shape == (1, 3, 8, 12)


def test_unet_and_heads_shapes_and_layout():
  torch.manual_seed(0)
  unet = nets.encoder_decoder_unet(nl_diff_enc_dec=3)
  head = nets.ldi_predictor(unet.out_channels, n_layers=2, n_layerwise_steps=3,
                            skip_channels=unet.skip_channels)
  imgs = torch.rand(1, 128, 128, 3)
  feat, feat_dec, skip_feat, end_points = unet(imgs)
  assert feat is None and feat_dec.shape == (1, 128, 16, 16)
  assert [s.shape[1] for s in skip_feat] == [512, 512, 256, 128, 64, 32]
  assert end_points['cnv7b'].shape == (1, 512, 1, 1)
  tex, masks, disps = head(feat_dec, skip_feat)
  assert tex.shape == (2, 1, 128, 128, 3) and disps.shape == (2, 1, 128, 128, 1)
  assert masks is None
  assert tex.stride()[3] == 1 and tex.stride()[4] == 128 * 128   # planar view
  assert float(tex.min()) >= 0 and float(tex.max()) <= 1          # sigmoid head
  with pytest.raises(ValueError):
    unet(torch.rand(1, 64, 64, 3))       # Appendix A.17: H, W multiples of 128
  masked = nets.ldi_predictor(128, n_layers=2, n_layerwise_steps=3,
                              skip_channels=unet.skip_channels, pred_masks=True)
  _, m, _ = masked(feat_dec, skip_feat)
  assert float(m[-1].min()) == 1.0        # background layer fully occupied


def _opts(tmpdir, **kw):
  sys.path.insert(0, PKG)
  import ldi_enc_dec as script
  args = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '1', '--n_layers', '2',
          '--img_height', '128', '--img_width', '128', '--num_iter', '2',
          '--indep_splat_wt', '0', '--compose_splat_wt', '0', '--cpu', 'true',
          '--log_freq', '1', '--save_latest_freq', '2',
          '--checkpoint_dir', str(tmpdir)]
  for k, v in kw.items():
    args += ['--' + k, str(v)]
  return script, script.apply_dataset_overrides(script.build_parser().parse_args(args))


def _plumbing_trainer(script, opts):
  """The training script's Trainer with a stand-in for compute_losses: the
  renderer and the six loss terms are HIP kernels (no CPU path; the real
  compute_losses is pinned on the GPU: tests/test_train_gpu.py), what runs here
  is everything around them -- model, optimiser, DDP, checkpoints, resume."""

  class PlumbingTrainer(script.Trainer):

    def compute_losses(self, staged):
      imgs_src, imgs_trg = staged[0], staged[1]
      ldi_src, ldi_trg = self.train_model(imgs_src, imgs_trg)
      total = ((ldi_src[0] - imgs_src).abs().mean() +
               (ldi_trg[0] - imgs_trg).abs().mean() +
               ldi_src[2].mean() + ldi_trg[2].mean())
      names = ('self_cons_loss', 'compose_splat_loss', 'indep_splat_loss',
               'incr_depth_loss', 'disp_smoothness_loss')
      scalars = {n: total.detach() * 0 for n in names}
      scalars['total_loss'] = total
      return total, scalars

  return PlumbingTrainer(opts)


def test_compute_losses_has_no_cpu_path(tmp_path):
  script, opts = _opts(tmp_path)
  tr = script.Trainer(opts)
  tr.setup()
  with pytest.raises(RuntimeError, match='ROCm device'):
    tr.train_step()


def test_eager_trainer_steps_saves_and_resumes(tmp_path):
  script, opts = _opts(tmp_path)
  assert opts.bg_layer_disp == 1e-3 and opts.max_disp == 0.4   # kitti overrides
  tr = _plumbing_trainer(script, opts)
  tr.setup()
  before = [p.detach().clone() for p in tr.model.parameters()]
  total, scalars = tr.train_step()
  assert set(scalars) == {'self_cons_loss', 'compose_splat_loss',
                          'indep_splat_loss', 'incr_depth_loss',
                          'disp_smoothness_loss', 'total_loss'}
  assert np.isfinite(float(total))
  assert all(p.grad is not None for p in tr.model.parameters())  # no dead params
  assert any(not torch.equal(a, b) for a, b in zip(before, tr.model.parameters()))
  tr.train()
  assert tr.global_step == 2
  assert os.path.exists(os.path.join(opts.checkpoint_dir, 'model.latest'))
  tr2 = _plumbing_trainer(script, opts)
  tr2.setup()
  assert tr2.global_step == 2            # auto-resume (train_utils.py:190-195)
  for a, b in zip(tr.model.parameters(), tr2.model.parameters()):
    assert torch.equal(a, b)


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _spawn(fn, args_of_port, nprocs):
  """mp.spawn on a port that was free a moment ago; if another process took it
  before the rendezvous bound it (EADDRINUSE), once more on a fresh one."""
  for attempt in range(4):
    try:
      return mp.spawn(fn, args=args_of_port(_free_port()), nprocs=nprocs, join=True)
    except Exception as e:  # pylint: disable=broad-except
      if 'EADDRINUSE' not in str(e) and 'address already in use' not in str(e).lower():
        raise
      if attempt == 3:
        raise


def _ddp_worker(rank, world, port, tmpdir, out, flat='false', complete='false'):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
  torch.set_num_threads(2)
  script, opts = _opts(tmpdir, num_iter=1, flat_grads=flat,
                       tf_checkpoint_complete=complete)
  tr = _plumbing_trainer(script, opts)
  tr.setup(backend='gloo')
  from torch.nn.parallel import DistributedDataParallel as DDP
  assert isinstance(tr.train_model, DDP) == (flat == 'false')
  tr.train_step()
  tr.train_step()
  flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()])
  out[rank] = (float(flat.double().sum()), float(flat.abs().double().sum()),
               float(tr.feed()[0].sum()))
  tr.dist.destroy_process_group()


def test_ddp_gradient_allreduce_two_gloo_ranks(tmp_path):
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  _spawn(_ddp_worker, lambda port: (world, port, str(tmp_path), out), world)
  # different data shards, identical parameters after the all-reduced step
  assert out[0][2] != out[1][2]
  assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
  # --flat_grads (the data-parallel step --hip_graph replays: no DDP wrapper,
  # every gradient a view of one buffer, one all-reduce): the same parameters
  out2 = mgr.dict()
  _spawn(_ddp_worker, lambda port: (world, port, str(tmp_path / 'flat'), out2, 'true'), world)
  assert out2[0][0] == out2[1][0] and out2[0][1] == out2[1][1]
  assert abs(out2[0][0] - out[0][0]) <= 1e-6 * abs(out[0][1])
  assert abs(out2[0][1] - out[0][1]) <= 1e-6 * abs(out[0][1])


def test_ddp_with_the_complete_tf_variable_list_two_gloo_ranks(tmp_path):
  """--tf_checkpoint_complete holds the variables the reference creates but
  never trains (the fc stack on the U-Net bottleneck, upcnv3 .. icnv1): all of
  them frozen, so DDP's reducer does not wait for them on the second step, and
  the trained parameters end where they end without the extra variables."""
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  _spawn(_ddp_worker, lambda port: (world, port, str(tmp_path), out, 'false', 'true'), world)
  assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
  script, opts = _opts(tmp_path / 'x', tf_checkpoint_complete='true')
  net = script.LdiNet(opts)
  frozen = [n for n, p in net.named_parameters() if not p.requires_grad]
  assert any('.fc.' in n for n in frozen) and any('upcnv1' in n for n in frozen)
  trainable = sum(p.numel() for p in net.parameters() if p.requires_grad)
  _, opts0 = _opts(tmp_path / 'y')
  assert trainable == sum(p.numel() for p in script.LdiNet(opts0).parameters())


def test_resume_picks_the_newest_checkpoint_and_pretrain_restore(tmp_path):
  """train_utils.py:176-200: resume from tf.train.latest_checkpoint (numbered
  or `latest`), else an optimistic (shape-tolerant) restore from
  ../<pretrain_name>/model-<pretrain_iter>."""
  import os
  import time
  import torch
  from lsi.nnutils import train_utils
  run, pre = tmp_path / 'run', tmp_path / 'base'
  os.makedirs(str(run)); os.makedirs(str(pre))
  assert train_utils.Trainer.latest_checkpoint(str(run)) is None
  torch.save({'model': {}, 'global_step': 50000}, str(run / 'model-50000'))
  assert train_utils.Trainer.latest_checkpoint(str(run)).endswith('model-50000')
  time.sleep(0.05)
  torch.save({'model': {}, 'global_step': 52000}, str(run / 'model.latest'))
  assert train_utils.Trainer.latest_checkpoint(str(run)).endswith('model.latest')
  a = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
  b = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 5))
  restored = train_utils.Trainer.optimistic_restore(b, a.state_dict())
  assert restored == ['0.bias', '0.weight']            # shapes of layer 1 differ
  assert torch.equal(b[0].weight, a[0].weight)


def test_batched_pair_pass_equals_two_passes(tmp_path):
  """LdiNet.forward runs source and target images through the network in ONE
  pass with per-view batch-norm statistics (nets.bn_groups); the reference makes
  two passes (ldi_enc_dec.py:175-228).  Same outputs, same gradients."""
  script, opts = _opts(tmp_path)
  torch.manual_seed(1)
  net = script.LdiNet(opts)
  src = torch.rand(2, 128, 128, 3)
  trg = torch.rand(2, 128, 128, 3)
  outs, grads = {}, {}
  for mode in (True, False):
    net.batched_pairs = mode
    net.zero_grad()
    a, b = net(src, trg)
    assert a[0].shape == (2, 2, 128, 128, 3) and b[2].shape == (2, 2, 128, 128, 1)
    outs[mode] = [a[0], a[2], b[0], b[2]]
    (a[0].sum() + 2 * a[2].sum() + 3 * b[0].sum() + 4 * b[2].sum()).backward()
    grads[mode] = [p.grad.clone() for p in net.parameters()]
  for x, y in zip(outs[True], outs[False]):
    torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-5)
  for x, y in zip(grads[True], grads[False]):
    torch.testing.assert_close(x, y, rtol=2e-3, atol=1e-4 * float(y.abs().max()) + 1e-7)
  # one batch-norm group per view: different statistics than the joint batch
  with nets_module().bn_groups(1):
    net.batched_pairs = False
    joint = net.predict(torch.cat([src, trg], 0))
  assert float((joint[0][:, :2] - outs[False][0]).abs().max()) > 1e-4


def nets_module():
  sys.path.insert(0, PKG)
  from lsi.nnutils import nets
  return nets


def test_strict_restore_tolerates_only_the_constant_fc_statistics():
  """Checkpoints written before SlimFC carried (constant, never updated)
  moving statistics still restore; anything else missing or unexpected is an
  error (train_utils.Trainer.strict_restore)."""
  import torch
  from lsi.nnutils import nets, train_utils
  torch.manual_seed(0)
  model = torch.nn.Sequential(nets.SlimFC(6, 4), nets.SlimFC(4, 3))
  state = model.state_dict()
  old = {k: v for k, v in state.items()
         if not (k.endswith('moving_mean') or k.endswith('moving_variance'))}
  assert len(old) < len(state)
  fresh = torch.nn.Sequential(nets.SlimFC(6, 4), nets.SlimFC(4, 3))
  train_utils.Trainer.strict_restore(fresh, old)
  assert torch.equal(fresh[0].fc.weight, model[0].fc.weight)
  broken = dict(old)
  broken.pop('0.fc.weight')
  with pytest.raises(RuntimeError, match='missing'):
    train_utils.Trainer.strict_restore(fresh, broken)
  with pytest.raises(RuntimeError, match='unexpected'):
    train_utils.Trainer.strict_restore(fresh, dict(old, extra=torch.zeros(1)))
