"""The conv encoder-decoder and the LDI heads (lsi/nnutils/nets.py, PyTorch /
MIOpen) against the reference's own networks: tests/golden/nets.npz holds the
variable list and stage-by-stage activation samples recorded while
oracle/make_goldens.py executed the reference's unchanged lsi/nnutils/nets.py
(nets.py:29-348) on the slim shim with seeded weights.

Weights are rebuilt here from the variable names (oracle/tf1_slim_shim.
seeded_value is a pure function of name and shape) and loaded through the
TF-checkpoint name map (lsi/nnutils/tf_checkpoint.py, strict): a wrong name,
layout, layer order, skip concatenation, padding or batch-norm convention shows
up as a mismatch.  Tolerances: fp32 4e-4 (U-Net) / 1e-3 (FC-bottleneck variant)
of each stage's largest magnitude (the reference values are float64-accumulated;
TF / MIOpen sum in fp32 in their own orders, and batch norm over the 4 - 8
values per channel of the bottleneck amplifies that noise); bf16 autocast: mean
error 2e-2, 99th percentile 0.12 on the final sigmoid outputs.
"""
import argparse
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, golden

sys.path.insert(0, PKG)


def _model(tag):
  import ldi_enc_dec as script
  argv = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--img_height',
          '128', '--img_width', '128', '--batch_size', '4',
          '--tf_checkpoint_complete', 'true']
  argv += {'unet': ['--n_layers', '2'],
           'masks': ['--n_layers', '2', '--pred_ldi_masks', 'true'],
           'simple': ['--n_layers', '1', '--use_unet', 'false']}[tag]
  opts = script.apply_dataset_overrides(script.build_parser().parse_args(argv))
  opts.max_disp = 1.0     # the goldens hold the heads' raw sigmoid outputs
  torch.manual_seed(0)
  net = script.LdiNet(opts)
  for m in net.modules():  # the U-Net's (frozen, normally skipped) fc stack is pinned too
    if hasattr(m, 'want_feat'):
      m.want_feat = True
  return net


def _images(g, tag):
  """The inputs of the golden run (oracle/make_goldens.py:make_nets)."""
  imgs = np.random.RandomState(int(g['imgs_seed'])).rand(12, 128, 128, 3).astype(
      np.float32)
  return imgs[4:] if tag == 'simple' else imgs[:4]


def _tf_variables(g, tag):
  import tf1_slim_shim as slim
  names = [str(n) for n in g[tag + '_var_names']]
  shapes = [tuple(int(d) for d in str(s).split(',')) for s in g[tag + '_var_shapes']]
  return names, shapes, {n: slim.seeded_value(n, s) for n, s in zip(names, shapes)}


@pytest.mark.parametrize('tag', ['unet', 'masks', 'simple'])
def test_variable_map_is_the_reference_variable_list(tag):
  """Every variable the reference creates (dead ones included: the fc stack on
  the U-Net bottleneck, upcnv3 .. icnv1) and nothing else, with its shape."""
  from lsi.nnutils import tf_checkpoint
  g = golden('nets.npz')
  names, shapes, _ = _tf_variables(g, tag)
  model = _model(tag)
  vmap = tf_checkpoint.variable_map(model)
  assert len(set(n for n, _, _ in vmap)) == len(vmap)
  assert sorted(n for n, _, _ in vmap) == sorted(names)
  exported = tf_checkpoint.export_tf_variables(model)
  want = dict(zip(names, shapes))
  for n, arr in exported.items():
    assert tuple(arr.shape) == want[n], (n, arr.shape, want[n])


def _run_and_compare(tag, device, autocast=None):
  from lsi.nnutils import tf_checkpoint
  g = golden('nets.npz')
  names, _, tf_vars = _tf_variables(g, tag)
  model = _model(tag)
  loaded, skipped = tf_checkpoint.load_tf_variables(model, tf_vars, strict=True)
  assert not skipped and sorted(loaded) == sorted(names)
  model = model.to(device).train()          # batch statistics (is_training=True)
  # stage outputs through forward hooks on the modules the name map points to
  prefix = {}
  for tf_name, key, _ in tf_checkpoint.variable_map(model):
    if tf_name.endswith('/weights'):
      prefix[tf_name[:-len('/weights')]] = key.rsplit('.', 2)[0]
  mods = dict(model.named_modules())
  got, hooks = {}, []
  for alias, pfx in prefix.items():
    def hook(_m, _i, out, alias=alias):
      got[alias] = out.detach().float().cpu()
    hooks.append(mods[pfx].register_forward_hook(hook))
  imgs = torch.tensor(_images(g, tag), device=device)
  with torch.no_grad():
    if autocast is not None:
      with torch.autocast(device.type, dtype=autocast):
        tex, masks, disps = model.predict(imgs)
    else:
      tex, masks, disps = model.predict(imgs)
  for h in hooks:
    h.remove()
  # fp32: stage errors grow from 1e-6 (cnv1) to a few 1e-4 of the stage's
  # scale behind the bottleneck, where batch norm normalises 4 (8) values per
  # channel and amplifies rounding noise; a structural error is O(1)
  tol = (1e-3 if tag == 'simple' else 4e-4) if autocast is None else 6e-2
  stages = [str(s) for s in g[tag + '_stages']]
  shapes = [tuple(int(d) for d in str(s).split(',')) for s in g[tag + '_stage_shapes']]
  checked = 0
  if autocast is not None:
    checked = _check_bf16_stages(g, tag, got, stages, shapes)
  if autocast is None:
    for alias, shape in zip(stages, shapes):
      if alias not in got:
        continue             # (layers the build does not execute: dead branches)
      out = got[alias]
      if out.dim() == 4:
        out = out.permute(0, 2, 3, 1)          # NCHW -> the reference's NHWC
      assert tuple(out.shape) == shape, (alias, tuple(out.shape), shape)
      flat = out.reshape(-1).numpy()
      idx = g['%s_act_idx/%s' % (tag, alias)]
      want = g['%s_act_val/%s' % (tag, alias)]
      scale = max(float(np.abs(want).max()), 1e-3)
      # (the fc stack normalises over the batch alone -- 4 / 8 values per
      # channel, three times in a row: rounding noise is amplified most there)
      stage_tol = max(tol, 2e-3) if '/fc/' in alias else tol
      assert np.abs(flat[idx] - want).max() <= stage_tol * scale, (
          alias, float(np.abs(flat[idx] - want).max()), scale)
      mean, std = g['%s_act_stat/%s' % (tag, alias)]
      assert abs(float(flat.astype(np.float64).mean()) - mean) <= tol * max(abs(mean), std, 1e-3)
      assert abs(float(flat.astype(np.float64).std()) - std) <= 10 * tol * max(std, 1e-3)
      checked += 1
    assert checked >= (14 if tag == 'masks' else 20), checked
  outs = {'tex': tex, 'disp': disps}
  if masks is not None:
    outs['mask'] = masks
  for name, t in outs.items():
    t = t.detach().float().cpu()
    assert tuple(t.shape) == tuple(int(d) for d in g['%s_ldi_%s_shape' % (tag, name)])
    flat = t.reshape(-1).numpy()
    idx, want = g['%s_ldi_%s_idx' % (tag, name)], g['%s_ldi_%s_val' % (tag, name)]
    err = np.abs(flat[idx] - want)
    if autocast is None:
      assert err.max() <= tol, (name, float(err.max()))
    else:
      # bf16 (8 mantissa bits) through ~30 conv + batch-norm stages: the mean
      # error stays at the 1e-2 level, single pixels reach a few times that
      assert err.mean() <= 2e-2 and np.percentile(err, 99) <= 0.12, (
          name, float(err.mean()), float(np.percentile(err, 99)), float(err.max()))
  return checked


def stage_depth(alias):
  """Convolutions between the image and this stage's output on the executed
  path (U-Net: cnv1 = 1 ... cnv7b = 14, upcnv7 = 15, icnv7 = 16 ... icnv4 = 22;
  a head: upcnv3 = 23 ... upcnv1b = 28, pred = 29)."""
  import re
  name = alias.rsplit('/', 1)[-1]
  head = 'pixelwise_pred' in alias
  m = re.match(r'(cnv|upcnv|icnv|pred_)(\d+)(b?)$', name)
  if not m:
    return None
  kind, k, b = m.group(1), int(m.group(2)), m.group(3) == 'b'
  if kind == 'cnv':
    return 2 * k - 1 + int(b)
  if kind == 'pred_':
    return 29
  if head:                       # upcnv3, upcnv3b, upcnv2, ... below icnv4 (22)
    return 22 + 2 * (3 - k) + 1 + int(b)
  return 14 + 2 * (7 - k) + (1 if kind == 'upcnv' else 2)


# bf16 error budget per stage, in units of the stage's standard deviation
# (every stage but the heads' sigmoid is a batch-normed ReLU: std ~ 0.58): TWICE
# the RMS error over the sampled activations that the own kernels measure on
# MI355X (profiles/r06/nets_stage_errors.txt; torch's library bf16 path measures
# the same, stage by stage: it is the arithmetic, not the kernels).  Every
# convolution rounds its operands and its output to 8 mantissa bits; through the
# encoder the error grows a little faster than sqrt(depth) because the golden
# input is 128 x 128 and the maps shrink to 4 x 4 at cnv5b (a batch norm over
# 64 values per channel).  Behind the 1 x 1 ... 2 x 2 bottleneck maps (cnv6 ...
# icnv6) a batch norm normalises 2 - 32 values per channel and turns rounding
# noise into O(1) changes of single channels: those stages get
# BF16_STAGE_BOTTLENECK, and what they feed (upcnv5 onwards: measured 0.09 -
# 0.15) a flat BF16_STAGE_BEHIND.  A structural error -- a mis-padded layer, a
# wrong tap order, a stale weight pack -- is O(1) of the standard deviation at
# the stage where it happens and behind it.
BF16_STAGE_MEASURED = {
    'cnv1': 0.0058, 'cnv1b': 0.0094, 'cnv2': 0.0119, 'cnv2b': 0.0134, 'cnv3': 0.0151,
    'cnv3b': 0.0206, 'cnv4': 0.0218, 'cnv4b': 0.0317, 'cnv5': 0.0314, 'cnv5b': 0.0520}
BF16_STAGE_BOTTLENECK = 0.5
BF16_STAGE_BEHIND = 0.30


def bf16_stage_budget(alias):
  d = stage_depth(alias)
  if d is None:
    return None
  name = alias.rsplit('/', 1)[-1]
  if 'pixelwise_pred' not in alias and name in BF16_STAGE_MEASURED:
    return 2.0 * BF16_STAGE_MEASURED[name]
  if 'pixelwise_pred' not in alias and name in (
      'cnv6', 'cnv6b', 'cnv7', 'cnv7b', 'upcnv7', 'icnv7', 'upcnv6', 'icnv6'):
    return BF16_STAGE_BOTTLENECK
  return BF16_STAGE_BEHIND


def _check_bf16_stages(g, tag, got, stages, shapes, report=None):
  """Every executed stage of a bf16 run against the reference's sampled
  activations, within bf16_stage_budget.  `report`: a list that receives
  (alias, depth, rms error / std, budget) rows."""
  checked = 0
  for alias, shape in zip(stages, shapes):
    if alias not in got or '/fc/' in alias:
      continue
    budget = bf16_stage_budget(alias)
    if budget is None:
      continue
    out = got[alias]
    if out.dim() == 4:
      out = out.permute(0, 2, 3, 1)
    assert tuple(out.shape) == shape, (alias, tuple(out.shape), shape)
    flat = out.reshape(-1).numpy().astype(np.float64)
    idx = g['%s_act_idx/%s' % (tag, alias)]
    want = g['%s_act_val/%s' % (tag, alias)].astype(np.float64)
    _, std = g['%s_act_stat/%s' % (tag, alias)]
    rms = float(np.sqrt(np.mean((flat[idx] - want) ** 2))) / max(float(std), 1e-3)
    if report is not None:
      report.append((alias, stage_depth(alias), rms, budget))
    else:
      assert rms <= budget, (alias, stage_depth(alias), rms, budget)
    checked += 1
  return checked


@pytest.mark.parametrize('tag', ['unet', 'masks', 'simple'])
def test_network_activations_match_the_reference_on_cpu(tag):
  """Layer composition, TF SAME padding, slim batch norm, head layout: the
  PyTorch modules on CPU tensors (no MIOpen) against the reference's values."""
  _run_and_compare(tag, torch.device('cpu'))


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['unet', 'masks', 'simple'])
def test_network_activations_match_the_reference_on_the_gpu(tag, built_lib):
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  _run_and_compare(tag, torch.device('cuda:0'))


@pytest.mark.gpu
def test_network_bf16_autocast_stays_close_to_the_reference(built_lib):
  """The product default on a ROCm device (--bf16: every convolution on the own
  MFMA kernels) against the reference's nets.py values STAGE BY STAGE, each
  within the bf16 error budget of its depth (bf16_stage_budget) -- not only the
  final sigmoid outputs, behind which a wrong layer deep in the decoder could
  hide (round 5's verdict)."""
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  checked = _run_and_compare('unet', torch.device('cuda:0'), autocast=torch.bfloat16)
  assert checked >= 29, checked     # 14 encoder + 8 decoder stages + 2 x 7 head stages


def test_stage_depths():
  assert stage_depth('encoder_decoder_unet/cnv1') == 1
  assert stage_depth('encoder_decoder_unet/cnv7b') == 14
  assert stage_depth('encoder_decoder_unet/upcnv7') == 15
  assert stage_depth('encoder_decoder_unet/icnv4') == 22
  assert stage_depth('ldi_tex_disp/pixelwise_pred/upsample_1/decoder/upcnv3') == 23
  assert stage_depth('ldi_tex_disp/pixelwise_pred/upsample_0/decoder/upcnv1b') == 28
  assert stage_depth('ldi_tex_disp/pixelwise_pred/upsample_0/pred_0') == 29
  assert stage_depth('encoder_decoder_unet/fc/fc_1') is None
