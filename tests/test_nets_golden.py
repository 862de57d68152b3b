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
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  _run_and_compare('unet', torch.device('cuda:0'), autocast=torch.bfloat16)
