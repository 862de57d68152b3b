"""Names and layouts of the reference's TF-1 checkpoint variables for the
PyTorch modules of lsi.nnutils.nets: import / export of `{tf_name: ndarray}`
dictionaries (what `tf.train.load_checkpoint(...).get_tensor(name)` yields; the
TF reader itself is not a dependency -- dump a checkpoint to .npz where TF
exists, load it here).

Variable scopes of the reference (slim):
  encoder_decoder_unet/<layer>/{weights, BatchNorm/{beta,moving_mean,moving_variance}}
      <layer> in cnv1 .. cnv7b, upcnv7 .. upcnv1, icnv7 .. icnv1   (nets.py:244-348)
  encoder_decoder_unet/fc/fc_{1,2,3}/...                        (nets.py:289-291)
  ldi_tex_disp/pixelwise_pred/upsample_<l>/decoder/upcnv<k>[b]/... (nets.py:73-114,
      117-161, 164-208)
  ldi_tex_disp/pixelwise_pred/upsample_<l>/pred_<l>/{weights, biases}
  (non-U-Net variant, nets.py:211-241: top-level scopes encoder/..., decoder/...)
Layouts: slim.conv2d weights are [kh, kw, in, out] (torch: [out, in, kh, kw]);
slim.conv2d_transpose weights are [kh, kw, out, in] (torch ConvTranspose2d:
[in, out, kh, kw]); fully_connected [in, out] (torch Linear: [out, in]).
Adam slots are not in the reference's checkpoints (train_utils.py:172-174).
"""
import numpy as np
import torch

from lsi.nnutils import nets


def _layer_entries(tf_scope, torch_prefix, module):
  """(tf_name, torch_key, kind) triples of one Slim* layer."""
  out = []
  if isinstance(module, nets.SlimConv2d):
    out.append((tf_scope + '/weights', torch_prefix + '.conv.weight', 'conv'))
    if module.conv.bias is not None:
      out.append((tf_scope + '/biases', torch_prefix + '.conv.bias', 'vec'))
    bn = torch_prefix + '.bn' if module.bn is not None else None
  elif isinstance(module, nets.SlimConvTranspose2d):
    out.append((tf_scope + '/weights', torch_prefix + '.conv.weight', 'convT'))
    bn = torch_prefix + '.bn'
  elif isinstance(module, nets.SlimFC):
    out.append((tf_scope + '/weights', torch_prefix + '.fc.weight', 'fc'))
    bn = torch_prefix
  else:
    return out
  if bn is not None:
    out.append((tf_scope + '/BatchNorm/beta', bn + '.beta', 'vec'))
    out.append((tf_scope + '/BatchNorm/moving_mean', bn + '.moving_mean', 'vec'))
    out.append((tf_scope + '/BatchNorm/moving_variance',
                bn + '.moving_variance', 'vec'))
  return out


def variable_map(model):
  """[(tf_name, torch state_dict key, kind)] for an LDI predictor model with
  attributes `enc_dec` (EncoderDecoderUnet | EncoderDecoderSimple) and
  `ldi_tex_disp` (LdiPredictor) -- ldi_enc_dec.LdiNet."""
  entries = []
  enc_dec = model.enc_dec
  if isinstance(enc_dec, nets.EncoderDecoderUnet):
    top = 'encoder_decoder_unet'
    for name, mod in enc_dec.encoder.named_children():
      entries += _layer_entries('%s/%s' % (top, name), 'enc_dec.encoder.' + name,
                                mod)
    for name, mod in enc_dec.named_children():
      if name.startswith('upcnv') or name.startswith('icnv'):
        entries += _layer_entries('%s/%s' % (top, name), 'enc_dec.' + name, mod)
    if enc_dec.fc is not None:
      for i, mod in enumerate(enc_dec.fc):
        entries += _layer_entries('%s/fc/fc_%d' % (top, i + 1),
                                  'enc_dec.fc.%d' % i, mod)
  else:
    # encoder_decoder_simple opens no scope of its own (nets.py:211-241): its
    # variables live under the top-level scopes `encoder` and `decoder`
    enc = enc_dec.encoder
    for name, mod in enc.encoder.named_children():
      entries += _layer_entries('encoder/%s' % name,
                                'enc_dec.encoder.encoder.' + name, mod)
    for i, mod in enumerate(enc.fc):
      entries += _layer_entries('encoder/fc/fc_%d' % (i + 1),
                                'enc_dec.encoder.fc.%d' % i, mod)
    for name, mod in enc_dec.decoder.named_children():
      entries += _layer_entries('decoder/%s' % name,
                                'enc_dec.decoder.' + name, mod)
  pp = model.ldi_tex_disp.pixelwise_pred
  for l, (dec, head) in enumerate(zip(pp.decoders, pp.preds)):
    scope = 'ldi_tex_disp/pixelwise_pred/upsample_%d' % l
    for name, mod in dec.named_children():
      entries += _layer_entries(
          '%s/decoder/%s' % (scope, name),
          'ldi_tex_disp.pixelwise_pred.decoders.%d.%s' % (l, name), mod)
    entries += _layer_entries('%s/pred_%d' % (scope, l),
                              'ldi_tex_disp.pixelwise_pred.preds.%d' % l, head)
  return entries


def _to_torch(arr, kind):
  arr = np.asarray(arr)
  if kind == 'conv':     # [kh, kw, in, out] -> [out, in, kh, kw]
    return np.transpose(arr, (3, 2, 0, 1))
  if kind == 'convT':    # [kh, kw, out, in] -> [in, out, kh, kw]
    return np.transpose(arr, (3, 2, 0, 1))
  if kind == 'fc':       # [in, out] -> [out, in]
    return np.transpose(arr, (1, 0))
  return arr


def _to_tf(arr, kind):
  arr = np.asarray(arr)
  if kind in ('conv', 'convT'):
    return np.transpose(arr, (2, 3, 1, 0))
  if kind == 'fc':
    return np.transpose(arr, (1, 0))
  return arr


def export_tf_variables(model):
  """{tf_name: ndarray} in TF layouts for every mapped variable."""
  sd = model.state_dict()
  return {tf_name: _to_tf(sd[key].detach().cpu().numpy(), kind)
          for tf_name, key, kind in variable_map(model)}


def load_tf_variables(model, tf_vars, strict=False):
  """Copies a {tf_name: ndarray} dictionary into the model.  Like the
  reference's optimistic_restorer (helpers.py:27-62): variables that are absent
  or whose shape differs are skipped (strict=True raises instead).  Returns
  (loaded, skipped) lists of TF names."""
  sd = model.state_dict()
  loaded, skipped = [], []
  with torch.no_grad():
    for tf_name, key, kind in variable_map(model):
      if tf_name not in tf_vars:
        if strict:
          raise KeyError('checkpoint has no variable %r' % tf_name)
        skipped.append(tf_name)
        continue
      val = _to_torch(tf_vars[tf_name], kind)
      if tuple(val.shape) != tuple(sd[key].shape):
        if strict:
          raise ValueError('%s: shape %s, model wants %s' %
                           (tf_name, val.shape, tuple(sd[key].shape)))
        skipped.append(tf_name)
        continue
      sd[key].copy_(torch.as_tensor(np.ascontiguousarray(val),
                                    dtype=sd[key].dtype))
      loaded.append(tf_name)
  return loaded, skipped
