"""CNN definitions (mirror of the reference's lsi/nnutils/nets.py) as
torch.nn.Modules on PyTorch-ROCm.  On a ROCm device with bf16 channels-last
activations (the trainer's default: --bf16 / --channels_last) every convolution
runs on this repo's own MFMA kernels (csrc/lsi_conv*.hip: first layer, implicit
GEMM, the heads' 32-channel layers, all weight gradients) and every batch norm
on csrc/lsi_bn.hip; fp32 activations (--bf16 false: the reference's own
arithmetic) go through the library (MIOpen) -- DESIGN.md 4.8 says why.

Conventions kept from the reference (tf.contrib.slim, nets.py:29-348):
  * tensors at the module boundary are B x H x W x C (channels-last logical
    shape); modules permute to NCHW views internally -- with the
    channels_last memory format that permute is free;
  * slim.conv2d: TF 'SAME' padding (asymmetric for stride 2: k=7 pads 2/3,
    k=5 pads 1/2, k=3 pads 0/1), no bias when followed by batch norm, ReLU,
    xavier-uniform weights;
  * slim.batch_norm defaults: no scale (gamma), learned offset (beta) only,
    epsilon 1e-3, and -- because the reference's train op never runs the
    UPDATE_OPS (train_utils.py:107-117) and evaluation defaults to
    batch_norm_training=True (ldi_pred_eval.py:45-46) -- always batch
    statistics; the moving averages stay at their initial values;
  * slim.conv2d_transpose [4,4] stride 2 'SAME' == ConvTranspose2d(4, 2, 1);
  * the l2 weights_regularizer is declared but never added to the loss
    (ldi_enc_dec.py:398-410): no weight decay.
Parameters the reference creates but never trains when n_layerwise_steps=3
(the `fc` stack on the bottleneck, upcnv3..icnv1) are only built on request
(`with_fc`, `nl_diff_enc_dec`), so that DDP's reducer never waits for them.
"""
import math
import os

import torch
from torch import nn
import torch.nn.functional as F

from lsi.nnutils import helpers as nn_helpers


def _same_pad(n, k, stride):
  """TF 'SAME' padding (before, after) along one axis of length n:
  total = max((ceil(n/stride) - 1)*stride + k - n, 0), the extra pixel after."""
  total = max((-(-n // stride) - 1) * stride + k - n, 0)
  return total // 2, total - total // 2


# symmetric SAME padding inside the convolution (see SlimConv2d);
# LSI_IMPLICIT_PAD=0 pads explicitly
IMPLICIT_PAD = os.environ.get('LSI_IMPLICIT_PAD', '1') != '0'
# The 3x3 convolutions over 32 channels at full resolution (`upcnv1b`, `pred_l`
# of every LDI layer) on the hand-written MFMA kernel (csrc/lsi_conv.hip) when
# the activations are bf16 channels-last on the GPU; LSI_MFMA_CONV=0: MIOpen
MFMA_CONV = os.environ.get('LSI_MFMA_CONV', '1') != '0'
# bf16 batch norm on large maps under autocast (LSI_BF16_BN=0 restores fp32)
BF16_BATCH_NORM = os.environ.get('LSI_BF16_BN', '1') != '0'
# Every other convolution with channel counts that are multiples of 32 on the
# implicit-GEMM MFMA kernel (csrc/lsi_conv_igemm.hip); LSI_IGEMM_CONV=0: MIOpen.
# LSI_IGEMM_MIN_PIXELS=1024 sends maps with fewer pixels (batch included; on the
# smaller side of the layer) to the library: the bottleneck layers `cnv6b`,
# `cnv7b`, `icnv7`, `upcnv6/7` are small GEMMs with few tiles for this kernel,
# where MIOpen is 1.1 - 1.4 x ahead -- 77 us of a 10 ms step in all
# (tools/conv_bench.py: profiles/r05/conv_bench.txt); by default they too run on
# the own kernel.
IGEMM_CONV = os.environ.get('LSI_IGEMM_CONV', '1') != '0'
IGEMM_MIN_PIXELS = int(os.environ.get('LSI_IGEMM_MIN_PIXELS', '0'))


def _igemm_pays(x, stride):
  n, _, h, w = x.shape
  return n * (-(-h // stride)) * (-(-w // stride)) >= IGEMM_MIN_PIXELS
# batch norm + ReLU of the conv layers as the fused HIP kernels (csrc/lsi_bn.hip)
# for channels-last activations on the GPU; LSI_FUSED_BN=0 keeps MIOpen's
FUSED_BN = os.environ.get('LSI_FUSED_BN', '1') != '0'


# Batch-statistics groups: the reference runs the network once on the source and
# once on the target images (ldi_enc_dec.py:175-228), so every batch norm sees
# one view's images only.  Inside `bn_groups(g)` a batch is g such sub-batches
# stacked along N, and every batch norm normalises each with its own statistics:
# the two passes become one (half the kernel launches, convolutions at twice the
# batch) with the same arithmetic.
_BN_GROUPS = [1]


class bn_groups(object):
  """with bn_groups(2): net(torch.cat([src, trg]))  ==  net(src), net(trg)."""

  def __init__(self, groups):
    self.groups = int(groups)

  def __enter__(self):
    self.saved = _BN_GROUPS[0]
    _BN_GROUPS[0] = self.groups
    return self

  def __exit__(self, *exc):
    _BN_GROUPS[0] = self.saved
    return False


def _per_group(fn, x):
  """fn on each of the _BN_GROUPS[0] sub-batches of x (views), concatenated."""
  g = _BN_GROUPS[0]
  if g <= 1:
    return fn(x)
  if x.shape[0] % g:
    raise ValueError('batch %d does not split into %d batch-norm groups' %
                     (x.shape[0], g))
  return torch.cat([fn(c) for c in x.chunk(g, dim=0)], dim=0)


def _bn_relu(bn, x, prestat=False):
  """relu(bn(x)): two HIP passes forward, two backward (statistics, normalise +
  ReLU + store in x's dtype) where the layout allows, else torch / MIOpen.
  prestat: the convolution kernel has left the statistics (see _stats_bn): only
  the second pass runs."""
  if FUSED_BN and bn.is_training and x.is_cuda:
    from lsi.nnutils import _hip_bn  # pylint: disable=g-import-not-at-top
    if _hip_bn.supported(x, _BN_GROUPS[0]):
      if not prestat:
        return _hip_bn.batch_norm_relu(x, bn.beta, bn.eps, True, _BN_GROUPS[0], False)
      try:
        return _hip_bn.batch_norm_relu(x, bn.beta, bn.eps, True, _BN_GROUPS[0], True)
      except Exception:
        # (the statistics the convolution left must not outlive this call: the
        # kernels would refuse the workspace -- NaN -- until somebody cleans it)
        _hip_bn.discard_stats(tuple(x.shape), x.device, 1, _BN_GROUPS[0])
        raise
    if prestat:
      # the producer accumulated for a consumer that cannot take this tensor
      # (layout, alignment): drop its sums, normalise the two-pass way
      _hip_bn.discard_stats(tuple(x.shape), x.device, 1, _BN_GROUPS[0])
  elif prestat:
    raise RuntimeError('batch-norm statistics were requested from the convolution '
                       'but the fused batch norm is off')
  return F.relu(bn(x))


CONV_BN_STATS = os.environ.get('LSI_CONV_BN_STATS', '1') != '0'


def _stats_bn(bn, activation, n, cout):
  """The sub-batch groups when the implicit-GEMM kernel may accumulate the batch
  statistics of its output for the batch norm + ReLU behind it (training
  statistics, the fused batch-norm kernels, whole images per group), else 0."""
  if not (CONV_BN_STATS and FUSED_BN and bn is not None and bn.is_training and
          activation == 'relu'):
    return 0
  from lsi.nnutils import _hip_bn  # pylint: disable=g-import-not-at-top
  g = _BN_GROUPS[0]
  if g < 1 or n % g or not _hip_bn.channels_ok(cout):
    return 0
  return g


class SlimBatchNorm(nn.Module):
  """slim.batch_norm defaults: center=True, scale=False, epsilon=1e-3."""

  def __init__(self, channels, eps=1e-3):
    super().__init__()
    self.beta = nn.Parameter(torch.zeros(channels))
    # constant gamma = 1 (slim: scale=False); a buffer, never trained.  Passed
    # explicitly because MIOpen's batch-norm backward needs a weight tensor.
    self.register_buffer('gamma', torch.ones(channels))
    self.register_buffer('moving_mean', torch.zeros(channels))
    self.register_buffer('moving_variance', torch.ones(channels))
    self.eps = eps
    self.is_training = True

  def forward(self, x):
    if self.is_training and _BN_GROUPS[0] > 1:
      return _per_group(self._forward, x)
    return self._forward(x)

  def _forward(self, x):
    # Under bf16 autocast the large maps stay in bf16 (MIOpen computes the
    # statistics in fp32 internally; gamma / beta are fp32): no cast to fp32 and
    # back around every batch norm, which was a tenth of the bf16 step
    # (profiles/r02/train_step_summary.json).  Small maps go through fp32: the
    # reference is fp32, and MIOpen's bf16 batch norm crashed on the small
    # bottleneck maps.
    if (x.dtype == torch.bfloat16 and self.is_training and BF16_BATCH_NORM and
        x.numel() // x.shape[1] >= 4096):
      return F.batch_norm(x, None, None, self.gamma, self.beta, True, 0.0,
                          self.eps)
    x = x.float()
    if self.is_training:
      if x.numel() // x.shape[1] > 64:
        return F.batch_norm(x, None, None, self.gamma, self.beta, True, 0.0,
                            self.eps)
      # few values per channel (the 1x1 .. 2x6 bottleneck maps; with one value
      # TF normalises with variance 0 and torch's fused kernel refuses): the
      # same arithmetic by hand
      var, mean = torch.var_mean(x, dim=(0, 2, 3), unbiased=False, keepdim=True)
      return (x - mean) * torch.rsqrt(var + self.eps) + self.beta.view(1, -1, 1, 1)
    return F.batch_norm(x, self.moving_mean, self.moving_variance, self.gamma,
                        self.beta, False, 0.0, self.eps)


class SlimConv2d(nn.Module):
  """slim.conv2d(inputs, num_outputs, [k, k], stride) with the arg_scope of
  nets.py (batch norm + ReLU), or bias + custom activation when normalizer is
  None (the `pred_l` heads, nets.py:143-154)."""

  def __init__(self, cin, cout, k, stride=1, batch_norm=True, activation='relu'):
    super().__init__()
    self.k, self.stride = k, stride
    self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=0,
                          bias=not batch_norm)
    nn.init.xavier_uniform_(self.conv.weight)
    if self.conv.bias is not None:
      nn.init.zeros_(self.conv.bias)
    self.bn = SlimBatchNorm(cout) if batch_norm else None
    self.activation = activation

  def forward(self, x, x2=None):
    """x2: the layer reads tf.concat([x, x2], axis=3) (a skip connection)."""
    if x2 is not None:
      return self.forward_cat(x, x2)
    if (MFMA_CONV and IGEMM_CONV and x.is_cuda and self.bn is not None and
        self.conv.weight.shape[1] <= 4 and
        (x.dtype == torch.bfloat16 or
         (x.dtype == torch.float32 and torch.is_autocast_enabled('cuda') and
          torch.get_autocast_dtype('cuda') == torch.bfloat16))):
      # `cnv1`: the image itself (fp32 under autocast: rounded to bf16 inside the
      # kernel, as autocast's cast would) through the first-layer kernel
      from lsi.nnutils import _hip_conv  # pylint: disable=g-import-not-at-top
      cout, cin = self.conv.weight.shape[:2]
      if _hip_conv.first_supported(x, cin, cout, self.k, self.stride):
        ph = _same_pad(x.shape[2], self.k, self.stride)
        pw = _same_pad(x.shape[3], self.k, self.stride)
        st = _stats_bn(self.bn, self.activation, x.shape[0], cout)
        y = _hip_conv.conv2d_first(x, self.conv.weight, self.stride, ph[0], pw[0],
                                   -(-x.shape[2] // self.stride),
                                   -(-x.shape[3] // self.stride), st)
        if st:
          return _bn_relu(self.bn, y, True)
        return self._bn_act(y)
    if MFMA_CONV and x.is_cuda and x.dtype == torch.bfloat16:
      from lsi.nnutils import _hip_conv  # pylint: disable=g-import-not-at-top
      cout, cin = self.conv.weight.shape[:2]
      head = self.bn is None and self.activation == 'sigmoid'
      if (head or self.bn is not None) and _hip_conv.supported(
          x, cin, cout, self.k, self.stride, head):
        if head:    # bias + sigmoid inside the kernel, fp32 RGBD pixels out
          return _hip_conv.conv3x3_c32_sigmoid(x, self.conv.weight, self.conv.bias)
        x = _hip_conv.conv3x3_c32(x, self.conv.weight)
        return self._bn_act(x)
      if (IGEMM_CONV and self.bn is not None and
          _hip_conv.igemm_supported(x, cin, cout, self.k, self.stride) and
          _igemm_pays(x, self.stride)):
        # every other batch-normed layer: the implicit-GEMM kernel (forward and
        # data gradient; weight gradient on lsi_conv3x3_wgrad or the library)
        ph = _same_pad(x.shape[2], self.k, self.stride)
        pw = _same_pad(x.shape[3], self.k, self.stride)
        st = _stats_bn(self.bn, self.activation, x.shape[0], cout)
        x = _hip_conv.conv2d(x, self.conv.weight, self.stride, ph[0], pw[0],
                             -(-x.shape[2] // self.stride), -(-x.shape[3] // self.stride), st)
        if st:
          return _bn_relu(self.bn, x, True)
        return self._bn_act(x)
      if (self.bn is not None and torch.is_grad_enabled() and
          self.conv.weight.requires_grad and
          _hip_conv.wgrad_supported(x, cin, cout, self.k, self.stride)):
        # forward and data gradient on the library, weight gradient on the
        # matrix-core kernel (K = pixels)
        x = _hip_conv.conv3x3_lib_own_wgrad(x, self.conv.weight)
        return self._bn_act(x)
    ph = _same_pad(x.shape[2], self.k, self.stride)
    pw = _same_pad(x.shape[3], self.k, self.stride)
    # Symmetric SAME padding (the stride-1 layers) goes into the convolution:
    # no padded copy of the activation (118 -> 127 samples/s fp32, 182 -> 192
    # bf16, tools/train_bench.py).  (In round 1 MIOpen fell back to naive
    # kernels for several bf16 shapes with implicit padding; with channels-last
    # activations and bf16 batch norm it no longer does.)  Asymmetric padding
    # (the stride-2 layers: TF pads one more pixel after) stays explicit.
    if IMPLICIT_PAD and ph[0] == ph[1] and pw[0] == pw[1]:
      x = F.conv2d(x, self.conv.weight, self.conv.bias, self.stride,
                   (ph[0], pw[0]))
    else:
      if ph[0] or ph[1] or pw[0] or pw[1]:
        x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
      x = self.conv(x)
    return self._bn_act(x)

  def forward_cat(self, x1, x2):
    """forward(tf.concat([x1, x2], axis=3)) -- the skip connections -- with the
    convolution kernels reading the two tensors where they can."""
    if (MFMA_CONV and IGEMM_CONV and self.bn is not None and x1.is_cuda and
        x1.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16):
      from lsi.nnutils import _hip_conv  # pylint: disable=g-import-not-at-top
      cout = self.conv.weight.shape[0]
      if (_hip_conv.cat_supported(x1, x2, cout, self.k, self.stride) and
          _igemm_pays(x1, self.stride) and
          _hip_conv._igemm_wgrad_bytes(_hip_conv._conv_desc(
              x1.shape[0], x1.shape[2], x1.shape[3], x1.shape[1] + x2.shape[1],
              -(-x1.shape[2] // self.stride), -(-x1.shape[3] // self.stride), cout,
              self.k, self.k, self.stride, _same_pad(x1.shape[2], self.k, self.stride)[0],
              _same_pad(x1.shape[3], self.k, self.stride)[0])) > 0):
        ph = _same_pad(x1.shape[2], self.k, self.stride)
        pw = _same_pad(x1.shape[3], self.k, self.stride)
        st = _stats_bn(self.bn, self.activation, x1.shape[0], cout)
        y = _hip_conv.conv2d_cat(x1, x2, self.conv.weight, self.stride, ph[0], pw[0],
                                 -(-x1.shape[2] // self.stride),
                                 -(-x1.shape[3] // self.stride), st)
        if st:
          return _bn_relu(self.bn, y, True)
        return self._bn_act(y)
    return self.forward(torch.cat([x1, x2], dim=1))

  def _bn_act(self, x):
    """Batch norm (if any) and the activation behind the convolution."""
    if self.bn is not None and self.activation == 'relu':
      return _bn_relu(self.bn, x)
    if self.bn is not None:
      x = self.bn(x)
    if self.activation == 'relu':
      return F.relu(x)
    if self.activation == 'sigmoid':
      return torch.sigmoid(x)
    return x


class SlimConvTranspose2d(nn.Module):
  """slim.conv2d_transpose(inputs, num_outputs, [4, 4], stride=2) + BN + ReLU."""

  def __init__(self, cin, cout):
    super().__init__()
    self.conv = nn.ConvTranspose2d(cin, cout, 4, stride=2, padding=1,
                                   bias=False)
    nn.init.xavier_uniform_(self.conv.weight)
    self.bn = SlimBatchNorm(cout)

  def forward(self, x):
    if MFMA_CONV and IGEMM_CONV and x.is_cuda and x.dtype == torch.bfloat16:
      from lsi.nnutils import _hip_conv  # pylint: disable=g-import-not-at-top
      cin, cout = self.conv.weight.shape[:2]
      # four parity classes of 2 x 2 taps on the implicit-GEMM kernel
      if (_hip_conv.convt_supported(x, cin, cout, 4, 2) and
          x.shape[0] * x.shape[2] * x.shape[3] >= IGEMM_MIN_PIXELS):
        st = _stats_bn(self.bn, 'relu', x.shape[0], cout)
        y = _hip_conv.conv_transpose2d(x, self.conv.weight, 2, 1, st)
        return _bn_relu(self.bn, y, bool(st))
    return _bn_relu(self.bn, self.conv(x))


class SlimFC(nn.Module):
  """slim.fully_connected + BN + ReLU (the `fc` stack, nets.py:66-67, 290-291)."""

  def __init__(self, cin, cout):
    super().__init__()
    self.fc = nn.Linear(cin, cout, bias=False)
    nn.init.xavier_uniform_(self.fc.weight)
    self.beta = nn.Parameter(torch.zeros(cout))
    self.register_buffer('gamma', torch.ones(cout))
    # (TF creates and checkpoints the moving statistics; never updated: the
    # reference runs no UPDATE_OPS, train_utils.py:107-117)
    self.register_buffer('moving_mean', torch.zeros(cout))
    self.register_buffer('moving_variance', torch.ones(cout))
    self.eps = 1e-3
    self.is_training = True

  def forward(self, x):
    x = self.fc(x)
    if self.is_training:
      return _per_group(
          lambda c: F.relu(F.batch_norm(c, None, None, self.gamma, self.beta,
                                        True, 0.0, self.eps)), x)
    return F.relu(F.batch_norm(x, self.moving_mean, self.moving_variance,
                               self.gamma, self.beta, False, 0.0, self.eps))


def own_kernel_param_layouts(module):
  """After module.to(memory_format=torch.channels_last): the 3 x 3 layers over 32
  channels (`upcnv1b`, the `pred_l` heads) go back to contiguous parameters --
  their kernels (csrc/lsi_conv.hip, lsi_conv_wgrad.hip) read the fp32 parameter
  itself and return its gradient contiguously; with channels-last strides every
  call made a contiguous copy first and autograd cloned every gradient into the
  parameter's layout (about 40 small kernels per 4-layer step).  The
  implicit-GEMM layers take either layout (_hip_conv._pack_layout)."""
  for m in module.modules():
    if (isinstance(m, SlimConv2d) and m.k == 3 and m.stride == 1 and
        m.conv.weight.shape[1] == 32 and
        (m.conv.weight.shape[0] in (16, 32) or m.bn is None)):
      with torch.no_grad():
        m.conv.weight.data = m.conv.weight.data.contiguous()
  return module


def set_is_training(module, is_training):
  """slim's `is_training` switch for every batch norm below `module`."""
  for m in module.modules():
    if isinstance(m, (SlimBatchNorm, SlimFC)):
      m.is_training = bool(is_training)
  return module


def _nhwc_to_nchw(x):
  return x.permute(0, 3, 1, 2)


def _nchw_to_nhwc(x):
  return x.permute(0, 2, 3, 1)


_ENC = [('cnv1', 32, 7, 2), ('cnv1b', 32, 7, 1), ('cnv2', 64, 5, 2),
        ('cnv2b', 64, 5, 1), ('cnv3', 128, 3, 2), ('cnv3b', 128, 3, 1),
        ('cnv4', 256, 3, 2), ('cnv4b', 256, 3, 1), ('cnv5', 512, 3, 2),
        ('cnv5b', 512, 3, 1), ('cnv6', 512, 3, 2), ('cnv6b', 512, 3, 1),
        ('cnv7', 512, 3, 2), ('cnv7b', 512, 3, 1)]


class _Encoder14(nn.Module):
  """The 14-conv stride-2 encoder shared by encoder_simple and the U-Net."""

  def __init__(self, cin=3):
    super().__init__()
    for name, cout, k, s in _ENC:
      setattr(self, name, SlimConv2d(cin, cout, k, s))
      cin = cout

  def forward(self, x):
    feats = {}
    for name, _, _, _ in _ENC:
      x = getattr(self, name)(x)
      feats[name] = x
    return feats


class EncoderSimple(nn.Module):
  """encoder_simple (reference nets.py:29-70): B x H x W x C -> B x nz."""

  def __init__(self, nz=1000, in_hw=(256, 256), in_channels=3):
    super().__init__()
    self.encoder = _Encoder14(in_channels)
    flat = 512 * (in_hw[0] // 128) * (in_hw[1] // 128)
    self.fc = nn.Sequential(SlimFC(flat, 2 * nz), SlimFC(2 * nz, nz),
                            SlimFC(nz, nz))

  def forward(self, inp_img):
    feats = self.encoder(_nhwc_to_nchw(inp_img))
    flat = _nchw_to_nhwc(feats['cnv7b']).reshape(inp_img.shape[0], -1)
    return self.fc(flat), feats


class DecoderSimple(nn.Module):
  """decoder_simple (reference nets.py:73-114): nconv up-convolution stages
  upcnv{nc} (4x4 transposed, stride 2) [+ skip concat] + upcnv{nc}b (3x3).

  skip_channels: channel counts of the skip_feat list the caller will pass
  (skip_feat[-nc + 1] is concatenated at stage nc > 1)."""

  def __init__(self, cin, nconv=7, skip_channels=None):
    super().__init__()
    n_filters = [32, 64, 128, 256] + [512] * max(nconv - 4, 0)
    self.nconv = nconv
    for nc in range(nconv, 0, -1):
      n_filt = n_filters[nc - 1]
      setattr(self, 'upcnv%d' % nc, SlimConvTranspose2d(cin, n_filt))
      cin_b = n_filt
      if nc > 1 and skip_channels is not None:
        cin_b += skip_channels[-nc + 1]
      setattr(self, 'upcnv%db' % nc, SlimConv2d(cin_b, n_filt, 3, 1))
      cin = n_filt
    self.out_channels = cin

  def forward(self, feat, skip_feat=None):
    if feat.dim() == 2:
      feat = feat[:, :, None, None]
    for nc in range(self.nconv, 0, -1):
      feat = getattr(self, 'upcnv%d' % nc)(feat)
      if nc > 1 and skip_feat is not None:
        feat = getattr(self, 'upcnv%db' % nc)(feat, skip_feat[-nc + 1])
      else:
        feat = getattr(self, 'upcnv%db' % nc)(feat)
    return feat


# The per-layer decoders of the LDI heads on one stream each (PixelwisePredictor):
# off by default; the Trainer switches it on for eager single-process runs (inside
# a captured HIP graph the fork / join edges cost more than the overlap wins).
HEAD_STREAMS = {'on': os.environ.get('LSI_HEAD_STREAMS', '') == '1'}
_HEAD_STREAM_POOL = {}


def enable_head_streams(on=True):
  old = HEAD_STREAMS['on']
  HEAD_STREAMS['on'] = bool(on)
  return old


class PixelwisePredictor(nn.Module):
  """pixelwise_predictor (reference nets.py:117-161): per layer an own
  DecoderSimple (n_layerwise_steps stages) and a 3x3 sigmoid head with bias.
  Returns L x B x H x W x nc (a planar-strided view: no NHWC copy)."""

  def __init__(self, cin, nc=3, n_layers=1, n_layerwise_steps=0,
               skip_channels=None):
    super().__init__()
    self.n_layers = n_layers
    self.decoders = nn.ModuleList()
    self.preds = nn.ModuleList()
    for _ in range(n_layers):
      dec = DecoderSimple(cin, nconv=n_layerwise_steps,
                          skip_channels=skip_channels)
      self.decoders.append(dec)
      self.preds.append(SlimConv2d(dec.out_channels, nc, 3, 1, batch_norm=False,
                                   activation='sigmoid'))

  def forward(self, feat, skip_feat=None):
    preds = []
    if HEAD_STREAMS['on'] and feat.is_cuda and len(self.decoders) > 1:
      # The L per-layer decoders are independent chains (reference nets.py:131-158:
      # one decoder_simple per layer): decoder l > 0 runs on its own stream, forked
      # behind the features and joined before the stack.  autograd runs a node's
      # backward on its forward's stream and synchronises across streams itself,
      # so the backward chains overlap the same way.  Every per-stream resource of
      # the kernels (batch-norm and weight-gradient workspaces) is keyed by stream.
      dev = feat.device
      main = torch.cuda.current_stream(dev)
      pool = _HEAD_STREAM_POOL.setdefault(dev.index, [])
      while len(pool) < len(self.decoders) - 1:
        pool.append(torch.cuda.Stream(device=dev))
      for i, (dec, head) in enumerate(zip(self.decoders, self.preds)):
        if i == 0:
          preds.append(head(dec(feat, skip_feat)))
          continue
        side = pool[i - 1]
        side.wait_stream(main)
        with torch.cuda.stream(side):
          preds.append(head(dec(feat, skip_feat)))
      for side in pool[:len(self.decoders) - 1]:
        main.wait_stream(side)
    else:
      for dec, head in zip(self.decoders, self.preds):
        preds.append(head(dec(feat, skip_feat)))
    stacked = torch.stack(preds, dim=0)        # L x B x nc x H x W
    return stacked.permute(0, 1, 3, 4, 2)      # L x B x H x W x nc (view)


class LdiPredictor(nn.Module):
  """ldi_predictor (reference nets.py:164-208): [textures, masks, disps]."""

  def __init__(self, cin, n_layers=1, n_layerwise_steps=0, skip_channels=None,
               pred_masks=False):
    super().__init__()
    self.pred_masks = pred_masks
    nc = 3 + 1 + (1 if pred_masks else 0)
    self.pixelwise_pred = PixelwisePredictor(
        cin, nc=nc, n_layers=n_layers, n_layerwise_steps=n_layerwise_steps,
        skip_channels=skip_channels)
    # per-channel factors of the RGBD prediction (1, 1, 1, disp_scale); not a
    # parameter, not in the checkpoint
    self.register_buffer('_scale4', torch.ones(4), persistent=False)
    self._scale4_value = 1.0

  def forward(self, feat, skip_feat=None, disp_scale=None):
    """disp_scale (optional): the disparities come back multiplied by it, as
    float32 (ldi_enc_dec.py:203-205 scales the network's disparities by
    max_disp).  Without masks the scaling is ONE multiply of the whole
    L x B x H x W x 4 prediction, so textures and disparities stay views of one
    buffer of RGBD pixels -- the layout the renderer reads with one 16-byte load
    per pixel (LSI_PACKED_RGBD)."""
    pred = self.pixelwise_pred(feat, skip_feat)
    if self.pred_masks:
      tex, masks, disps = pred[..., 0:3], pred[..., 3:4], pred[..., 4:5]
      # the reference applies the sigmoid a second time here (nets.py:202)
      masks = nn_helpers.enforce_bg_occupied(torch.sigmoid(masks))
      if disp_scale is not None:
        tex, masks, disps = tex.float(), masks.float(), disps.float() * disp_scale
    else:
      if disp_scale is not None:
        # (the factors live on the device: no host copy, nothing a captured HIP
        # graph could not hold)
        if float(self._scale4_value) != float(disp_scale):
          self._scale4.copy_(torch.tensor([1.0, 1.0, 1.0, float(disp_scale)]))
          self._scale4_value = float(disp_scale)
        pred = pred.float() * self._scale4
      tex, disps = pred[..., 0:3], pred[..., 3:4]
      masks = None   # all ones (nets.py:204); None = ones for the renderer
    return [tex, masks, disps]


class EncoderDecoderUnet(nn.Module):
  """encoder_decoder_unet (reference nets.py:244-348).

  forward(inp_img B x H x W x C) -> (feat, feat_dec, skip_feat, end_points):
  feat_dec = feats_dec[-1 - nl_diff_enc_dec] (NCHW), skip_feat =
  [cnv6b, cnv5b, cnv4b, cnv3b, cnv2b, cnv1b] (NCHW), feat = bottleneck `fc`
  features (None unless with_fc).  H and W must be multiples of 128.
  Decoder stages below the returned one are not built -- unless
  `with_dead_decoder`: the reference creates (and checkpoints) upcnv / icnv down
  to stage 1 and the `fc` stack although nothing trains them when
  nl_diff_enc_dec > 0 (nets.py:289-291, 328-345); with the two switches the
  module holds exactly the reference's variable list (frozen, never executed
  below the returned stage), for TF-checkpoint import / export."""

  _DEC = [('7', 512, 'cnv6b'), ('6', 512, 'cnv5b'), ('5', 256, 'cnv4b'),
          ('4', 128, 'cnv3b'), ('3', 64, 'cnv2b'), ('2', 32, 'cnv1b'),
          ('1', 32, None)]

  def __init__(self, nz=1000, nl_diff_enc_dec=0, in_channels=3, with_fc=False,
               in_hw=None, with_dead_decoder=False):
    super().__init__()
    self.encoder = _Encoder14(in_channels)
    self.nl_diff_enc_dec = nl_diff_enc_dec
    self.n_dec = 7 - nl_diff_enc_dec
    skip_c = {'cnv6b': 512, 'cnv5b': 512, 'cnv4b': 256, 'cnv3b': 128,
              'cnv2b': 64, 'cnv1b': 32}
    cin = 512
    for i, (tag, cout, skip) in enumerate(self._DEC):
      if i >= self.n_dec and not with_dead_decoder:
        break
      up = SlimConvTranspose2d(cin, cout)
      ic = SlimConv2d(cout + (skip_c[skip] if skip else 0), cout, 3, 1)
      if i >= self.n_dec:  # never executed: keep it out of the optimiser / DDP
        up.requires_grad_(False)
        ic.requires_grad_(False)
      setattr(self, 'upcnv' + tag, up)
      setattr(self, 'icnv' + tag, ic)
      cin = cout
      if i == self.n_dec - 1:
        self.out_channels = cout
    self.skip_channels = [512, 512, 256, 128, 64, 32]
    self.fc = None
    if with_fc:
      flat = 512 * (in_hw[0] // 128) * (in_hw[1] // 128)
      self.fc = nn.Sequential(SlimFC(flat, 2 * nz), SlimFC(2 * nz, nz),
                              SlimFC(nz, nz))
      # The U-Net's callers never use `feat` (ldi_enc_dec.py:196-205 takes
      # feat_dec and skip_feat): in the reference's graph the stack is created
      # and checkpointed but no gradient reaches it.  Frozen here -- out of the
      # optimiser and of DDP's reducer -- and executed (without autograd) only
      # when a caller asks for it with `want_feat`.
      self.fc.requires_grad_(False)
    self.want_feat = False

  def forward(self, inp_img):
    if inp_img.shape[1] % 128 or inp_img.shape[2] % 128:
      raise ValueError('encoder_decoder_unet needs H and W divisible by 128')
    feats = self.encoder(_nhwc_to_nchw(inp_img))
    feat = None
    if self.fc is not None and self.want_feat:
      with torch.no_grad():
        feat = self.fc(_nchw_to_nhwc(feats['cnv7b']).reshape(inp_img.shape[0], -1))
    x = feats['cnv7b']
    for tag, _, skip in self._DEC[:self.n_dec]:
      x = getattr(self, 'upcnv' + tag)(x)
      if skip is not None:
        x = getattr(self, 'icnv' + tag)(x, feats[skip])
      else:
        x = getattr(self, 'icnv' + tag)(x)
      feats['icnv' + tag] = x
    skip_feat = [feats[k] for k in ('cnv6b', 'cnv5b', 'cnv4b', 'cnv3b', 'cnv2b',
                                    'cnv1b')]
    return feat, x, skip_feat, feats


class EncoderDecoderSimple(nn.Module):
  """encoder_decoder_simple (reference nets.py:211-241): FC bottleneck, then
  nupconv - nl_diff_enc_dec up-convolutions from a 1x1 map."""

  def __init__(self, nz=1000, nupconv=8, nl_diff_enc_dec=0, in_hw=(256, 256),
               in_channels=3):
    super().__init__()
    self.encoder = EncoderSimple(nz=nz, in_hw=in_hw, in_channels=in_channels)
    self.decoder = DecoderSimple(nz, nconv=nupconv - nl_diff_enc_dec)
    self.out_channels = self.decoder.out_channels
    self.skip_channels = None

  def forward(self, inp_img):
    feat, enc_int = self.encoder(inp_img)
    feat_dec = self.decoder(feat)
    return feat, feat_dec, None, enc_int


# Factory functions with the reference's names and keyword arguments.  TF's
# `reuse=True` (second call shares weights) is expressed in torch by calling
# the same module instance again.
def encoder_simple(nz=1000, is_training=True, **kw):
  return set_is_training(EncoderSimple(nz=nz, **kw), is_training)


def decoder_simple(cin, nconv=7, is_training=True, skip_channels=None):
  return set_is_training(DecoderSimple(cin, nconv, skip_channels), is_training)


def pixelwise_predictor(cin, nc=3, n_layers=1, n_layerwise_steps=0,
                        skip_channels=None, is_training=True):
  return set_is_training(
      PixelwisePredictor(cin, nc, n_layers, n_layerwise_steps, skip_channels),
      is_training)


def ldi_predictor(cin, n_layers=1, n_layerwise_steps=0, skip_channels=None,
                  pred_masks=False, is_training=True):
  return set_is_training(
      LdiPredictor(cin, n_layers, n_layerwise_steps, skip_channels, pred_masks),
      is_training)


def encoder_decoder_simple(nz=1000, nupconv=8, is_training=True,
                           nl_diff_enc_dec=0, **kw):
  return set_is_training(
      EncoderDecoderSimple(nz, nupconv, nl_diff_enc_dec, **kw), is_training)


def encoder_decoder_unet(nz=1000, is_training=True, nl_diff_enc_dec=0, **kw):
  return set_is_training(
      EncoderDecoderUnet(nz, nl_diff_enc_dec, **kw), is_training)


def count_parameters(module):
  return sum(p.numel() for p in module.parameters() if p.requires_grad)
