"""NumPy stand-in for the tf.contrib.slim layers the reference's networks use.

TEST INFRASTRUCTURE ONLY (oracle tooling; see tf1_numpy_shim.py).  It lets
`oracle/make_goldens.py` execute the reference's UNCHANGED
`lsi/nnutils/nets.py` (encoder_decoder_unet, ldi_predictor,
encoder_decoder_simple; nets.py:29-348) eagerly and record

  * the variable list the reference creates -- every `variable_scope/name` with
    its shape, including the variables that are created but never trained (the
    `fc` stack on the U-Net bottleneck, `upcnv3 .. icnv1` when
    n_layerwise_steps = 3), and
  * activations of seeded weights, stage by stage,

which is what pins the PyTorch/MIOpen network of the build (layer order, skip
concatenation, channel counts, TF `SAME` padding, slim batch norm) and the
TF-checkpoint name map (lsi/nnutils/tf_checkpoint.py).

Published semantics restated here (TF 1.4 / slim, each stated once):
  * `slim.conv2d`: NHWC correlation, filter [kh, kw, in, out], padding 'SAME'
    (out = ceil(in / stride); total padding max((out-1)*stride + k - in, 0),
    the smaller half BEFORE), no bias when a normalizer_fn is given, then the
    normaliser, then the activation;
  * `slim.conv2d_transpose`: the adjoint of that correlation with filter
    [kh, kw, out, in] (no flip), 'SAME': out = in * stride;
  * `slim.batch_norm` with slim's defaults center=True, scale=False,
    epsilon=0.001, decay=0.999: is_training=True normalises with the batch
    moments over N, H, W (biased variance) and adds beta; the moving statistics
    are variables (created, initial 0 / 1) updated only through UPDATE_OPS;
  * `slim.fully_connected`: x @ W [in, out] (+ biases without a normaliser);
  * `slim.stack(x, layer, [a, b, c], scope='fc')` creates scopes fc/fc_1 ..;
  * `slim.arg_scope` supplies keyword defaults to the listed layer functions;
  * outputs are collected under the alias = the layer's full variable scope.
Arithmetic: float64 accumulation, results rounded to float32 per layer (TF's
kernels are fp32 with their own summation orders: tolerance level).

Weights are not stored in the fixtures (37 M parameters): `seeded_value(name,
shape)` is a pure function of the TF variable name, called here when the
reference creates a variable and by the tests to rebuild the same values.
"""
import contextlib
import sys
import types
import zlib

import numpy as np

import tf1_numpy_shim as tf

Tensor = tf.Tensor

# ---------------------------------------------------------------------------
# variables and scopes
# ---------------------------------------------------------------------------
_SCOPE = []          # stack of variable-scope names
VARIABLES = {}       # full name -> float32 array   (creation order kept)
COLLECTIONS = {}     # collection name -> [(alias, Tensor)]


def reset():
  del _SCOPE[:]
  VARIABLES.clear()
  COLLECTIONS.clear()


def seeded_value(name, shape):
  """Deterministic value of the variable `name`: a pure function of the name
  (CRC32 seeds a RandomState) and the shape.  Weights: uniform with the xavier
  bound of the fan-in / fan-out (activations stay O(1) through 30 layers);
  beta / biases: small non-zero offsets (so that a wrong beta placement shows);
  moving statistics keep TF's initial 0 / 1."""
  rs = np.random.RandomState(zlib.crc32(name.encode('utf-8')) & 0x7fffffff)
  leaf = name.rsplit('/', 1)[-1]
  if leaf == 'moving_mean':
    return np.zeros(shape, np.float32)
  if leaf == 'moving_variance':
    return np.ones(shape, np.float32)
  if leaf in ('beta', 'biases'):
    return rs.uniform(-0.1, 0.1, shape).astype(np.float32)
  shape = tuple(shape)
  if len(shape) == 4:      # [kh, kw, a, b]
    rf = shape[0] * shape[1]
    fan_in, fan_out = rf * shape[2], rf * shape[3]
  else:                    # [in, out]
    fan_in, fan_out = shape
  bound = np.sqrt(6.0 / (fan_in + fan_out))
  return rs.uniform(-bound, bound, shape).astype(np.float32)


def _scope_name():
  return '/'.join(_SCOPE)


def _get_variable(leaf, shape):
  name = (_scope_name() + '/' + leaf) if _SCOPE else leaf
  if name not in VARIABLES:
    VARIABLES[name] = seeded_value(name, tuple(int(d) for d in shape))
  assert VARIABLES[name].shape == tuple(shape), (name, VARIABLES[name].shape, shape)
  return VARIABLES[name]


class _VarScope(object):

  def __init__(self, name):
    self.name = name
    self.original_name_scope = name + '/'


@contextlib.contextmanager
def variable_scope(name, reuse=False, **_k):  # pylint: disable=unused-argument
  _SCOPE.append(name)
  try:
    yield _VarScope(_scope_name())
  finally:
    _SCOPE.pop()


# ---------------------------------------------------------------------------
# arg_scope
# ---------------------------------------------------------------------------
_ARG_STACK = []


@contextlib.contextmanager
def arg_scope(fns, **kwargs):
  _ARG_STACK.append((tuple(f.__name__ for f in fns), kwargs))
  try:
    yield
  finally:
    _ARG_STACK.pop()


def _scoped(fn):
  """Layer function whose keyword defaults come from the enclosing arg_scopes
  (inner scopes override outer ones, explicit arguments override both)."""

  def wrapped(*args, **kwargs):
    merged = {}
    for names, kw in _ARG_STACK:
      if fn.__name__ in names:
        merged.update(kw)
    merged.update(kwargs)
    return fn(*args, **merged)

  wrapped.__name__ = fn.__name__
  return wrapped


def _collect(outputs_collections, alias, out):
  if outputs_collections:
    COLLECTIONS.setdefault(outputs_collections, []).append((alias, out))


def convert_collection_to_dict(collection):
  """tensorflow.contrib.layers.python.layers.utils.convert_collection_to_dict:
  alias -> output of every layer collected under `collection`."""
  return dict(COLLECTIONS.get(collection, []))


# ---------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------
def _same_pads(n_in, k, stride):
  n_out = -(-n_in // stride)
  total = max((n_out - 1) * stride + k - n_in, 0)
  return n_out, total // 2, total - total // 2


def _conv_nhwc(x, w, stride):
  """'SAME' correlation of x [N,H,W,Ci] with w [kh,kw,Ci,Co] (float64)."""
  n, h, wd, ci = x.shape
  kh, kw, _, co = w.shape
  ho, pt, pb = _same_pads(h, kh, stride)
  wo, pl, pr = _same_pads(wd, kw, stride)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  out = np.zeros((n, ho, wo, co), np.float64)
  w2 = w.reshape(kh * kw, ci, co)
  for ky in range(kh):
    for kx in range(kw):
      patch = xp[:, ky:ky + (ho - 1) * stride + 1:stride,
                 kx:kx + (wo - 1) * stride + 1:stride, :]
      out += np.tensordot(patch, w2[ky * kw + kx], axes=([3], [0]))
  return out


def _conv_transpose_nhwc(x, w, stride):
  """Adjoint of the 'SAME' stride-s correlation: x [N,H,W,Ci], w
  [kh,kw,Co,Ci] -> [N, H*s, W*s, Co]; y[o*s + k - pad_before] += x[o] w[k]."""
  n, h, wd, ci = x.shape
  kh, kw, co, _ = w.shape
  ho, wo = h * stride, wd * stride
  _, pt, _ = _same_pads(ho, kh, stride)
  _, pl, _ = _same_pads(wo, kw, stride)
  full = np.zeros((n, (h - 1) * stride + kh, (wd - 1) * stride + kw, co), np.float64)
  for ky in range(kh):
    for kx in range(kw):
      contrib = np.tensordot(x, w[ky, kx].T, axes=([3], [0]))  # [N,H,W,Co]
      full[:, ky:ky + (h - 1) * stride + 1:stride,
           kx:kx + (wd - 1) * stride + 1:stride, :] += contrib
  return full[:, pt:pt + ho, pl:pl + wo, :]


def batch_norm(x, is_training=True, center=True, scale=False, epsilon=0.001,
               decay=0.999, scope=None, **_k):  # pylint: disable=unused-argument
  xa = np.asarray(tf._f(x), np.float64)
  c = xa.shape[-1]
  with variable_scope(scope or 'BatchNorm'):
    beta = _get_variable('beta', (c,)) if center else None
    gamma = _get_variable('gamma', (c,)) if scale else None
    mean_v = _get_variable('moving_mean', (c,))
    var_v = _get_variable('moving_variance', (c,))
  axes = tuple(range(xa.ndim - 1))
  if is_training:
    mean = xa.mean(axis=axes)
    var = xa.var(axis=axes)      # biased, as tf.nn.moments
  else:
    mean, var = mean_v.astype(np.float64), var_v.astype(np.float64)
  y = (xa - mean) / np.sqrt(var + epsilon)
  if gamma is not None:
    y = y * gamma
  if beta is not None:
    y = y + beta
  return Tensor(y.astype(np.float32))


def _finish(out, c_out, normalizer_fn, normalizer_params, activation_fn,
            outputs_collections, alias):
  if normalizer_fn is None:
    out = out + _get_variable('biases', (c_out,)).astype(np.float64)
    out = Tensor(out.astype(np.float32))
  else:
    out = normalizer_fn(Tensor(out.astype(np.float32)),
                        **(normalizer_params or {}))
  if activation_fn is not None:
    out = activation_fn(out)
  _collect(outputs_collections, alias, out)
  return out


@_scoped
def conv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME',
           activation_fn=None, normalizer_fn=None, normalizer_params=None,
           weights_regularizer=None, outputs_collections=None, scope=None,
           **_k):  # pylint: disable=unused-argument
  assert padding == 'SAME'
  x = np.asarray(tf._f(inputs), np.float64)
  kh, kw = kernel_size
  with variable_scope(scope):
    alias = _scope_name()
    w = _get_variable('weights', (kh, kw, x.shape[-1], num_outputs))
    out = _conv_nhwc(x, w.astype(np.float64), stride)
    return _finish(out, num_outputs, normalizer_fn, normalizer_params,
                   activation_fn, outputs_collections, alias)


@_scoped
def conv2d_transpose(inputs, num_outputs, kernel_size, stride=1, padding='SAME',
                     activation_fn=None, normalizer_fn=None,
                     normalizer_params=None, weights_regularizer=None,
                     outputs_collections=None, scope=None,
                     **_k):  # pylint: disable=unused-argument
  assert padding == 'SAME'
  x = np.asarray(tf._f(inputs), np.float64)
  kh, kw = kernel_size
  with variable_scope(scope):
    alias = _scope_name()
    w = _get_variable('weights', (kh, kw, num_outputs, x.shape[-1]))
    out = _conv_transpose_nhwc(x, w.astype(np.float64), stride)
    return _finish(out, num_outputs, normalizer_fn, normalizer_params,
                   activation_fn, outputs_collections, alias)


@_scoped
def fully_connected(inputs, num_outputs, activation_fn=None, normalizer_fn=None,
                    normalizer_params=None, weights_regularizer=None,
                    outputs_collections=None, scope=None,
                    **_k):  # pylint: disable=unused-argument
  x = np.asarray(tf._f(inputs), np.float64)
  with variable_scope(scope):
    alias = _scope_name()
    w = _get_variable('weights', (x.shape[-1], num_outputs))
    out = x @ w.astype(np.float64)
    return _finish(out, num_outputs, normalizer_fn, normalizer_params,
                   activation_fn, outputs_collections, alias)


def flatten(inputs, scope=None, **_k):  # pylint: disable=unused-argument
  x = tf._f(inputs)
  return Tensor(x.reshape(x.shape[0], -1))


def stack(inputs, layer, stack_args, scope=None, **kwargs):
  """slim.stack: layer applied repeatedly, scopes <scope>/<scope>_<i>."""
  out = inputs
  with variable_scope(scope):
    for i, a in enumerate(stack_args):
      out = layer(out, a, scope='%s_%d' % (scope, i + 1), **kwargs)
  return out


def l2_regularizer(scale):  # declared by the reference, never added to a loss
  return ('l2', scale)


def sigmoid(x):
  xa = np.asarray(tf._f(x), np.float64)
  return Tensor((1.0 / (1.0 + np.exp(-xa))).astype(np.float32))


def install():
  """tensorflow (+ .contrib.slim, .contrib.layers...utils) -> the shims."""
  me = sys.modules[__name__]
  tf.install()
  tf.variable_scope = variable_scope
  tf.nn.sigmoid = sigmoid
  tf.sigmoid = sigmoid
  if not hasattr(Tensor, 'ndims'):
    tf._Shape.ndims = property(lambda self: len(self))
  contrib = types.ModuleType('tensorflow.contrib')
  contrib.slim = me
  tf.contrib = contrib
  sys.modules['tensorflow.contrib'] = contrib
  sys.modules['tensorflow.contrib.slim'] = me
  chain = ['tensorflow.contrib.layers', 'tensorflow.contrib.layers.python',
           'tensorflow.contrib.layers.python.layers']
  parent = contrib
  for name in chain:
    mod = types.ModuleType(name)
    setattr(parent, name.rsplit('.', 1)[-1], mod)
    sys.modules[name] = mod
    parent = mod
  utils = types.ModuleType('tensorflow.contrib.layers.python.layers.utils')
  utils.convert_collection_to_dict = convert_collection_to_dict
  parent.utils = utils
  sys.modules['tensorflow.contrib.layers.python.layers.utils'] = utils
  return me
