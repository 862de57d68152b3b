"""Eager NumPy stand-in for the TensorFlow-1.4 ops used by the reference's hot path.

TEST INFRASTRUCTURE ONLY (oracle tooling).  Nothing in the product path
(`layered-scene-inference_amd/`) may import this file.

Why it exists
-------------
The reference (`/root/reference`, google/layered-scene-inference) is pure
Python-2 / TensorFlow-1.4 graph code (docs/installation.md:3).  TensorFlow is a
third-party dependency that is absent from `/root/reference`, from this
container and from the GPU box (no network).  The arithmetic of the hot path
therefore lives in two places:

  * the reference's own *composition* of ops (lsi/geometry/*.py,
    lsi/nnutils/helpers.py, lsi/loss/loss.py) -- present, importable, and
  * TensorFlow 1.4.0's kernels for ~45 stock ops (scatter_nd, gather, matmul,
    matrix_inverse, exp, ...) -- absent.

This module restates the *published semantics* of those stock ops on NumPy
float32 arrays so that `oracle/make_goldens.py` can execute the reference's
unchanged source files (read from /root/reference at golden-generation time,
never copied) and record golden input/output vectors under `tests/golden/`.

What is pinned and what is not: the goldens pin OUR restatement against the
reference's op composition, argument order, constants and control flow.  They
do not pin the last-ulp behaviour of TF's own kernels (Eigen `exp`, LU-based
`matrix_inverse`, reduction trees): the reference ships no tests or golden
vectors at that boundary, so that part of parity is
"unpinned" and is covered by stated fp32 tolerances instead.

Reading of TF semantics used here (each is a choice, stated once):
  * all float tensors are float32; python scalars are converted to float32
    before any arithmetic (TF `convert_to_tensor` with a float32 peer);
  * `matmul` accumulates sequentially over k with separately rounded multiply
    and add (no FMA): TF-1.4 CPU wheels are built for SSE4.1 without FMA, and
    Eigen's gebp kernel walks k in order;
  * `cast(float -> int32)` truncates toward zero;
  * `scatter_nd` adds duplicate indices (sequential, update order);
  * `reduce_*` use NumPy's float32 reductions (pairwise sum) -- tolerance level;
  * `x -= y` on a tensor rebinds (tensors are immutable), which is what Python
    does when `__isub__` is not defined.
"""
import builtins
import contextlib
import sys
import types

import numpy as np

float32 = np.float32
int32 = np.int32
int64 = np.int64
bool_ = np.bool_

# Optional recorder: when set to a list, every scatter_nd call appends
# (indices[int32 N], updates[float32 N], out_len) -- used to capture the
# reference's projected pixel indices for the bit-exact parity fixtures.
SCATTER_LOG = None


class _Shape(list):
  """TensorShape stand-in: a list of python ints with as_list()."""

  def as_list(self):
    return list(self)


class Tensor(object):
  """Immutable eager tensor wrapping a NumPy array."""
  __array_priority__ = 1000

  def __init__(self, a):
    if isinstance(a, Tensor):
      a = a.a
    self.a = np.asarray(a)

  # -- shape / dtype -------------------------------------------------------
  def get_shape(self):
    return _Shape(int(d) for d in self.a.shape)

  @property
  def shape(self):
    return self.get_shape()

  @property
  def dtype(self):
    return self.a.dtype

  def __len__(self):
    return self.a.shape[0]

  def __iter__(self):
    for i in builtins.range(self.a.shape[0]):
      yield Tensor(self.a[i])

  def __getitem__(self, key):
    if isinstance(key, tuple):
      key = tuple(_raw(k) if isinstance(k, Tensor) else k for k in key)
    elif isinstance(key, Tensor):
      key = _raw(key)
    return Tensor(self.a[key])

  def __repr__(self):
    return 'Tensor(%r)' % (self.a,)

  def __index__(self):
    return int(self.a)

  def __int__(self):
    return int(self.a)

  def __float__(self):
    return float(self.a)

  # -- arithmetic (never in place) ---------------------------------------------
  def __add__(self, o):
    return Tensor(_bin(np.add, self, o))

  def __radd__(self, o):
    return Tensor(_bin(np.add, o, self))

  def __sub__(self, o):
    return Tensor(_bin(np.subtract, self, o))

  def __rsub__(self, o):
    return Tensor(_bin(np.subtract, o, self))

  def __mul__(self, o):
    return Tensor(_bin(np.multiply, self, o))

  def __rmul__(self, o):
    return Tensor(_bin(np.multiply, o, self))

  def __truediv__(self, o):
    return Tensor(_bin(np.true_divide, self, o))

  def __rtruediv__(self, o):
    return Tensor(_bin(np.true_divide, o, self))

  __div__ = __truediv__
  __rdiv__ = __rtruediv__

  def __neg__(self):
    return Tensor(-self.a)


def _raw(x):
  return x.a if isinstance(x, Tensor) else x


def _peer(x, dtype):
  """Convert a python scalar / list to an array with the peer tensor's dtype."""
  if isinstance(x, Tensor):
    return x.a
  if isinstance(x, np.ndarray):
    return x
  return np.asarray(x, dtype=dtype)


def _bin(fn, a, b):
  if isinstance(a, Tensor):
    dt = a.a.dtype
  elif isinstance(b, Tensor):
    dt = b.a.dtype
  else:
    dt = np.float32
  ra, rb = _peer(a, dt), _peer(b, dt)
  if fn is np.true_divide and np.issubdtype(ra.dtype, np.integer):
    ra = ra.astype(np.float64)
  with np.errstate(all='ignore'):
    return fn(ra, rb)


def _f(x):
  """To a float32 array (python floats become float32 first, like TF)."""
  if isinstance(x, Tensor):
    return x.a
  return np.asarray(x, dtype=np.float32)


def _dtype(d):
  if d is None:
    return np.float32
  if isinstance(d, str):
    return np.dtype(d).type
  return np.dtype(d).type


def _dims(shape):
  """Shape argument -> tuple of ints (integral floats allowed: ldi.py:113-125)."""
  if isinstance(shape, Tensor):
    shape = shape.a.tolist()
  out = []
  for d in shape:
    d = _raw(d)
    fd = float(d)
    if fd != int(fd):
      raise ValueError('non-integral dimension %r' % (d,))
    out.append(int(fd))
  return tuple(out)


# -- graph scaffolding ---------------------------------------------------------
@contextlib.contextmanager
def name_scope(*_a, **_k):
  yield


@contextlib.contextmanager
def control_dependencies(*_a, **_k):
  yield


@contextlib.contextmanager
def variable_scope(*_a, **_k):
  yield


def Print(x, *_a, **_k):
  return x


def stop_gradient(x):
  return Tensor(x)


def assert_equal(a, b, *_a, **_k):
  if not np.array_equal(_raw(a), _raw(b)):
    raise AssertionError('tf.assert_equal failed: %r vs %r' % (a, b))
  return None


def convert_to_tensor(x, dtype=None):
  if isinstance(x, Tensor):
    return x
  a = np.asarray(x)
  if dtype is not None:
    a = a.astype(_dtype(dtype))
  elif a.dtype == np.float64:
    a = a.astype(np.float32)
  elif a.dtype == np.int64:
    a = a.astype(np.int32)
  return Tensor(a)


def constant(value, dtype=None, shape=None):
  a = np.asarray(value, dtype=_dtype(dtype) if dtype is not None else None)
  if dtype is None and a.dtype == np.float64:
    a = a.astype(np.float32)
  if shape is not None:
    a = np.broadcast_to(a, _dims(shape)).copy() if a.ndim == 0 else a.reshape(
        _dims(shape))
  return Tensor(a)


# -- creation / shape ops --------------------------------------------------------
def zeros(shape, dtype='float32'):
  return Tensor(np.zeros(_dims(shape), dtype=_dtype(dtype)))


def ones(shape=None, dtype='float32'):
  return Tensor(np.ones(_dims(shape), dtype=_dtype(dtype)))


def shape(x, out_type=np.int32):
  return Tensor(np.asarray(_raw(x).shape, dtype=_dtype(out_type)))


def range(*args):  # pylint: disable=redefined-builtin
  return Tensor(np.arange(*[int(_raw(a)) for a in args], dtype=np.int32))


def reshape(x, shp):
  if isinstance(shp, Tensor):
    shp = shp.a.tolist()
  shp = tuple(int(_raw(d)) for d in shp)
  return Tensor(np.reshape(_raw(x), shp))


def transpose(x, perm=None):
  return Tensor(np.transpose(_raw(x), perm))


def expand_dims(x, axis):
  return Tensor(np.expand_dims(_raw(x), axis))


def tile(x, multiples):
  return Tensor(np.tile(_raw(x), tuple(int(m) for m in multiples)))


def concat(values, axis):
  return Tensor(np.concatenate([_f(v) for v in values], axis=axis))


def stack(values, axis=0):
  return Tensor(np.stack([_raw(convert_to_tensor(v)) for v in values],
                         axis=axis))


def split(value, num_or_size_splits, axis=0):
  a = _raw(value)
  if isinstance(num_or_size_splits, int):
    parts = np.split(a, num_or_size_splits, axis=axis)
  else:
    idx = np.cumsum(list(num_or_size_splits))[:-1]
    parts = np.split(a, idx, axis=axis)
  return [Tensor(p) for p in parts]


def cast(x, dtype):
  dt = _dtype(dtype)
  a = _raw(x)
  a = np.asarray(a)
  if np.issubdtype(dt, np.integer) and np.issubdtype(a.dtype, np.floating):
    with np.errstate(all='ignore'):
      return Tensor(np.trunc(a).astype(dt))
  return Tensor(a.astype(dt))


# -- elementwise math --------------------------------------------------------------
def add(a, b, name=None):
  return Tensor(_bin(np.add, a, b))


def add_n(values):
  out = _raw(values[0])
  for v in values[1:]:
    out = out + _raw(v)
  return Tensor(out)


def divide(a, b, name=None):
  return Tensor(_bin(np.true_divide, a, b))


def floor(x):
  return Tensor(np.floor(_f(x)))


def abs(x):  # pylint: disable=redefined-builtin
  return Tensor(np.abs(_f(x)))


def exp(x, name=None):
  with np.errstate(all='ignore'):
    return Tensor(np.exp(_f(x)))


def log(x):
  with np.errstate(all='ignore'):
    return Tensor(np.log(_f(x)))


def square(x):
  a = _f(x)
  return Tensor(a * a)


def clip_by_value(x, lo, hi):
  a = _f(x)
  return Tensor(np.minimum(np.maximum(a, _peer(lo, a.dtype)),
                           _peer(hi, a.dtype)))


def equal(a, b):
  return Tensor(_bin(np.equal, a if isinstance(a, Tensor) else Tensor(_f(a)),
                     b))


def greater(a, b):
  return Tensor(_bin(np.greater, a if isinstance(a, Tensor) else Tensor(_f(a)),
                     b))


def less(a, b):
  return Tensor(_bin(np.less, a if isinstance(a, Tensor) else Tensor(_f(a)),
                     b))


# -- linear algebra ----------------------------------------------------------------
def matmul(a, b, name=None):
  """Batched matmul; sequential-k, multiply and add rounded separately."""
  a, b = _f(a), _f(b)
  k_dim = a.shape[-1]
  assert b.shape[-2] == k_dim, (a.shape, b.shape)
  out = a[..., :, 0:1] * b[..., 0:1, :]
  for k in builtins.range(1, k_dim):
    out = out + a[..., :, k:k + 1] * b[..., k:k + 1, :]
  return Tensor(out.astype(np.float32))


def matrix_inverse(x, name=None):
  return Tensor(np.linalg.inv(_f(x)).astype(np.float32))


# -- reductions ----------------------------------------------------------------------
def _reduce(fn, x, axis, keep_dims):
  a = _f(x)
  return Tensor(np.asarray(fn(a, axis=axis, keepdims=bool(keep_dims)),
                           dtype=a.dtype))


def reduce_sum(x, axis=None, keep_dims=False):
  return _reduce(np.sum, x, axis, keep_dims)


def reduce_mean(x, axis=None, keep_dims=False):
  return _reduce(np.mean, x, axis, keep_dims)


def reduce_max(x, axis=None, keep_dims=False):
  return _reduce(np.max, x, axis, keep_dims)


def reduce_min(x, axis=None, keep_dims=False):
  return _reduce(np.min, x, axis, keep_dims)


def cumsum(x, axis=0):
  return Tensor(np.cumsum(_f(x), axis=axis, dtype=np.float32))


def argmax(x, axis=0):
  return Tensor(np.argmax(_raw(x), axis=axis).astype(np.int64))


def one_hot(indices, depth, axis=-1):
  idx = _raw(indices)
  eye = np.arange(depth)
  oh = (idx[..., None] == eye).astype(np.float32)
  if axis != -1:
    oh = np.moveaxis(oh, -1, axis)
  return Tensor(oh)


# -- gather / scatter ------------------------------------------------------------------
def gather(params, indices):
  return Tensor(_raw(params)[_raw(indices)])


def scatter_nd(indices, updates, shp):
  """Dense tensor of `shp` with `updates` added at `indices` (duplicates add)."""
  idx = _raw(indices)
  upd = _raw(updates)
  out_shape = tuple(int(d) for d in _raw(shp).tolist())
  assert idx.ndim == 2 and idx.shape[1] == 1 and len(out_shape) == 1, (
      'shim scatter_nd only covers the rank-1 use at sampling.py:278')
  flat = idx[:, 0]
  if flat.size and (flat.min() < 0 or flat.max() >= out_shape[0]):
    raise IndexError('scatter_nd index out of range')
  out = np.zeros(out_shape, dtype=upd.dtype)
  np.add.at(out, flat, upd)
  if SCATTER_LOG is not None:
    SCATTER_LOG.append((flat.astype(np.int32).copy(), upd.copy(),
                        out_shape[0]))
  return Tensor(out)


class _NN(object):

  @staticmethod
  def relu(x):
    return Tensor(np.maximum(_f(x), np.float32(0)))


nn = _NN()


class _Image(object):
  """tf.image: only what the view-synthesis loss uses (ldi_enc_dec.py:337-340):
  resize_images(..., method=AREA) to an integer fraction of the size, i.e. the
  exact box mean (TF's resize_area averages the source pixels a target pixel
  covers; for integer factors every one of them has weight 1)."""

  class ResizeMethod(object):
    BILINEAR, NEAREST_NEIGHBOR, BICUBIC, AREA = 0, 1, 2, 3

  @staticmethod
  def resize_images(images, size, method=0, align_corners=False):
    a = _f(images)
    ht, wt = [int(v) for v in size]
    b, h, w, c = a.shape
    if method != _Image.ResizeMethod.AREA or h % ht or w % wt:
      raise NotImplementedError('shim: AREA resize by integer factors only')
    fy, fx = h // ht, w // wt
    out = a.reshape(b, ht, fy, wt, fx, c).astype(np.float32)
    acc = np.zeros((b, ht, wt, c), np.float32)
    for dy in builtins.range(fy):      # rows, then columns, in order
      for dx in builtins.range(fx):
        acc = acc + out[:, :, dy, :, dx, :]
    return Tensor((acc * np.float32(1.0 / (fy * fx))).astype(np.float32))


image = _Image()


def install():
  """Registers this module as `tensorflow` plus an `absl.logging` stub."""
  me = sys.modules[__name__]
  sys.modules['tensorflow'] = me
  absl = types.ModuleType('absl')
  absl_logging = types.ModuleType('absl.logging')
  absl_logging.info = lambda *a, **k: None
  absl.logging = absl_logging
  sys.modules.setdefault('absl', absl)
  sys.modules.setdefault('absl.logging', absl_logging)
  return me


def load_reference(ref_root='/root/reference'):
  """Executes the reference's seven hot-path modules (unchanged source, read
  in place) on this shim and returns them as a dict of module objects.

  Only callable where /root/reference exists (this container); never on the
  GPU box.  py2 `range`-returns-a-list semantics are injected per module
  (helpers.py:75-77 mutates a range).
  """
  import os
  install()
  names = [
      ('lsi', None),
      ('lsi.nnutils', None),
      ('lsi.geometry', None),
      ('lsi.loss', None),
      ('lsi.nnutils.helpers', 'lsi/nnutils/helpers.py'),
      ('lsi.geometry.sampling', 'lsi/geometry/sampling.py'),
      ('lsi.geometry.projection', 'lsi/geometry/projection.py'),
      ('lsi.geometry.homography', 'lsi/geometry/homography.py'),
      ('lsi.geometry.layers', 'lsi/geometry/layers.py'),
      ('lsi.geometry.ldi', 'lsi/geometry/ldi.py'),
      ('lsi.loss.loss', 'lsi/loss/loss.py'),
  ]
  saved = {n: sys.modules.get(n) for n, _ in names}
  mods = {}
  try:
    for name, rel in names:
      mod = types.ModuleType(name)
      if rel is None:
        mod.__path__ = []
      else:
        path = os.path.join(ref_root, rel)
        mod.__file__ = path
        mod.__dict__['range'] = lambda *a: list(builtins.range(*a))
      sys.modules[name] = mod
      parent, _, leaf = name.rpartition('.')
      if parent:
        setattr(sys.modules[parent], leaf, mod)
      if rel is not None:
        with open(path, 'r') as f:
          src = f.read()
        exec(compile(src, path, 'exec'), mod.__dict__)  # pylint: disable=exec-used
      mods[name] = mod
  finally:
    # Leave no `lsi` package behind: the product package is also called `lsi`.
    for n, old in saved.items():
      if old is None:
        sys.modules.pop(n, None)
      else:
        sys.modules[n] = old
  return mods
