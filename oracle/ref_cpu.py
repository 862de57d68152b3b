"""ctypes loader for the plain-C oracle `oracle/lsi_ref_cpu.c`.

TEST INFRASTRUCTURE ONLY (see the header of lsi_ref_cpu.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liblsi_ref_cpu.so')
_lib = None


def build(force=False):
  src = os.path.join(_HERE, 'lsi_ref_cpu.c')
  if force or not os.path.exists(_SO) or (
      os.path.getmtime(_SO) < os.path.getmtime(src)):
    subprocess.check_call(['make', '-s', '-C', _HERE, 'liblsi_ref_cpu.so'])
  return _SO


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = ctypes.CDLL(_SO)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int32)
    _lib.lsi_ref_forward_splat.restype = ctypes.c_int
    _lib.lsi_ref_forward_splat.argtypes = (
        [fp, fp, fp, fp] + [ctypes.c_int] * 6 + [ctypes.c_double] * 4 +
        [ctypes.c_int, fp, fp, fp, ip, fp, ctypes.c_int])
    _lib.lsi_ref_forward_splat_ws.restype = ctypes.c_int
    _lib.lsi_ref_forward_splat_ws.argtypes = (
        _lib.lsi_ref_forward_splat.argtypes + [fp, ctypes.c_size_t])
    _lib.lsi_ref_canvas_floats.restype = ctypes.c_size_t
    _lib.lsi_ref_canvas_floats.argtypes = [ctypes.c_int] * 6
    _lib.lsi_ref_num_threads.restype = ctypes.c_int
  return _lib


def num_threads():
  return int(lib().lsi_ref_num_threads())


def _p(a, ty=ctypes.c_float):
  if a is None:
    return ctypes.cast(None, ctypes.POINTER(ty))
  return a.ctypes.data_as(ctypes.POINTER(ty))


def forward_splat(tex, mask, disp, mat, trg_downsampling=1, bg_layer_disp=0,
                  max_disp=1, zbuf_scale=10, compose_layers=True,
                  want_disp=True, debug=False, nthreads=0):
  """Same contract as lsi_oracle.forward_splat (ldi.py:71-182), fused C."""
  tex = np.ascontiguousarray(tex, np.float32)
  disp = np.ascontiguousarray(disp, np.float32)
  mat = np.ascontiguousarray(mat, np.float32)
  if mask is not None:
    mask = np.ascontiguousarray(mask, np.float32)
  nl, b, h, w, _ = tex.shape
  ht, wt = h * trg_downsampling, w * trg_downsampling
  assert ht == int(ht) and wt == int(wt)
  ht, wt = int(ht), int(wt)
  no = 1 if compose_layers else nl
  img = np.empty((no, b, ht, wt, 3), np.float32)
  wts = np.empty((no, b, ht, wt, 1), np.float32)
  dsp = np.empty((no, b, ht, wt, 1), np.float32) if want_disp else None
  idx4 = np.empty((nl, b, h * w, 4), np.int32) if debug else None
  upd4 = np.empty((nl, b, h * w, 4), np.float32) if debug else None
  rc = lib().lsi_ref_forward_splat(
      _p(tex), _p(mask), _p(disp), _p(mat), nl, b, h, w, ht, wt,
      float(trg_downsampling), float(bg_layer_disp), float(max_disp),
      float(zbuf_scale), int(bool(compose_layers)), _p(img), _p(wts), _p(dsp),
      _p(idx4, ctypes.c_int32), _p(upd4), int(nthreads))
  if rc != 0:
    raise RuntimeError('lsi_ref_forward_splat failed: %d' % rc)
  out = {'img': img, 'wts': wts, 'disp': dsp}
  if debug:
    out.update(idx4=idx4, upd4=upd4)
  return out


class Context(object):
  """Pre-allocated outputs and canvas scratch for repeated compose-mode calls
  of one shape (bench.py's CPU baseline: no allocation or first-touch page
  faults inside the timed loop)."""

  def __init__(self, nl, b, h, w, ht, wt, nthreads=0):
    self.shape = (nl, b, h, w, ht, wt)
    self.nthreads = int(nthreads)
    n = int(lib().lsi_ref_canvas_floats(nl, b, h, ht, wt, self.nthreads))
    self.canv = np.zeros((n,), np.float32)   # touched once here
    self.img = np.zeros((1, b, ht, wt, 3), np.float32)
    self.wts = np.zeros((1, b, ht, wt, 1), np.float32)

  def forward_splat(self, tex, mask, disp, mat, trg_downsampling, bg_layer_disp,
                    max_disp, zbuf_scale):
    nl, b, h, w, ht, wt = self.shape
    assert tex.shape == (nl, b, h, w, 3) and tex.dtype == np.float32
    rc = lib().lsi_ref_forward_splat_ws(
        _p(tex), _p(mask), _p(disp), _p(mat), nl, b, h, w, ht, wt,
        float(trg_downsampling), float(bg_layer_disp), float(max_disp),
        float(zbuf_scale), 1, _p(self.img), _p(self.wts), _p(None),
        _p(None, ctypes.c_int32), _p(None), self.nthreads, _p(self.canv),
        self.canv.size)
    if rc != 0:
      raise RuntimeError('lsi_ref_forward_splat_ws failed: %d' % rc)
    return {'img': self.img, 'wts': self.wts}
