"""C-ABI boundary checks that need no GPU: the library builds, loads, exports
every symbol include/lsi_hip.h declares, host-only entry points behave, and the
Python mirror refuses to run without a ROCm device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


def declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'lsi_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(lsi_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_are_exported(built_lib):
  handle = ctypes.CDLL(built_lib)
  names = declared_symbols()
  assert len(names) >= 14
  for n in names:
    assert hasattr(handle, n), 'symbol %s declared but not exported' % n


def test_binding_covers_the_header(built_lib):
  from lsi import _C
  assert sorted(_C.SIGNATURES) == declared_symbols()
  lib = _C.lib()
  assert lib.lsi_version() == 100
  assert lib.lsi_strerror(0) == b'ok'
  assert b'workspace' in lib.lsi_strerror(-3)
  assert b'unknown' in lib.lsi_strerror(-99)


def test_desc_struct_matches_header_layout(built_lib):
  from lsi import _C
  # 6 int32 + 13 int64 + 4 float + uint32 + 5 int32 = 24 + 104 + 16 + 24
  assert ctypes.sizeof(_C.LsiSplatDesc) == 176
  assert _C.LsiSplatDesc.adapt.offset == 168
  assert ctypes.sizeof(_C.LsiStreamAdapt) == 32
  assert _C.LsiSplatDesc.tex_sl.offset == 24
  assert _C.LsiSplatDesc.trg_downsampling.offset == 128
  assert _C.LsiSplatDesc.flags.offset == 144


def test_bg_weight_matches_reference_known_answers(built_lib):
  from lsi import _C
  g = golden('known_answers.npz')
  assert abs(_C.bg_weight(1e-3, 0.4, 50) / float(g['bg_wt_kitti']) - 1) < 2e-6
  assert abs(_C.bg_weight(2e-1, 1.0, 50) / float(g['bg_wt_synth']) - 1) < 2e-6
  assert _C.bg_weight(0.0, 1.0, 10) == 0.0


def _desc(_C, L=2, B=2, H=32, W=96, s=0.5, flags=1):
  d = _C.LsiSplatDesc()
  d.L, d.B, d.H, d.W, d.Ht, d.Wt = L, B, H, W, int(H * s), int(W * s)
  d.trg_downsampling, d.max_disp, d.zbuf_scale, d.bg_wt = s, 0.4, 50.0, 1e-11
  d.flags = flags
  return d


def test_rowband_precondition_and_workspace(built_lib):
  from lsi import _C
  lib = _C.lib()
  d = _desc(_C)
  g = golden('fs_kitti_L2_s05.npz')       # rectified stereo: row-band exact
  m = np.ascontiguousarray(g['M'], np.float32)
  assert lib.lsi_rowband_ok(ctypes.byref(d), m.ctypes.data) == 1
  g2 = golden('fs_general_L3_s05.npz')    # 3-D translation: not row-band
  d2 = _desc(_C, L=3, B=2, H=32, W=32)
  m2 = np.ascontiguousarray(g2['M'], np.float32)
  assert lib.lsi_rowband_ok(ctypes.byref(d2), m2.ctypes.data) == 0
  mb = m.copy()
  mb[0, 2, 2] = -1.0                      # normaliser negative: rejected
  assert lib.lsi_rowband_ok(ctypes.byref(d), mb.ctypes.data) == 0
  # one allocation serves every path: max(ATOMIC canvases, STREAM exchange area)
  def stream_need(npass):  # 1-row bands: counters (256-B granules) + 2 rows/band
    return (npass * 2 * 16 * 4 + 255) // 256 * 256 + npass * 2 * 16 * 2 * 48 * 16
  # compose without disparity: one 4-channel canvas per batch element
  assert lib.lsi_splat_workspace_bytes(ctypes.byref(d)) == max(
      2 * 16 * 48 * 4 * 4, stream_need(1))
  d.flags = 1 | 2                         # + disparity: L canvases x 5 channels
  assert lib.lsi_splat_workspace_bytes(ctypes.byref(d)) == max(
      2 * 2 * 16 * 48 * 5 * 4, stream_need(1))
  d.flags = 0                             # independent layers
  assert lib.lsi_splat_workspace_bytes(ctypes.byref(d)) == max(
      2 * 2 * 16 * 48 * 4 * 4, stream_need(2))
  assert lib.lsi_splat_bwd_workspace_bytes(ctypes.byref(d)) == 2 * 2 * 16 * 48 * 16
  bad = _desc(_C, L=0)
  assert lib.lsi_splat_workspace_bytes(ctypes.byref(bad)) == 0
  # a descriptor that names its path asks for that path's need only: the
  # any-pose path keeps 8 disparity ranges per (layer, view), STREAM counters
  # and boundary rows -- not the ATOMIC canvases
  d.flags = 1
  d.path = _C.LSI_PATH_TILE
  assert lib.lsi_splat_workspace_bytes(ctypes.byref(d)) == 2 * 2 * 8 * 8
  d.path = _C.LSI_PATH_STREAM
  assert lib.lsi_splat_workspace_bytes(ctypes.byref(d)) == stream_need(1)


def test_argument_errors_are_reported_before_any_launch(built_lib):
  from lsi import _C
  lib = _C.lib()
  d = _desc(_C)
  null = ctypes.c_void_p(None)
  rc = lib.lsi_splat_fwd(ctypes.byref(d), null, null, null, null, null, null,
                         null, null, 0, null)
  assert rc == -2                          # LSI_ENULL
  bad = _desc(_C, L=-1)
  rc = lib.lsi_splat_fwd(ctypes.byref(bad), null, null, null, null, null, null,
                         null, null, 0, null)
  assert rc == -1                          # LSI_EINVAL
  assert lib.lsi_bilinear_fwd(0, 4, 4, 3, 4, 4, null, null, null, null) == -1
  assert lib.lsi_scatter_add(1, 8, 0, null, null, null, null) == 0


def test_no_cpu_fallback(built_lib):
  from lsi.geometry import ldi, sampling
  tex = torch.rand(1, 1, 8, 8, 3)
  disp = torch.rand(1, 1, 8, 8, 1)
  eye = torch.eye(4).unsqueeze(0)
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    ldi.forward_splat_matrix([tex, None, disp], eye)
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    sampling.bilinear(torch.rand(1, 4, 4, 1), torch.rand(1, 4, 4, 2))
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    sampling.splat(torch.rand(1, 4, 4, 1), torch.rand(1, 4, 4, 2),
                   torch.zeros(1, 4, 4, 1))


def test_missing_library_fails_loudly(built_lib, monkeypatch):
  from lsi import _C
  monkeypatch.setattr(_C, '_lib', None)
  monkeypatch.setattr(_C, 'SO_PATH', '/nonexistent/liblsi_hip.so')
  with pytest.raises(RuntimeError, match='liblsi_hip.so is missing'):
    _C.lib()
