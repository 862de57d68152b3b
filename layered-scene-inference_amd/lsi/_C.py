"""ctypes binding of liblsi_hip.so (C ABI: include/lsi_hip.h).

There is NO fallback: if the shared library is missing, or an op is handed a
tensor that does not live on a ROCm device, this module raises.  PyTorch is used
only for device memory, streams and autograd bookkeeping.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# LSI_HIP_LIB=hooks selects the instrumented build (build.py --hooks) whose
# stream kernel honours the timing-experiment bits of LsiSplatDesc.reserved
# (any other value picks liblsi_hip_<value>.so: experiment builds)
SO_PATH = os.path.join(_PKG, 'liblsi_hip_%s.so' % os.environ['LSI_HIP_LIB'] if
                       os.environ.get('LSI_HIP_LIB') else 'liblsi_hip.so')

LSI_OK = 0
LSI_COMPOSE, LSI_WANT_DISP, LSI_HAS_MASK, LSI_WS_KEEP = 1, 2, 4, 8
LSI_DETERMINISTIC = 16
LSI_PACKED_RGBD = 32
LSI_PATH_AUTO, LSI_PATH_ATOMIC, LSI_PATH_ROWBAND, LSI_PATH_STREAM = 0, 1, 2, 3
LSI_PATH_TILE = 4
PATH_NAMES = {1: 'atomic', 2: 'rowband', 3: 'stream', 4: 'tile'}

_c_f = ctypes.POINTER(ctypes.c_float)
_c_i = ctypes.POINTER(ctypes.c_int32)


class LsiStreamAdapt(ctypes.Structure):
  _fields_ = [('state', ctypes.c_int32), ('pending', ctypes.c_int32),
              ('seq', ctypes.c_uint32), ('calls', ctypes.c_uint32),
              ('ctr_dev', ctypes.c_void_p), ('ctr_host', ctypes.c_void_p)]


class LsiSplatDesc(ctypes.Structure):
  _fields_ = (
      [(n, ctypes.c_int32) for n in ('L', 'B', 'H', 'W', 'Ht', 'Wt')] +
      [(n, ctypes.c_int64) for n in (
          'tex_sl', 'tex_sb', 'tex_sy', 'tex_sx', 'tex_sc',
          'disp_sl', 'disp_sb', 'disp_sy', 'disp_sx',
          'mask_sl', 'mask_sb', 'mask_sy', 'mask_sx')] +
      [(n, ctypes.c_float) for n in (
          'trg_downsampling', 'max_disp', 'zbuf_scale', 'bg_wt')] +
      [('flags', ctypes.c_uint32), ('path', ctypes.c_int32),
       ('tune_rows', ctypes.c_int32), ('tune_threads', ctypes.c_int32),
       ('tune_window', ctypes.c_int32), ('reserved', ctypes.c_int32),
       ('adapt', ctypes.c_void_p)])


class LsiLossDesc(ctypes.Structure):
  _fields_ = (
      [(n, ctypes.c_int32) for n in ('L', 'B', 'H', 'W')] +
      [(n, ctypes.c_int64) for n in (
          'img_sl', 'img_sb', 'img_sy', 'img_sx', 'img_sc',
          'mask_sl', 'mask_sb', 'mask_sy', 'mask_sx',
          'disp_sl', 'disp_sb', 'disp_sy', 'disp_sx',
          'trg_sb', 'trg_sy', 'trg_sx', 'trg_sc')] +
      [(n, ctypes.c_float) for n in ('bg_layer_disp', 'max_disp', 'zbuf_scale')] +
      [('reserved', ctypes.c_int32)])


class LsiConvDesc(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int32) for n in (
      'N', 'H', 'W', 'Cin', 'OH', 'OW', 'Cout', 'KH', 'KW', 'stride', 'pad_t', 'pad_l')]


class LsiPackJob(ctypes.Structure):
  _fields_ = ([('w', ctypes.c_void_p), ('dst', ctypes.c_void_p)] +
              [(n, ctypes.c_int32) for n in ('D0', 'D1', 'khw', 'tr', 'ntaps', 'block0')] +
              [('tap', ctypes.c_int8 * 56)])


class LsiConvIO(ctypes.Structure):
  _fields_ = ([(n, ctypes.c_void_p) for n in ('x', 'x2', 'packed', 'out', 'out2',
                                              'bn_workspace', 'workspace')] +
              [('workspace_bytes', ctypes.c_size_t), ('c1', ctypes.c_int32),
               ('groups', ctypes.c_int32)])


# name -> (restype, argtypes); every symbol include/lsi_hip.h declares.
_I32, _I64, _VP, _SZ = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
_DP = ctypes.POINTER(LsiSplatDesc)
_LP = ctypes.POINTER(LsiLossDesc)
_CP = ctypes.POINTER(LsiConvDesc)
_F32 = ctypes.c_float
SIGNATURES = {
    'lsi_version': (ctypes.c_int, []),
    'lsi_strerror': (ctypes.c_char_p, [ctypes.c_int]),
    'lsi_bg_weight': (ctypes.c_float, [ctypes.c_double] * 3),
    'lsi_rowband_ok': (ctypes.c_int, [_DP, _VP]),
    'lsi_stream_ok': (ctypes.c_int, [_DP, _VP]),
    'lsi_splat_workspace_bytes': (_SZ, [_DP]),
    'lsi_stream_adapt_state': (ctypes.c_int, [ctypes.POINTER(LsiStreamAdapt)]),
    'lsi_splat_fwd': (ctypes.c_int, [_DP] + [_VP] * 8 + [_SZ, _VP]),
    'lsi_splat_bwd_workspace_bytes': (_SZ, [_DP]),
    'lsi_splat_bwd': (ctypes.c_int, [_DP] + [_VP] * 12 + [_SZ, _VP]),
    'lsi_splat_fwd_both': (ctypes.c_int, [_DP] + [_VP] * 9 + [_SZ, _VP]),
    'lsi_splat_bwd_both': (ctypes.c_int, [_DP] + [_VP] * 16 + [_SZ, _VP]),
    'lsi_project_indices': (ctypes.c_int, [_DP] + [_VP] * 6),
    'lsi_splat_generic': (ctypes.c_int, [_I32] * 6 + [_VP] * 4),
    'lsi_splat_generic_bwd': (ctypes.c_int, [_I32] * 6 + [_VP] * 6),
    'lsi_scatter_add': (ctypes.c_int, [_I32, _I64, _I64] + [_VP] * 4),
    'lsi_bilinear_fwd': (ctypes.c_int, [_I32] * 6 + [_VP] * 4),
    'lsi_conv3x3_pred_bwd_workspace_bytes': (_SZ, [_I32] * 3),
    'lsi_conv3x3_wgrad_workspace_bytes': (_SZ, [_I32] * 5),
    'lsi_conv3x3_wgrad': (ctypes.c_int, [_I32] * 5 + [_VP] * 4 + [_SZ, _VP]),
    'lsi_conv3x3_pred_bwd': (ctypes.c_int, [_I32] * 4 + [_VP] * 7 + [_SZ, _VP]),
    'lsi_conv3x3_c32_fwd': (ctypes.c_int, [_I32] * 5 + [_VP] * 3 + [_F32, _VP, _VP]),
    'lsi_projection_matrices': (ctypes.c_int, [_I32] + [_VP] * 4 + [_I32, _VP]),
    'lsi_bilinear_taps': (ctypes.c_int, [_I32] * 6 + [_VP] * 5),
    'lsi_bilinear_bwd': (ctypes.c_int, [_I32] * 6 + [_VP] * 6),
    'lsi_bilinear_taps_bwd': (ctypes.c_int, [_I32] * 6 + [_VP] * 5 + [_VP]),
    'lsi_loss_workspace_bytes': (_SZ, []),
    'lsi_zbuf_comp_loss_fwd': (ctypes.c_int, [_LP] + [_VP] * 6 + [_SZ, _VP]),
    'lsi_zbuf_comp_loss_bwd': (ctypes.c_int, [_LP] + [_VP] * 9),
    'lsi_disp_reg_loss_fwd': (ctypes.c_int, [_I32] * 4 + [_I64] * 4 + [_VP] * 3 +
                              [_SZ, _VP]),
    'lsi_disp_reg_loss_bwd': (ctypes.c_int, [_I32] * 4 + [_I64] * 4 + [_VP] * 4),
    'lsi_view_synth_loss_fwd': (ctypes.c_int, [_I32] * 8 + [_VP] * 2 + [_I64] * 4 +
                                [_VP] * 2 + [_SZ, _VP]),
    'lsi_view_synth_loss_bwd': (ctypes.c_int, [_I32] * 8 + [_VP] * 2 + [_I64] * 4 +
                                [_VP] * 3),
    'lsi_compose_fwd': (ctypes.c_int, [_I32, _I64, _I32] + [_VP] * 3 +
                        [_I32, _F32, _F32, _VP, _VP]),
    'lsi_compose_depth_fwd': (ctypes.c_int, [_I32, _I64] + [_VP] * 2 +
                              [_I32, _F32, _F32, _F32, _VP, _VP]),
    'lsi_conv2d_supported': (ctypes.c_int, [_CP]),
    'lsi_conv2d_packed_bytes': (_SZ, [_CP]),
    'lsi_conv2d_pack': (ctypes.c_int, [_CP, _I32, _VP, _VP, _SZ, _VP]),
    'lsi_conv2d_pack_job': (ctypes.c_int, [_CP, _I32, _VP, _VP, _SZ,
                                           ctypes.POINTER(LsiPackJob), _c_i]),
    'lsi_conv2d_pack_many': (ctypes.c_int, [_VP, _I32, _I32, _VP]),
    'lsi_conv2d_fwd': (ctypes.c_int, [_CP] + [_VP] * 4),
    'lsi_conv2d_bwd_data': (ctypes.c_int, [_CP] + [_VP] * 4),
    'lsi_conv2d_fwd_bnstats': (ctypes.c_int, [_CP] + [_VP] * 4 + [_I32, _VP]),
    'lsi_conv2d_bwd_data_bnstats': (ctypes.c_int, [_CP] + [_VP] * 4 + [_I32, _VP]),
    'lsi_conv2d_fwd_cat': (ctypes.c_int, [_CP, _VP, _VP, _I32, _VP, _VP, _VP, _I32, _VP]),
    'lsi_conv2d_bwd_data_cat': (ctypes.c_int, [_CP, _VP, _VP, _VP, _VP, _I32, _VP]),
    'lsi_conv2d_wgrad_cat': (ctypes.c_int, [_CP, _VP, _VP, _I32, _VP, _VP, _I32, _VP, _SZ, _VP]),
    'lsi_conv2d_workspace_bytes': (_SZ, [_CP, _I32]),
    'lsi_conv2d_run': (ctypes.c_int, [_CP, _I32, ctypes.POINTER(LsiConvIO), _VP]),
    'lsi_conv2d_wgrad_workspace_bytes': (_SZ, [_CP]),
    'lsi_conv2d_wgrad': (ctypes.c_int, [_CP] + [_VP] * 4 + [_SZ, _VP]),
    'lsi_conv2d_first_supported': (ctypes.c_int, [_CP]),
    'lsi_conv2d_first_fwd': (ctypes.c_int, [_CP, _VP, _I32, _VP, _I32, _VP, _VP, _I32, _VP]),
    'lsi_conv2d_first_wgrad_workspace_bytes': (_SZ, [_CP]),
    'lsi_conv2d_first_wgrad': (ctypes.c_int, [_CP, _VP, _I32, _VP, _VP, _I32, _VP, _SZ, _VP]),
    'lsi_bn_workspace_floats': (_SZ, [_I64, _I32, _I32, _I32]),
    'lsi_bn_relu_fwd': (ctypes.c_int, [_VP] * 5 + [_I64, _I32, _I32, _I32, _F32, _I32, _VP]),
    'lsi_bn_relu_bwd': (ctypes.c_int, [_VP] * 7 + [_I64, _I32, _I32, _I32, _I32, _VP]),
    'lsi_bn_relu_norm': (ctypes.c_int, [_VP] * 5 + [_I64, _I32, _I32, _I32, _F32, _I32, _VP]),
    'lsi_bn_stats_discard': (ctypes.c_int, [_VP, _I32, _VP]),
}

_lib = None


def lib():
  """The loaded shared library; raises if it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(SO_PATH):
      raise RuntimeError(
          'liblsi_hip.so is missing (%s). Build it with '
          '`python layered-scene-inference_amd/build.py`; there is no CPU or '
          'PyTorch fallback for the lsi HIP ops.' % SO_PATH)
    handle = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(handle, name)  # AttributeError => ABI mismatch, loudly
      fn.restype = res
      fn.argtypes = args
    _lib = handle
  return _lib


def check(rc, what):
  if rc != LSI_OK:
    msg = lib().lsi_strerror(rc).decode()
    raise RuntimeError('%s failed: %s (code %d)' % (what, msg, rc))


def require_device(*tensors):
  """Every tensor must be fp32 on one ROCm device; returns that device."""
  dev = None
  for t in tensors:
    if t is None:
      continue
    if not t.is_cuda:
      raise RuntimeError(
          'lsi HIP ops need tensors on a ROCm GPU (got device %s); there is no '
          'CPU fallback' % t.device)
    if t.dtype != torch.float32 and t.dtype != torch.int32:
      raise RuntimeError('lsi HIP ops are fp32 (got %s)' % t.dtype)
    if dev is None:
      dev = t.device
    elif t.device != dev:
      raise RuntimeError('tensors on different devices: %s vs %s' %
                         (dev, t.device))
  return dev


def ptr(t):
  """Device address for a c_void_p parameter: a plain int (every entry point has
  its argtypes declared in SIGNATURES, so ctypes converts it itself -- a
  c_void_p object per argument costs half a microsecond of an eager step that
  makes hundreds of these calls)."""
  return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device):
  """The current HIP stream of `device` as an integer.  torch's raw accessor
  where it exists: `torch.cuda.current_stream(device).cuda_stream` builds a
  Stream object per call -- 5 us, 160 times per eager training step
  (profiles/r06/host_profile_L4.txt: 0.8 ms of a 10.6 ms launch side)."""
  if _RAW_STREAM is not None:
    idx = device.index
    if idx is None:
      idx = torch.cuda.current_device()
    return _RAW_STREAM(idx)
  return torch.cuda.current_stream(device).cuda_stream


def bg_weight(bg_layer_disp, max_disp, zbuf_scale):
  return float(lib().lsi_bg_weight(float(bg_layer_disp), float(max_disp),
                                   float(zbuf_scale)))
