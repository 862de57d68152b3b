"""Layered depth images: forward splat renderer and disparity smoothness.

Mirror of the reference's `lsi/geometry/ldi.py` call surface on eager
torch.Tensors, rendering through liblsi_hip.so (HIP, gfx950).

An LDI is {textures, masks, disps}:
  textures: L x B x H x W x C   masks: L x B x H x W x 1   disps: L x B x H x W x 1
(reference ldi.py:18-21).  Any strides are accepted (e.g. a permuted NCHW conv
output) -- the kernels take element strides, nothing is copied.
"""
import collections
import ctypes
import threading
import weakref

import torch

from lsi import _C
from lsi.geometry import projection


def gradient(pred):
  """x and y forward differences (reference ldi.py:33-44). pred: L x B x H x W x C."""
  dy = pred[:, :, 1:, :, :] - pred[:, :, :-1, :, :]
  dx = pred[:, :, :, 1:, :] - pred[:, :, :, :-1, :]
  return dx, dy


def disp_smoothness_loss(pred_disp):
  """Mean absolute second differences (reference ldi.py:47-68): the fused HIP
  kernel (lsi_disp_reg_loss_fwd / _bwd); CPU tensors raise."""
  from lsi.loss import _hip  # pylint: disable=g-import-not-at-top
  return _hip.disp_regularisers(pred_disp)[0]


def _desc(tex, mask, disp, ht, wt, s, max_disp, zbuf_scale, bg_wt, flags, path,
          band_rows=0, threads=0):
  nl, b, h, w, _ = tex.shape
  d = _C.LsiSplatDesc()
  d.L, d.B, d.H, d.W, d.Ht, d.Wt = nl, b, h, w, ht, wt
  (d.tex_sl, d.tex_sb, d.tex_sy, d.tex_sx, d.tex_sc) = tex.stride()
  (d.disp_sl, d.disp_sb, d.disp_sy, d.disp_sx) = disp.stride()[:4]
  if mask is not None:
    (d.mask_sl, d.mask_sb, d.mask_sy, d.mask_sx) = mask.stride()[:4]
  d.trg_downsampling, d.max_disp, d.zbuf_scale = s, max_disp, zbuf_scale
  d.bg_wt = bg_wt
  # colour and disparity interleaved as RGBD pixels in one buffer (the 4-channel
  # output of a channels-last conv head, sliced): one 16-byte load per pixel
  if (mask is None and tex.stride()[3:] == (4, 1) and disp.stride()[3] == 4 and
      tex.stride()[:3] == disp.stride()[:3] and
      disp.data_ptr() == tex.data_ptr() + 12 and tex.data_ptr() % 16 == 0 and
      all(st % 4 == 0 for st in tex.stride()[:3])):
    flags |= _C.LSI_PACKED_RGBD
  d.flags, d.path = flags, path
  d.tune_rows, d.tune_threads = band_rows, threads
  return d


def select_path(desc, mat_host, path='auto'):
  """Chooses the kernel family and stores it (plus the STREAM window size) in
  `desc`.  `mat_host` is a CPU copy of the B x 4 x 4 projection matrices (None
  => the any-pose tile path).  path: 'auto' | 'atomic' | 'rowband' | 'stream' |
  'tile'.  auto = stream when its precondition holds (rectified pairs), else
  tile (any projection, no fp32 atomics)."""
  lib = _C.lib()
  if path in ('atomic', 'tile') or mat_host is None:
    if path in ('rowband', 'stream'):
      raise RuntimeError('%s path needs a host copy of the matrices' % path)
    desc.path = _C.LSI_PATH_ATOMIC if path == 'atomic' else _C.LSI_PATH_TILE
    return desc.path
  m = mat_host.contiguous()
  mp = ctypes.c_void_p(m.data_ptr())
  win = int(lib.lsi_stream_ok(ctypes.byref(desc), mp)) if path in (
      'auto', 'stream') else 0
  band = bool(lib.lsi_rowband_ok(ctypes.byref(desc), mp))
  if path == 'stream' and not win:
    raise RuntimeError('stream path requested but its precondition does not '
                       'hold (lsi_stream_ok)')
  if path == 'rowband' and not band:
    raise RuntimeError('rowband path requested but the projection matrices '
                       'do not satisfy its precondition (lsi_rowband_ok)')
  if win:
    desc.path, desc.tune_window = _C.LSI_PATH_STREAM, win
  elif path == 'rowband':
    desc.path = _C.LSI_PATH_ROWBAND
  else:
    desc.path = _C.LSI_PATH_TILE
  return desc.path


def plan_key(shape, trg_downsampling, max_disp, mat_host):
  """(kernel family, STREAM window) the renderer would choose for an
  L x B x H x W LDI and these host matrices: what a captured HIP graph of a
  training step bakes in (a different key needs a re-capture)."""
  nl, b, h, w = shape
  tex = torch.empty((nl, b, h, w, 3), device='meta')
  disp = torch.empty((nl, b, h, w, 1), device='meta')
  ht, wt = int(h * trg_downsampling), int(w * trg_downsampling)
  desc = _desc(tex, None, disp, ht, wt, float(trg_downsampling),
               float(max_disp), 1.0, 0.0, 0, 0)
  select_path(desc, mat_host.to(torch.float32), 'auto')
  return int(desc.path), int(desc.tune_window)


_WS_CACHE = {}
_WS_LOCK = threading.Lock()
_MAT_CACHE = collections.OrderedDict()   # (device, matrix bytes) -> device tensor
_MAT_CACHE_SIZE = 64


def _device_matrices(src2trg_mat, mat_host, dev):
  """The B x 4 x 4 matrices on the renderer's device.  A host tensor is
  uploaded once per distinct content (a few kilobytes: keyed by its bytes) and
  kept -- an eager caller that renders with the same cameras again (the two
  directions of a stereo pair, an evaluation loop) then issues no copy at all;
  a pageable host-to-device copy would wait for the kernels queued before it
  and serialise the host with the GPU.  New content goes through pinned memory,
  asynchronously."""
  if src2trg_mat.is_cuda or dev.type != 'cuda':
    # (a CPU `dev`: the call is about to fail in require_device -- no CPU path)
    return src2trg_mat.detach().to(dev, torch.float32)
  host = (mat_host if mat_host is not None else src2trg_mat)
  host = host.detach().to('cpu', torch.float32).contiguous()
  if host.dim() != 3 or tuple(host.shape[1:]) != (4, 4):
    raise ValueError('projection matrices must be B x 4 x 4, got %s' % (tuple(host.shape),))
  key = (dev.index, host.numpy().tobytes())
  with _WS_LOCK:
    hit = _MAT_CACHE.get(key)
    if hit is not None:
      _MAT_CACHE.move_to_end(key)
      return hit
  mat = host.pin_memory().to(dev, non_blocking=True)
  with _WS_LOCK:
    _MAT_CACHE[key] = mat
    while len(_MAT_CACHE) > _MAT_CACHE_SIZE:
      _MAT_CACHE.popitem(last=False)
  return mat


def _stream_workspace(desc, dev):
  """Zero-filled once, then kept by the library (lsi_hip.h, LSI_WS_KEEP): one
  buffer per (device, stream, call geometry); calls on a stream are ordered.
  Sized for the path the call takes (STREAM: counters, boundary rows and the
  disparity pass; the any-pose paths: their canvases / ranges).  A buffer that
  has been handed out is never freed or replaced: a captured HIP graph has its
  address baked in and relies on the counters the library left zero -- a call
  of the same geometry that needs more gets its own, larger buffer."""
  need = int(_C.lib().lsi_splat_workspace_bytes(ctypes.byref(desc)))
  key = (dev.index, _C.stream_ptr(dev), desc.path,
         desc.L, desc.B, desc.H, desc.W, desc.Ht, desc.Wt, desc.tune_rows,
         desc.flags & ~_C.LSI_WS_KEEP)
  with _WS_LOCK:
    kept = _WS_CACHE.setdefault(key, [])
    for ws in kept:
      if ws.numel() >= need:
        return ws, ws.numel()
    ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
    kept.append(ws)
  return ws, ws.numel()


class StreamAdapt(object):
  """The caller-owned record of the compact STREAM kernel's adaptive build
  choice (include/lsi_hip.h: LsiStreamAdapt) with the 16 bytes of device and of
  pinned host memory it points at.  One per (device, stream, call geometry);
  calls that use it hold `lock` (ctypes releases the GIL inside the call: two
  threads rendering the same geometry on the same stream must not interleave
  their probes)."""

  def __init__(self, dev):
    self.rec = _C.LsiStreamAdapt()
    self.dev_ctr = torch.zeros((4,), dtype=torch.int32, device=dev)
    self.host_ctr = torch.zeros((4,), dtype=torch.int32).pin_memory()
    self.rec.ctr_dev = self.dev_ctr.data_ptr()
    self.rec.ctr_host = self.host_ctr.data_ptr()
    self.lock = threading.Lock()

  def state(self):
    """0 undecided, 1 twelve waves x two register sets, 2 sixteen x one."""
    with self.lock:
      return int(_C.lib().lsi_stream_adapt_state(ctypes.byref(self.rec)))


_ADAPT = {}


def stream_adapt(desc, dev):
  """The record for this call geometry on the current stream (created on first
  use, kept for the life of the process: a descriptor saved for a backward or
  baked into a graph points at it), or None where the library would ignore it
  (other paths, per-layer outputs, explicit tuning)."""
  if (desc.path != _C.LSI_PATH_STREAM or not (desc.flags & _C.LSI_COMPOSE) or
      desc.tune_threads or desc.tune_rows or dev.type != 'cuda'):
    return None
  key = (dev.index, _C.stream_ptr(dev), desc.L, desc.B, desc.H,
         desc.W, desc.Ht, desc.Wt, desc.tune_window, desc.flags & _C.LSI_PACKED_RGBD,
         float(desc.trg_downsampling), float(desc.max_disp))
  with _WS_LOCK:
    ad = _ADAPT.get(key)
    if ad is None:
      ad = _ADAPT[key] = StreamAdapt(dev)
  return ad


class _NoLock(object):
  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False


_NOLOCK = _NoLock()


class _ForwardSplat(torch.autograd.Function):
  """lsi_splat_fwd / lsi_splat_bwd (include/lsi_hip.h)."""

  @staticmethod
  def forward(ctx, tex, mask, disp, mat, mat_host, cfg):
    dev = _C.require_device(tex, mask, disp, mat)
    nl, b, h, w, c = tex.shape
    if c != 3:
      raise ValueError('forward_splat renders 3-channel textures (got %d)' % c)
    s = cfg['trg_downsampling']
    ht, wt = h * s, w * s
    if ht != int(ht) or wt != int(wt):
      raise ValueError('H*trg_downsampling and W*trg_downsampling must be '
                       'integral (reference ldi.py:113-125)')
    ht, wt = int(ht), int(wt)
    flags = 0
    if cfg['compose_layers']:
      flags |= _C.LSI_COMPOSE
    if cfg['compute_trg_disp']:
      flags |= _C.LSI_WANT_DISP
    if mask is not None:
      flags |= _C.LSI_HAS_MASK
    if cfg.get('deterministic'):
      flags |= _C.LSI_DETERMINISTIC
    bg_wt = _C.bg_weight(cfg['bg_layer_disp'], cfg['max_disp'],
                         cfg['zbuf_scale'])
    desc = _desc(tex, mask, disp, ht, wt, float(s), float(cfg['max_disp']),
                 float(cfg['zbuf_scale']), bg_wt, flags, 0,
                 cfg.get('band_rows', 0), cfg.get('threads', 0))
    desc.reserved = int(cfg.get('experiment', 0))
    select_path(desc, mat_host, cfg.get('path', 'auto'))
    nlo = 1 if cfg['compose_layers'] else nl
    img = torch.empty((nlo, b, ht, wt, 3), dtype=torch.float32, device=dev)
    wts = torch.empty((nlo, b, ht, wt, 1), dtype=torch.float32, device=dev)
    dsp = (torch.empty((nlo, b, ht, wt, 1), dtype=torch.float32, device=dev)
           if cfg['compute_trg_disp'] else None)
    lib = _C.lib()
    ws, ws_bytes = None, 0
    if desc.path in (_C.LSI_PATH_ATOMIC, _C.LSI_PATH_TILE):
      ws_bytes = int(lib.lsi_splat_workspace_bytes(ctypes.byref(desc)))
      ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    elif desc.path == _C.LSI_PATH_STREAM:
      ws, ws_bytes = _stream_workspace(desc, dev)
      desc.flags |= _C.LSI_WS_KEEP
    mat = mat.contiguous()
    ad = stream_adapt(desc, dev) if cfg.get('adapt', True) else None
    desc.adapt = ctypes.addressof(ad.rec) if ad is not None else None
    with (ad.lock if ad is not None else _NOLOCK):
      rc = lib.lsi_splat_fwd(ctypes.byref(desc), _C.ptr(tex), _C.ptr(disp),
                             _C.ptr(mask), _C.ptr(mat), _C.ptr(img), _C.ptr(wts),
                             _C.ptr(dsp), _C.ptr(ws), ws_bytes,
                             _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_fwd')
    ctx.desc = desc
    ctx.has_mask = mask is not None
    ctx.save_for_backward(tex, mask if mask is not None else tex.new_empty(0),
                          disp, mat, img, wts)
    if dsp is None:
      dsp = img.new_empty(0)
    ctx.mark_non_differentiable(dsp)
    return img, wts, dsp

  @staticmethod
  def backward(ctx, g_img, g_wts, _g_dsp):
    tex, mask, disp, mat, img, wts = ctx.saved_tensors
    mask = mask if ctx.has_mask else None
    desc = ctx.desc
    dev = tex.device
    nl, b, h, w, _ = tex.shape
    g_img = g_img.contiguous() if g_img is not None else torch.zeros_like(img)
    g_wts = g_wts.contiguous() if g_wts is not None else None
    g_tex = torch.empty((nl, b, h, w, 3), dtype=torch.float32, device=dev)
    g_disp = torch.empty((nl, b, h, w, 1), dtype=torch.float32, device=dev)
    g_mask = (torch.empty((nl, b, h, w, 1), dtype=torch.float32, device=dev)
              if mask is not None else None)
    lib = _C.lib()
    ws_bytes = int(lib.lsi_splat_bwd_workspace_bytes(ctypes.byref(desc)))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = lib.lsi_splat_bwd(ctypes.byref(desc), _C.ptr(tex), _C.ptr(disp),
                           _C.ptr(mask), _C.ptr(mat), _C.ptr(img), _C.ptr(wts),
                           _C.ptr(g_img), _C.ptr(g_wts), _C.ptr(g_tex),
                           _C.ptr(g_disp), _C.ptr(g_mask), _C.ptr(ws), ws_bytes,
                           _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_bwd')
    return g_tex, g_mask, g_disp, None, None, None


class _ForwardSplatBoth(torch.autograd.Function):
  """lsi_splat_fwd_both / lsi_splat_bwd_both: the per-layer and the composed
  rendering of one LDI from one sweep over its pixels."""

  @staticmethod
  def forward(ctx, tex, mask, disp, mat, mat_host, cfg):
    dev = _C.require_device(tex, mask, disp, mat)
    nl, b, h, w, c = tex.shape
    if c != 3:
      raise ValueError('forward_splat renders 3-channel textures (got %d)' % c)
    s = cfg['trg_downsampling']
    ht, wt = h * s, w * s
    if ht != int(ht) or wt != int(wt):
      raise ValueError('H*trg_downsampling and W*trg_downsampling must be '
                       'integral (reference ldi.py:113-125)')
    ht, wt = int(ht), int(wt)
    flags = _C.LSI_HAS_MASK if mask is not None else 0
    if cfg.get('deterministic'):
      flags |= _C.LSI_DETERMINISTIC
    bg_wt = _C.bg_weight(cfg['bg_layer_disp'], cfg['max_disp'],
                         cfg['zbuf_scale'])
    desc = _desc(tex, mask, disp, ht, wt, float(s), float(cfg['max_disp']),
                 float(cfg['zbuf_scale']), bg_wt, flags, 0,
                 cfg.get('band_rows', 0), cfg.get('threads', 0))
    desc.reserved = int(cfg.get('experiment', 0))
    select_path(desc, mat_host, cfg.get('path', 'auto'))
    img = torch.empty((nl, b, ht, wt, 3), dtype=torch.float32, device=dev)
    wts = torch.empty((nl, b, ht, wt, 1), dtype=torch.float32, device=dev)
    img_c = torch.empty((1, b, ht, wt, 3), dtype=torch.float32, device=dev)
    wts_c = torch.empty((1, b, ht, wt, 1), dtype=torch.float32, device=dev)
    lib = _C.lib()
    if desc.path == _C.LSI_PATH_STREAM:
      # the stream path needs no canvases: the kept, zero-filled-once workspace
      # (no L*B*Ht*Wt*16-byte memset in every training step)
      ws, ws_bytes = _stream_workspace(desc, dev)
      desc.flags |= _C.LSI_WS_KEEP
    else:
      ws_bytes = int(lib.lsi_splat_workspace_bytes(ctypes.byref(desc)))
      ws = torch.zeros((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    mat = mat.contiguous()
    rc = lib.lsi_splat_fwd_both(ctypes.byref(desc), _C.ptr(tex), _C.ptr(disp),
                                _C.ptr(mask), _C.ptr(mat), _C.ptr(img),
                                _C.ptr(wts), _C.ptr(img_c), _C.ptr(wts_c),
                                _C.ptr(ws), ws_bytes, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_fwd_both')
    ctx.desc = desc
    ctx.has_mask = mask is not None
    ctx.save_for_backward(tex, mask if mask is not None else tex.new_empty(0),
                          disp, mat, img, wts, img_c, wts_c)
    return img, wts, img_c, wts_c

  @staticmethod
  def backward(ctx, g_img, g_wts, g_img_c, g_wts_c):
    tex, mask, disp, mat, img, wts, img_c, wts_c = ctx.saved_tensors
    mask = mask if ctx.has_mask else None
    desc = ctx.desc
    dev = tex.device
    nl, b, h, w, _ = tex.shape

    def c(t):
      return None if t is None else t.contiguous()

    g_img, g_wts, g_img_c, g_wts_c = c(g_img), c(g_wts), c(g_img_c), c(g_wts_c)
    if g_img is None and g_wts is not None:
      g_img = torch.zeros_like(img)
    if g_img_c is None and g_wts_c is not None:
      g_img_c = torch.zeros_like(img_c)
    g_tex = torch.empty((nl, b, h, w, 3), dtype=torch.float32, device=dev)
    g_disp = torch.empty((nl, b, h, w, 1), dtype=torch.float32, device=dev)
    g_mask = (torch.empty((nl, b, h, w, 1), dtype=torch.float32, device=dev)
              if mask is not None else None)
    lib = _C.lib()
    ws_bytes = int(lib.lsi_splat_bwd_workspace_bytes(ctypes.byref(desc)))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = lib.lsi_splat_bwd_both(
        ctypes.byref(desc), _C.ptr(tex), _C.ptr(disp), _C.ptr(mask),
        _C.ptr(mat), _C.ptr(img), _C.ptr(wts), _C.ptr(img_c), _C.ptr(wts_c),
        _C.ptr(g_img), _C.ptr(g_wts), _C.ptr(g_img_c), _C.ptr(g_wts_c),
        _C.ptr(g_tex), _C.ptr(g_disp), _C.ptr(g_mask), _C.ptr(ws), ws_bytes,
        _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_bwd_both')
    return g_tex, g_mask, g_disp, None, None, None


def forward_splat_both(ldi_src, src2trg_mat, trg_downsampling=1,
                       bg_layer_disp=0, max_disp=1, zbuf_scale=10,
                       mat_host=None, path='auto', deterministic=False,
                       band_rows=0, threads=0, experiment=0):
  """The two renderings the reference's training step makes of every LDI --
  per layer (compose_layers=False) and composed (compose_layers=True),
  reference ldi_enc_dec.py:302-334 -- from ONE sweep over the source pixels
  (lsi_splat_fwd_both), and one gather pass in the backward.

  Returns (img [L,B,Ht,Wt,3], wts [L,...,1], img_c [1,B,Ht,Wt,3], wts_c).
  """
  tex, mask, disp = ldi_src
  if mat_host is None and path not in ('atomic', 'tile'):
    mat_host = src2trg_mat.detach().to('cpu', torch.float32)
  mat = _device_matrices(src2trg_mat, mat_host, tex.device)
  cfg = dict(trg_downsampling=trg_downsampling, bg_layer_disp=bg_layer_disp,
             max_disp=max_disp, zbuf_scale=zbuf_scale, path=path,
             deterministic=bool(deterministic), band_rows=band_rows,
             threads=threads, experiment=experiment)
  return _ForwardSplatBoth.apply(tex, mask, disp, mat, mat_host, cfg)


def forward_splat_matrix(ldi_src, src2trg_mat, compose_layers=True,
                         compute_trg_disp=False, trg_downsampling=1,
                         bg_layer_disp=0, max_disp=1, zbuf_scale=10,
                         mat_host=None, path='auto', band_rows=0, threads=0,
                         experiment=0, deterministic=False):
  """forward_splat with the B x 4 x 4 src->trg projection matrix given as data.

  `mat_host` (optional CPU copy of the matrices) lets the row-band LDS path be
  selected without a device->host copy; when omitted and `src2trg_mat` is on the
  GPU it is fetched once (one small synchronising copy).  `band_rows`,
  `threads` and `experiment` (LsiSplatDesc.reserved) are tuning/test knobs;
  `threads=1024` asks the compact stream kernel for its 16-wave build at any
  size (by itself it takes it for small launches only): 12 - 15 % faster on
  folded / noisy disparity fields, a tie or 0.5 % slower on smooth ones.
  `deterministic` (LSI_DETERMINISTIC) asks for bitwise run-to-run reproducible
  sums on the stream path (fixed merge order; a little slower).
  """
  tex, mask, disp = ldi_src
  if mat_host is None and path not in ('atomic', 'tile'):
    mat_host = src2trg_mat.detach().to('cpu', torch.float32)
  mat = _device_matrices(src2trg_mat, mat_host, tex.device)
  cfg = dict(compose_layers=bool(compose_layers),
             compute_trg_disp=bool(compute_trg_disp),
             trg_downsampling=trg_downsampling, bg_layer_disp=bg_layer_disp,
             max_disp=max_disp, zbuf_scale=zbuf_scale, path=path,
             band_rows=band_rows, threads=threads, experiment=experiment,
             deterministic=bool(deterministic))
  img, wts, dsp = _ForwardSplat.apply(tex, mask, disp, mat, mat_host, cfg)
  if compute_trg_disp:
    return img, wts, dsp
  return img, wts


_HOST_COPIES = {}     # id(tensor) -> (weakref, version, fp32 CPU copy)
_GRID_CHECKED = {}    # id(tensor) -> (weakref, version, is the grid)


def _host_copy(x):
  """fp32 CPU copy of a camera tensor.  A tensor on the GPU costs one
  synchronising device-to-host copy (the kernel choice needs the matrices on
  the host); the copy is remembered for as long as THIS tensor object is alive
  and its version counter has not moved -- never by address, which the caching
  allocator hands out again -- so a caller that renders with the same camera
  tensors again (the two directions of a pair, an evaluation loop) pays it
  once."""
  if not x.is_cuda:
    return x.detach().to('cpu', torch.float32)
  key = id(x)
  hit = _HOST_COPIES.get(key)
  if hit is not None and hit[0]() is x and hit[1] == x._version:
    return hit[2]
  host = x.detach().to('cpu', torch.float32)
  _HOST_COPIES[key] = (weakref.ref(x, lambda _r, k=key: _HOST_COPIES.pop(k, None)),
                       x._version, host)
  return host


def _is_pixel_grid(pixel_coords_src, disps):
  """Whether `pixel_coords_src` is the pixel-centre grid helpers.pixel_coords(B,
  H, W) -- what the fused kernels generate themselves and what every call site
  of the reference passes (train_utils.py:86-88).  The WHOLE tensor is compared
  where it lives (one elementwise pass + reduction, one scalar back), once per
  tensor object and version.  None means the grid."""
  if pixel_coords_src is None:
    return True
  p = pixel_coords_src
  key = id(p)
  hit = _GRID_CHECKED.get(key)
  if hit is not None and hit[0]() is p and hit[1] == p._version:
    return hit[2]
  b, h, w = disps.shape[1:4]
  if tuple(p.shape) != (b, h, w, 3):
    raise ValueError('pixel_coords_src must be B x H x W x 3 = %s, got %s'
                     % ((b, h, w, 3), tuple(p.shape)))
  q = p.detach()
  xs = torch.arange(w, device=q.device, dtype=torch.float32) + 0.5
  ys = torch.arange(h, device=q.device, dtype=torch.float32) + 0.5
  same = bool(q.dtype == torch.float32 and
              ((q[..., 0] == xs.view(1, 1, w)) & (q[..., 1] == ys.view(1, h, 1)) &
               (q[..., 2] == 1.0)).all())
  _GRID_CHECKED[key] = (weakref.ref(p, lambda _r, k=key: _GRID_CHECKED.pop(k, None)),
                        p._version, same)
  return same


def _forward_splat_coords(ldi_src, pixel_coords_src, mat, focal_disps, compose_layers,
                          compute_trg_disp, trg_downsampling, bg_layer_disp, max_disp,
                          zbuf_scale):
  """forward_splat for source coordinates that are NOT the pixel grid
  (reference ldi.py:134: `coords_src = concat([pixel_coords_src, disps_l])`):
  the projection as helpers.transform_pts (the sequential-k product: the same
  target pixel indices as the reference), the three splats of a layer as ONE
  5-channel lsi_splat_generic (colour x weight, weight, disparity x weight),
  the normalisation as elementwise ops -- every step differentiable.  The fused
  kernels derive the grid from their thread indices and cannot take this input;
  no call site of the reference needs it, so this route is
  built from the generic ops rather than tuned."""
  from lsi.geometry import sampling  # pylint: disable=g-import-not-at-top
  from lsi.nnutils import helpers  # pylint: disable=g-import-not-at-top
  tex, masks, disps = ldi_src
  dev = _C.require_device(tex, masks, disps)
  nl, b, h, w, nc = tex.shape
  ht, wt = h * trg_downsampling, w * trg_downsampling
  if ht != int(ht) or wt != int(wt):
    raise ValueError('H*trg_downsampling and W*trg_downsampling must be '
                     'integral (reference ldi.py:113-125)')
  ht, wt = int(ht), int(wt)
  bg_wt = _C.bg_weight(bg_layer_disp, max_disp, zbuf_scale)
  pc = pixel_coords_src.to(dev, torch.float32)
  mat = mat.to(dev, torch.float32)
  s = float(trg_downsampling)
  canv = []
  for l in range(nl):
    d = disps[l]
    if focal_disps is not None:
      d = d - focal_disps.to(dev, torch.float32).view(b, 1, 1, 1)
    q = helpers.transform_pts(torch.cat([pc, d], dim=-1), mat)
    uv = helpers.divide_safe(q[..., 0:2], q[..., 2:3]) * s
    dt = helpers.divide_safe(q[..., 3:4], q[..., 2:3])
    if focal_disps is not None:
      dt = dt + focal_disps.to(dev, torch.float32).view(b, 1, 1, 1)
    pw = helpers.zbuffer_weights(dt / max_disp, scale=zbuf_scale)
    if masks is not None:
      pw = pw * masks[l]
    # (a pixel of zero weight adds nothing to the disparity canvas either:
    # oracle/lsi_oracle.py forward_splat, the build's definition for non-finite dt)
    dterm = torch.where(pw != 0, dt * pw, torch.zeros_like(dt))
    src = torch.cat([tex[l] * pw, pw, dterm], dim=-1)
    init = torch.cat([torch.full((b, ht, wt, nc + 1), bg_wt, device=dev),
                      torch.zeros((b, ht, wt, 1), device=dev)], dim=-1)
    canv.append(sampling.splat(src, uv.contiguous(), init))
  canv = torch.stack(canv)
  img, wts, dsp = canv[..., :nc], canv[..., nc:nc + 1], canv[..., nc + 1:]
  dsp = helpers.divide_safe(dsp, wts)
  if compose_layers:
    img = img.sum(dim=0, keepdim=True)
    wts = wts.sum(dim=0, keepdim=True)
    dsp = dsp.amax(dim=0, keepdim=True)
  img = helpers.divide_safe(img, wts)
  if compute_trg_disp:
    return img, wts, dsp
  return img, wts


def forward_splat(ldi_src,
                  pixel_coords_src,
                  k_s,
                  k_t,
                  rot,
                  t,
                  focal_disps=None,
                  compose_layers=True,
                  compute_trg_disp=False,
                  trg_downsampling=1,
                  bg_layer_disp=0,
                  max_disp=1,
                  zbuf_scale=10):
  """Forward splat the ldi_src (reference ldi.py:71-182; same signature).

  Args:
    ldi_src: [textures, masks, disps]; masks may be None (all ones).
    pixel_coords_src: B x H x W x 3 (or None = the pixel-centre grid).  The
        fused kernels generate (x+0.5, y+0.5, 1) themselves -- the only value the
        reference's callers pass (helpers.pixel_coords, train_utils.py:86-88):
        the tensor is compared with that grid on its device (whole tensor,
        once per tensor version) and any other coordinates are rendered by
        the generic route (_forward_splat_coords: transform_pts +
        lsi_splat_generic per layer), as ldi.py:134 would.
    k_s, k_t: B x 3 x 3 intrinsics; rot: B x 3 x 3; t: B x 3 x 1.  Camera
        tensors on the CPU avoid a device->host copy when choosing the kernel.
    focal_disps: optional B x 1 x 1 x 1 (reference ldi.py:130-143, lytro data):
        the source disparity is shifted by it before the projection and the
        target disparity shifted back.  Done around the kernels: the shifted
        disparities are one elementwise pass, and D + f = (q3 + f n) / n is
        folded into row 3 of the projection matrix (not index-critical: only
        the z-buffer weight and the disparity output see it).
  Returns:
    trg_img nl x B x Ht x Wt x 3, trg_wts nl x B x Ht x Wt x 1 (un-normalised)
    [, trg_disp nl x B x Ht x Wt x 1]; nl = 1 if compose_layers else L.
  """
  mat_host = projection.forward_projection_matrix(
      _host_copy(k_s), _host_copy(k_t), _host_copy(rot), _host_copy(t))
  if not _is_pixel_grid(pixel_coords_src, ldi_src[2]):
    return _forward_splat_coords(
        ldi_src, pixel_coords_src, mat_host, focal_disps, compose_layers, compute_trg_disp,
        trg_downsampling, bg_layer_disp, max_disp, zbuf_scale)
  if focal_disps is not None:
    tex, masks, disps = ldi_src
    f = focal_disps.detach().to(torch.float32).reshape(-1)
    if f.numel() != disps.shape[1]:
      raise ValueError('focal_disps: one value per batch element (B x 1 x 1 x 1)')
    ldi_src = [tex, masks, disps - f.to(disps.device).view(1, -1, 1, 1, 1)]
    mat_host = mat_host.clone()
    mat_host[:, 3, :] += f.to('cpu').view(-1, 1) * mat_host[:, 2, :]
  return forward_splat_matrix(
      ldi_src, mat_host, compose_layers=compose_layers,
      compute_trg_disp=compute_trg_disp, trg_downsampling=trg_downsampling,
      bg_layer_disp=bg_layer_disp, max_disp=max_disp, zbuf_scale=zbuf_scale,
      mat_host=mat_host)


def project_indices(disp, mask, src2trg_mat, trg_downsampling=1, max_disp=1,
                    zbuf_scale=10):
  """Parity/debug view: per source pixel the four flat target indices and the
  four weight-splat updates (lsi_project_indices).  disp/mask: L x B x H x W x 1.
  Returns idx4 int32 [L,B,H*W,4], upd4 fp32 [L,B,H*W,4]."""
  dev = _C.require_device(disp, mask)
  nl, b, h, w = disp.shape[:4]
  ht, wt = int(h * trg_downsampling), int(w * trg_downsampling)
  tex_stub = disp.new_empty((nl, b, h, w, 1))
  flags = _C.LSI_HAS_MASK if mask is not None else 0
  desc = _desc(tex_stub, mask, disp, ht, wt, float(trg_downsampling),
               float(max_disp), float(zbuf_scale), 0.0, flags, 0)
  mat = src2trg_mat.detach().to(dev, torch.float32).contiguous()
  idx4 = torch.empty((nl, b, h * w, 4), dtype=torch.int32, device=dev)
  upd4 = torch.empty((nl, b, h * w, 4), dtype=torch.float32, device=dev)
  rc = _C.lib().lsi_project_indices(ctypes.byref(desc), _C.ptr(disp),
                                    _C.ptr(mask), _C.ptr(mat), _C.ptr(idx4),
                                    _C.ptr(upd4), _C.stream_ptr(dev))
  _C.check(rc, 'lsi_project_indices')
  return idx4, upd4
