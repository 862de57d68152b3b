"""torch.autograd binding of the MFMA 3x3 convolution over 32 input channels
(csrc/lsi_conv.hip; include/lsi_hip.h: lsi_conv3x3_c32_fwd) -- the
full-resolution layers of the LDI heads (reference nets.py:104-111, 150-158)
on bf16 channels-last activations.

  conv3x3_c32(x, weight)            32 -> 16 | 32 channels, bf16 out (pre batch norm)
  conv3x3_c32_sigmoid(x, weight, bias)   32 -> <= 4 channels, bias + sigmoid, fp32 RGBD pixels

Forward and the data gradient of the 32 -> 32 layer run on the hand-written
kernel (the data gradient of a stride-1 SAME convolution is the same
convolution with the weights flipped and transposed); its weight gradient (a
GEMM with K = all pixels) on lsi_conv3x3_wgrad (csrc/lsi_conv_wgrad.hip: both
operands transposed by the LDS transpose read), which also takes the weight
gradients of the other 3x3 layers at full and half resolution
(conv3x3_lib_own_wgrad: forward and data gradient on MIOpen).  The prediction head is hand-written end to
end: lsi_conv3x3_pred_bwd forms sigmoid'(z) * g in registers and computes the
data gradient (K = 9 taps x 4 channels) and the 4 x 288 + 4 weight / bias
gradients (K = pixels) on the matrix cores.
"""
import os
import torch

from lsi import _C


def supported(x, cin, cout, k, stride, pred):
  """bf16 channels-last activations on the GPU, 3x3 stride 1, 32 input
  channels, widths that are multiples of 16 pixels."""
  if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16):
    return False
  if k != 3 or stride != 1 or cin != 32 or x.shape[1] != 32 or x.shape[3] % 16:
    return False
  if pred:
    if not 1 <= cout <= 4:
      return False
  elif cout not in (16, 32):
    return False
  return (x.is_contiguous(memory_format=torch.channels_last) and
          x.data_ptr() % 16 == 0)


def _launch(x, weight, cout, mode, bias=None):
  """mode 0: 32 -> cout, bf16 out; 1: prediction head (bias, sigmoid, fp32 RGBD
  pixels); 2: data gradient of the 32 -> 32 layer.  `weight`: the layer's
  parameter (cout x 32 x 3 x 3); the kernel reads it as fp32 and rounds to bf16
  itself -- no packing or casting launches."""
  n, _, h, w = x.shape
  dev = x.device
  weight = weight.detach()
  if weight.dtype != torch.float32 or not weight.is_contiguous():
    weight = weight.float().contiguous()
  if mode == 1:
    out = torch.empty((n, 4, h, w), dtype=torch.float32, device=dev,
                      memory_format=torch.channels_last)
  else:
    out = torch.empty((n, cout, h, w), dtype=torch.bfloat16, device=dev,
                      memory_format=torch.channels_last)
  rc = _C.lib().lsi_conv3x3_c32_fwd(n, h, w, cout, mode, _C.ptr(x), _C.ptr(weight),
                                    _C.ptr(bias), 1.0, _C.ptr(out),
                                    _C.stream_ptr(dev))
  _C.check(rc, 'lsi_conv3x3_c32_fwd')
  return out


# ---- weight gradients on a second stream ---------------------------------------
# In the backward pass a layer's weight gradient is needed by nobody until the
# optimiser steps, while its data gradient is on the critical path -- and most of
# that path (the bottleneck maps' convolutions, every batch-norm pass on a small
# map) is chains of dependent round trips that leave the chip idle.  With
# enable_wgrad_stream(True) a layer's backward() launches the weight gradient on
# a per-device side stream (ordered behind the incoming gradient by an event,
# BEFORE the data gradient goes onto the main stream), and a callback at the end
# of the backward pass joins the side stream back into the main one.  Inside a
# captured HIP graph the fork / join become graph edges.
#   * only when the parameter's .grad is None (autograd then takes the tensor as
#     it is: no kernel touches it before the join); accumulation into an existing
#     gradient (a flat DDP buffer, gradient accumulation) stays on the main stream;
#   * the tensors the side stream reads (saved activations, the incoming gradient)
#     are kept alive until the join -- the caching allocator would otherwise hand
#     their memory to main-stream kernels while the side stream still reads them.
# Off by default: a wrapper that consumes gradients from hooks on the main stream
# (torch DDP's reducer) would read them before the join.  The Trainer switches it
# on for single-process runs.
_WGRAD_STREAM = {'on': os.environ.get('LSI_WGRAD_STREAM', '') == '1'}
_SIDE_STREAMS = {}
_SIDE_PENDING = []   # [(main stream, side stream, kept tensors)] of the running backward pass


def enable_wgrad_stream(on=True):
  """Weight gradients on a side stream (see above); returns the previous setting."""
  old = _WGRAD_STREAM['on']
  _WGRAD_STREAM['on'] = bool(on)
  return old


# queued: the join callback of the running backward pass is on the engine's list;
# seq: _FWD_SEQ when it was queued.  A forward call of this module since then means
# that pass is over -- if its callback never ran (the pass raised), the next
# backward joins what it left and queues its own.
_SIDE_CB = {'queued': False, 'seq': -1}
_FWD_SEQ = [0]


def _join_side_streams():
  _SIDE_CB['queued'] = False
  pend, _SIDE_PENDING[:] = list(_SIDE_PENDING), []
  seen = set()
  for main, side, _ in pend:
    key = (main.cuda_stream, side.cuda_stream)
    if key in seen:
      continue
    seen.add(key)
    ev = torch.cuda.Event()
    ev.record(side)
    main.wait_event(ev)
  # (`pend` -- the kept tensors -- is dropped here: behind the join in stream order)


def _wgrad_async(weight, fn, *keep):
  """gw = fn(): the launch(es) of a weight gradient -- on the side stream when
  enabled and safe (module comment), else where the caller is."""
  if not (_WGRAD_STREAM['on'] and weight.is_cuda and weight.grad is None):
    return fn()
  dev = weight.device
  main = torch.cuda.current_stream(dev)
  side = _SIDE_STREAMS.get(dev.index)
  if side is None:
    side = _SIDE_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
  ev = torch.cuda.Event()
  ev.record(main)
  side.wait_event(ev)
  with torch.cuda.stream(side):
    gw = fn()
  if _SIDE_CB['queued'] and _SIDE_CB['seq'] != _FWD_SEQ[0]:
    _SIDE_CB['queued'] = False
  if not _SIDE_CB['queued']:
    if _SIDE_PENDING:
      # (left by a backward pass that raised before its callback ran: join it now)
      _join_side_streams()
    torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
    _SIDE_CB['queued'] = True
    _SIDE_CB['seq'] = _FWD_SEQ[0]
  _SIDE_PENDING.append((main, side, keep))
  return gw


class _Conv3x3C32(torch.autograd.Function):
  """32 -> cout (16 | 32) channels, no bias, bf16 out."""

  @staticmethod
  def forward(ctx, x, weight):
    ctx.save_for_backward(x, weight)
    return _launch(x, weight, weight.shape[0], 0)

  @staticmethod
  def backward(ctx, g):
    x, weight = ctx.saved_tensors
    g = g.contiguous(memory_format=torch.channels_last)
    gx = gw = None
    cout = weight.shape[0]
    if ctx.needs_input_grad[1]:   # (first: it may go to the side stream)
      gw = _wgrad_async(weight, lambda: _weight_grad(x, g, weight), x, g)
    if ctx.needs_input_grad[0]:
      if cout == 32 and g.dtype == torch.bfloat16 and g.data_ptr() % 16 == 0:
        # dL/dx = conv(g, W flipped in space, transposed in channels): mode 2
        gx = _launch(g, weight, 32, 2)
      else:
        gx = torch.ops.aten.convolution_backward(
            g, x, weight.to(g.dtype), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
            [True, False, False])[0]
    return gx, gw


# ---- weight gradient on the matrix cores (csrc/lsi_conv_wgrad.hip) ---------------
import os
WGRAD_MIN_PIXELS = int(os.environ.get('LSI_WGRAD_MIN_PIXELS', '200000'))  # below: aten (tools/time_wgrad.py)
_WGRAD_WS = {}


def wgrad_supported(x, cin, cout, k, stride):
  """3x3 stride-1 layers on bf16 channels-last GPU activations with channel
  counts that are multiples of 32, at resolutions where the K = pixels GEMM
  fills the chip (lsi_conv3x3_wgrad: 72 us against MIOpen's 227 at 8x32x256x768,
  108 against 238 at 8x96x128x384; a tie at 8x192x64x192)."""
  return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and k == 3 and
          stride == 1 and cin % 32 == 0 and cout % 32 == 0 and x.shape[1] == cin and
          x.shape[0] * x.shape[2] * x.shape[3] >= WGRAD_MIN_PIXELS and
          x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0)


def _wgrad_workspace(dev, nbytes):
  """Partial sums of the pixel blocks: per (device, stream) the largest buffer
  asked for so far; smaller ones stay alive (kernels in flight may use them)."""
  key = (dev.index, _C.stream_ptr(dev))
  bufs = _WGRAD_WS.setdefault(key, [])
  if not bufs or bufs[-1].numel() * 4 < nbytes:
    bufs.append(torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev))
  return bufs[-1]


def _weight_grad(x, g, weight):
  """dL/dW of a 3x3 stride-1 SAME convolution: lsi_conv3x3_wgrad where it
  applies, aten.convolution_backward (MIOpen) elsewhere."""
  cout, cin = weight.shape[:2]
  if (wgrad_supported(x, cin, cout, 3, 1) and g.dtype == torch.bfloat16 and
      g.data_ptr() % 16 == 0 and g.is_contiguous(memory_format=torch.channels_last)):
    n, _, h, w = x.shape
    dev = x.device
    lib = _C.lib()
    nbytes = lib.lsi_conv3x3_wgrad_workspace_bytes(n, h, w, cin, cout)
    ws = _wgrad_workspace(dev, nbytes)
    gw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
    rc = lib.lsi_conv3x3_wgrad(n, h, w, cin, cout, _C.ptr(x), _C.ptr(g), _C.ptr(gw),
                               _C.ptr(ws), ws.numel() * 4, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_conv3x3_wgrad')
    return gw.to(weight.dtype)
  return torch.ops.aten.convolution_backward(
      g, x, weight.to(g.dtype), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
      [False, True, False])[1].to(weight.dtype)


class _Conv3x3LibOwnWgrad(torch.autograd.Function):
  """A 3x3 stride-1 SAME convolution without bias whose forward and data
  gradient stay on the library (MIOpen) and whose weight gradient runs on
  lsi_conv3x3_wgrad: the layers of the heads the forward kernel does not take
  (`upcnv2b`: 96 -> 64 at half resolution)."""

  @staticmethod
  def forward(ctx, x, weight):
    ctx.save_for_backward(x, weight)
    return torch.ops.aten.convolution(x, weight.to(x.dtype), None, [1, 1], [1, 1], [1, 1],
                                      False, [0, 0], 1)

  @staticmethod
  def backward(ctx, g):
    x, weight = ctx.saved_tensors
    g = g.contiguous(memory_format=torch.channels_last)
    gx = gw = None
    if ctx.needs_input_grad[0]:
      gx = torch.ops.aten.convolution_backward(
          g, x, weight.to(g.dtype), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
          [True, False, False])[0]
    if ctx.needs_input_grad[1]:
      gw = _weight_grad(x, g, weight)
    return gx, gw


def conv3x3_lib_own_wgrad(x, weight):
  return _Conv3x3LibOwnWgrad.apply(x, weight)


def _pred_workspace(dev, n, h, w):
  """Partial sums of the head's weight gradient (the same never-dropped buffers
  as the other weight gradients)."""
  return _wgrad_workspace(dev, _C.lib().lsi_conv3x3_pred_bwd_workspace_bytes(n, h, w))


class _Conv3x3C32Sigmoid(torch.autograd.Function):
  """32 -> cout (<= 4) channels + bias + sigmoid, fp32 out N x 4 x H x W
  (channels last: RGBD pixels; channels past cout hold sigmoid(0))."""

  @staticmethod
  def forward(ctx, x, weight, bias):
    cout = weight.shape[0]
    y = _launch(x, weight, cout, 1,
                None if bias is None else bias.detach().float().contiguous())
    ctx.save_for_backward(x, weight, y)
    ctx.has_bias = bias is not None
    return y[:, :cout] if cout < 4 else y

  @staticmethod
  def backward(ctx, g):
    x, weight, y = ctx.saved_tensors
    cout = weight.shape[0]
    n, _, h, w = x.shape
    if cout == 4 and g.dtype == torch.float32:
      # lsi_conv3x3_pred_bwd: gz = g * y * (1 - y) is formed in registers; the
      # data gradient on the matrix cores (K = 9 taps x 4 channels), weight and
      # bias gradients on the matrix cores too (K = pixels, gz as bf16 hi + lo;
      # partial sums of the pixel blocks in the workspace, folded by a second
      # kernel)
      g = g.contiguous(memory_format=torch.channels_last)
      dev = x.device
      gx = (torch.empty((n, 32, h, w), dtype=torch.bfloat16, device=dev,
                        memory_format=torch.channels_last)
            if ctx.needs_input_grad[0] else None)
      want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
      gwb = ws = None
      if want_w:
        gwb = torch.empty((cout * 288 + cout,), dtype=torch.float32, device=dev)
        ws = _pred_workspace(dev, n, h, w)
      wt = weight.detach()
      if wt.dtype != torch.float32 or not wt.is_contiguous():
        wt = wt.float().contiguous()
      rc = _C.lib().lsi_conv3x3_pred_bwd(n, h, w, cout, _C.ptr(g), _C.ptr(y), _C.ptr(x),
                                         _C.ptr(wt), _C.ptr(gx), _C.ptr(gwb),
                                         _C.ptr(ws), 0 if ws is None else ws.numel() * 4,
                                         _C.stream_ptr(dev))
      _C.check(rc, 'lsi_conv3x3_pred_bwd')
      gw = gb = None
      if want_w:
        gw = gwb[:cout * 288].view(cout, 32, 3, 3).to(weight.dtype)
        gb = gwb[cout * 288:].to(weight.dtype) if ctx.has_bias else None
      return gx, gw, gb
    ys = y[:, :cout]
    gz = (g * ys * (1.0 - ys)).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1],
            ctx.has_bias and ctx.needs_input_grad[2]]
    gx, gw, gb = torch.ops.aten.convolution_backward(
        gz, x, weight.to(torch.bfloat16), [cout] if ctx.has_bias else None, [1, 1],
        [1, 1], [1, 1], False, [0, 0], 1, mask)
    return (gx, None if gw is None else gw.to(weight.dtype),
            None if gb is None else gb.to(weight.dtype))


def conv3x3_c32(x, weight):
  return _Conv3x3C32.apply(x, weight)


def conv3x3_c32_sigmoid(x, weight, bias):
  return _Conv3x3C32Sigmoid.apply(x, weight, bias)


# ---- every other convolution: the implicit-GEMM kernel (csrc/lsi_conv_igemm.hip) ----
import ctypes

IGEMM_MIN_PIXELS = int(os.environ.get('LSI_IGEMM_MIN_PIXELS', '0'))


def _conv_desc(n, h, w, cin, oh, ow, cout, kh, kw, stride, pad_t, pad_l):
  d = _C.LsiConvDesc()
  (d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.KH, d.KW, d.stride, d.pad_t,
   d.pad_l) = (n, h, w, cin, oh, ow, cout, kh, kw, stride, pad_t, pad_l)
  return d


def _cl_bf16(t):
  return (t.is_cuda and t.dim() == 4 and t.dtype == torch.bfloat16 and
          t.is_contiguous(memory_format=torch.channels_last) and t.data_ptr() % 16 == 0)


_DESC_OK = {}


def _desc_supported(n, h, w, cin, oh, ow, cout, k, stride, pad):
  """lsi_conv2d_supported for the geometry (the library's own word: channel
  multiples, kernel size, the 2^31-element bound of its int32 offsets), cached."""
  key = (n, h, w, cin, oh, ow, cout, k, stride, pad)
  ok = _DESC_OK.get(key)
  if ok is None:
    d = _conv_desc(n, h, w, cin, oh, ow, cout, k, k, stride, pad, pad)
    ok = _DESC_OK[key] = bool(_C.lib().lsi_conv2d_supported(ctypes.byref(d)))
  return ok


def igemm_supported(x, cin, cout, k, stride):
  """bf16 channels-last GPU activations, channel counts that are multiples of
  32, kernels up to 7 x 7, stride 1 or 2, fewer than 2^31 elements per tensor
  (lsi_conv2d_supported)."""
  if not (_cl_bf16(x) and x.shape[1] == cin and 1 <= k <= 7 and stride in (1, 2) and
          x.shape[0] * x.shape[2] * x.shape[3] >= IGEMM_MIN_PIXELS):
    return False
  n, _, h, w = x.shape
  oh, ow = -(-h // stride), -(-w // stride)
  # (TF SAME: the pad before; what SlimConv2d passes)
  pad = max((oh - 1) * stride + k - h, 0) // 2
  return _desc_supported(n, h, w, cin, oh, ow, cout, k, stride, pad)


def _f32(weight):
  weight = weight.detach()
  if weight.dtype != torch.float32 or not weight.is_contiguous():
    weight = weight.float().contiguous()
  return weight


import weakref


class _Pack(object):
  """One layer's weights in the kernel's operand order for one direction."""
  __slots__ = ('wref', 'version', 'geo', 'buf', 'mode', 'desc', 'managed')


_PACKED = {}   # (id(weight), mode) -> _Pack
_PACK_TABLE = {}  # device index -> {key: (key, device table, njobs, blocks)}


def _pack_layout(w):
  """Layout bit of lsi_conv2d_pack for a parameter it can read in place: fp32,
  stored contiguously (0) or with channels-last strides (2:
  module.to(memory_format=torch.channels_last)); None: a copy is needed."""
  if w.dtype == torch.float32:
    if w.is_contiguous():
      return 0
    if w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last):
      return 2
  return None


def _pack_source(weight):
  """(tensor whose memory lsi_conv2d_pack reads, layout bit)."""
  w = weight.detach()
  cl = _pack_layout(w)
  if cl is None:
    return w.float().contiguous(), 0
  return w, cl


_HOOK = [None]       # handle of the global optimiser post-step hook
PACK_CHECK = int(os.environ.get('LSI_PACK_CHECK', '0'))  # verify every Nth trusted pack
_CHECK_CALLS = [0]


def _install_optimizer_hook():
  """torch.optim's GLOBAL post-step hook (every optimiser of the process,
  whoever built it): after an optimiser has stepped, the packs of the parameters
  it owns are re-made -- repack_all(), one launch per device.  Installed the
  first time a trainable parameter is packed.  This is what makes a pack
  trustworthy: nothing has to remember to call anything (round 5's stale packs
  were a convention between the Trainer and this module that nothing enforced)."""
  if _HOOK[0] is None:
    from torch.optim.optimizer import register_optimizer_step_post_hook
    _HOOK[0] = register_optimizer_step_post_hook(_after_optimizer_step)


def _after_optimizer_step(optimizer, args, kwargs):
  del args, kwargs
  if not _PACKED:
    return
  owned = set()
  for group in optimizer.param_groups:
    for p in group['params']:
      owned.add(id(p))
  if any(k[0] in owned for k in _PACKED):
    repack_all(_owned=owned)


def _packed(desc, mode, weight, training=False):
  """The layer's weights in the kernel's operand order (lsi_conv2d_pack).

  Who keeps a pack fresh.  Frozen parameters: the version counter (load_state_dict,
  copy_, in-place ops move it).  Trainable parameters: torch's fused optimisers
  update them WITHOUT moving the counter, so the counter alone cannot be trusted
  (round 5: every implicit-GEMM layer of the trainer ran on its initial
  weights, silently).  Therefore
    * a trainable parameter is packed on EVERY forward call (the backward of the
      same call takes the other direction's pack from the cache: the forward call
      drops it) -- until
    * an optimiser that owns it has stepped: the global post-step hook
      (_install_optimizer_hook) then re-makes its packs after every step of ANY
      optimiser that owns it, with one launch for all layers, and marks them
      *managed*; managed packs are trusted as long as the version counter has
      not moved.  Nobody has to call anything: a plain
      `torch.optim.Adam(..., fused=True)` loop is as safe as the Trainer's.
  What is left uncovered is a write through `.data` (no counter, no optimiser)
  to a parameter whose packs are managed; LSI_PACK_CHECK=N verifies every Nth
  trusted pack against a fresh one and raises."""
  key = (id(weight), mode)
  hit = _PACKED.get(key)
  # (`training`: a forward call -- not the backward of one, which takes what the
  # forward left -- with a parameter somebody may be updating)
  training = training and weight.requires_grad
  if training:
    _install_optimizer_hook()
  if training and not (hit is not None and hit.managed):
    _PACKED.pop((id(weight), 1 - mode), None)   # (this step's backward packs afresh)
    hit_ok = False
  else:
    hit_ok = True
  geo = (desc.Cin, desc.Cout, desc.KH, desc.KW, desc.stride, desc.pad_t, desc.pad_l)
  if hit is not None and hit.wref() is weight and hit.buf.device == weight.device:
    if hit_ok and hit.version == weight._version and hit.geo == geo:
      if PACK_CHECK and hit.managed:
        _CHECK_CALLS[0] += 1
        if _CHECK_CALLS[0] % PACK_CHECK == 0:
          _verify_pack(hit, weight)
      return hit.buf
  lib = _C.lib()
  dev = weight.device
  nbytes = lib.lsi_conv2d_packed_bytes(ctypes.byref(desc))
  if (hit is not None and hit.wref() is weight and hit.buf.device == dev and
      hit.buf.numel() * 2 == nbytes and hit.geo == geo):
    buf = hit.buf          # (same place: a captured graph or a job table keeps its address)
  else:
    buf = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=dev)
  src, cl = _pack_source(weight)
  rc = lib.lsi_conv2d_pack(ctypes.byref(desc), mode | cl, _C.ptr(src), _C.ptr(buf),
                           nbytes, _C.stream_ptr(dev))
  _C.check(rc, 'lsi_conv2d_pack')
  e = _Pack()
  e.wref = weakref.ref(weight, lambda _r, k=key: _PACKED.pop(k, None))
  e.version = weight._version
  e.geo = geo
  e.buf, e.mode, e.desc = buf, mode, desc
  e.managed = False
  _PACKED[key] = e
  return buf


def _verify_pack(e, weight):
  """LSI_PACK_CHECK: the trusted pack against a fresh one (synchronises)."""
  src, cl = _pack_source(weight)
  fresh = torch.empty_like(e.buf)
  rc = _C.lib().lsi_conv2d_pack(ctypes.byref(e.desc), e.mode | cl, _C.ptr(src), _C.ptr(fresh),
                                fresh.numel() * 2, _C.stream_ptr(weight.device))
  _C.check(rc, 'lsi_conv2d_pack')
  if not torch.equal(fresh, e.buf):
    raise RuntimeError('stale packed weights: a %s parameter was updated behind the '
                       'optimiser hook and the version counter (a write through .data?); '
                       'call _hip_conv.repack_all() after such updates'
                       % (tuple(weight.shape),))


def repack_all(device=None, _owned=None):
  """Re-packs, with ONE launch per device (lsi_conv2d_pack_many), the weights of
  every layer the implicit-GEMM kernels have run so far.  The global optimiser
  hook calls it after every optimiser step (`_owned`: the ids of the stepping
  optimiser's parameters -- only their packs are refreshed and, from then on,
  *managed*: trusted by _packed()); captured in a HIP graph together with the
  optimiser's step, the packs belong to the graph.  A direct call (after a
  hand-made update, say) refreshes every pack but marks nothing: only an
  optimiser that keeps stepping keeps packs trusted.  fp32 parameters stored
  contiguously or with channels-last strides are read in place; others are
  dropped from the cache (_packed() packs a copy at the next call)."""
  if not _PACKED:
    return 0
  lib = _C.lib()
  by_dev = {}
  for k, e in list(_PACKED.items()):
    w = e.wref()
    if w is None:
      continue
    if _owned is not None and k[0] not in _owned:
      continue
    if not w.is_cuda or _pack_layout(w) is None:
      # (not fp32 in one of the two layouts: _packed() packs a copy, now)
      if device is None or w.device == device:
        _PACKED.pop(k, None)
      continue
    if device is not None and w.device != device:
      continue
    if not w.requires_grad and e.version == w._version:
      continue   # (a frozen parameter that has not moved)
    by_dev.setdefault(w.device.index, []).append((e, w))
  n = 0
  for idx, items in by_dev.items():
    key = tuple((w.data_ptr(), e.buf.data_ptr(), e.mode) for e, w in items)
    # (every table ever built stays alive under its key: a captured graph has
    # its device address baked in)
    tabs = _PACK_TABLE.setdefault(idx, {})
    tab = tabs.get(key)
    if tab is None:
      jobs = (_C.LsiPackJob * len(items))()
      nb = ctypes.c_int32(0)
      blocks = 0
      for j, (e, w) in enumerate(items):
        rc = lib.lsi_conv2d_pack_job(ctypes.byref(e.desc), e.mode | _pack_layout(w), w.data_ptr(),
                                     e.buf.data_ptr(), e.buf.numel() * 2,
                                     ctypes.byref(jobs[j]), ctypes.byref(nb))
        _C.check(rc, 'lsi_conv2d_pack_job')
        jobs[j].block0 = blocks
        blocks += nb.value
      host = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
      tab = (key, host.to(items[0][1].device), len(items), blocks)
      tabs[key] = tab
    dev = items[0][1].device
    with torch.cuda.device(dev):
      rc = lib.lsi_conv2d_pack_many(tab[1].data_ptr(), tab[2], tab[3], _C.stream_ptr(dev))
    _C.check(rc, 'lsi_conv2d_pack_many')
    for e, w in items:
      e.version = w._version
      if _owned is not None:
        e.managed = True
    n += len(items)
  return n


SPLITK = os.environ.get('LSI_IGEMM_SPLITK', '1') != '0'
_RUN_BYTES = {}


def _run_workspace(desc, mode, dev):
  """(pointer, bytes) of the workspace lsi_conv2d_run splits the contraction
  over the input channels in (the bottleneck maps); (0, 0) where it does not
  split.  The buffer is the weight gradients' (per device and stream: the calls
  that share it are ordered on that stream)."""
  if not SPLITK:
    return 0, 0
  key = (mode, desc.N, desc.H, desc.W, desc.Cin, desc.OH, desc.OW, desc.Cout, desc.KH,
         desc.KW, desc.stride, desc.pad_t, desc.pad_l)
  n = _RUN_BYTES.get(key)
  if n is None:
    n = _RUN_BYTES[key] = int(_C.lib().lsi_conv2d_workspace_bytes(ctypes.byref(desc), mode))
  if not n:
    return 0, 0
  ws = _wgrad_workspace(dev, n)
  return ws.data_ptr(), ws.numel() * 4


def _run(desc, mode, packed, dev, x=0, x2=0, out=0, out2=0, c1=0, bn_ws=0, groups=0):
  """lsi_conv2d_run: forward (mode 0) / data gradient (mode 1) of the
  descriptor's convolution with every option of the C ABI (two input tensors,
  two gradient tensors, batch-norm sums, the split over the input channels)."""
  io = _C.LsiConvIO()
  io.x, io.x2, io.packed, io.out, io.out2 = x, x2 or None, packed, out, out2 or None
  io.bn_workspace = bn_ws or None
  io.c1, io.groups = int(c1), int(groups)
  wp, wb = _run_workspace(desc, mode, dev)
  io.workspace, io.workspace_bytes = wp or None, wb
  rc = _C.lib().lsi_conv2d_run(ctypes.byref(desc), mode, ctypes.byref(io), _C.stream_ptr(dev))
  if rc:
    _C.check(rc, 'lsi_conv2d_run(mode %d)' % mode)


def _igemm(entry, desc, src, weight, out, bn_groups=0, training=False):
  """bn_groups > 0: the kernel also adds the batch-norm sums of `out` to the
  workspace (as lsi_conv2d_*_bnstats) for the lsi_bn_relu_norm that has to follow."""
  mode = 1 if entry == 'lsi_conv2d_bwd_data' else 0
  if training:          # (a forward call of a layer: see _SIDE_CB)
    _FWD_SEQ[0] += 1
  packed = _packed(desc, mode, weight, training)
  dev = src.device
  bn_ws = 0
  if bn_groups:
    from lsi.nnutils import _hip_bn  # pylint: disable=g-import-not-at-top
    bn_ws = _hip_bn.stats_workspace(tuple(out.shape), dev, 1, bn_groups).data_ptr()
  _run(desc, mode, packed.data_ptr(), dev, x=src.data_ptr(), out=out.data_ptr(), bn_ws=bn_ws,
       groups=bn_groups)
  return out


def _empty_cl(n, c, h, w, dev):
  return torch.empty((n, c, h, w), dtype=torch.bfloat16, device=dev,
                     memory_format=torch.channels_last)


OWN_WGRAD = os.environ.get('LSI_IGEMM_WGRAD', '1') != '0'
IGEMM_WGRAD_MIN_PIXELS = int(os.environ.get('LSI_IGEMM_WGRAD_MIN_PIXELS', '0'))
_WGRAD_BYTES = {}


def _igemm_wgrad_bytes(d):
  """lsi_conv2d_wgrad_workspace_bytes per geometry (0: the library takes it)."""
  key = (d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.KH, d.KW, d.stride, d.pad_t, d.pad_l)
  n = _WGRAD_BYTES.get(key)
  if n is None:
    n = int(_C.lib().lsi_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
    # (every map size since the transposing fold -- the bottleneck layers' 2 x 6
    # ... 8 x 24 maps, too: 32 - 66 us against the library's 25 - 92,
    # tools/conv_bench.py; LSI_IGEMM_WGRAD_MIN_PIXELS sends small maps back)
    if d.N * d.OH * d.OW < IGEMM_WGRAD_MIN_PIXELS:
      n = 0
    _WGRAD_BYTES[key] = n
  return n


def _igemm_wgrad(d, x, gy, weight, x2=None):
  """lsi_conv2d_wgrad[_cat]: x (and x2: the input as two tensors) = the
  descriptor's input, gy its output gradient.  The gradient comes out in the
  parameter's own memory layout (contiguous or channels-last strides), so that
  autograd's accumulation takes it as it is instead of cloning it into that
  layout (one copy kernel per parameter and step)."""
  dev = x.device
  nbytes = _igemm_wgrad_bytes(d)
  ws = _wgrad_workspace(dev, nbytes)
  cl = 2 if (weight.dim() == 4 and not weight.is_contiguous() and
             weight.is_contiguous(memory_format=torch.channels_last)) else 0
  gw = torch.empty(tuple(weight.shape), dtype=torch.float32, device=dev,
                   memory_format=torch.channels_last if cl else torch.contiguous_format)
  rc = _C.lib().lsi_conv2d_wgrad_cat(
      ctypes.byref(d), x.data_ptr(), x2.data_ptr() if x2 is not None else 0,
      x.shape[1] if x2 is not None else 0, gy.data_ptr(), gw.data_ptr(), cl, ws.data_ptr(),
      ws.numel() * 4, _C.stream_ptr(dev))
  if rc:
    _C.check(rc, 'lsi_conv2d_wgrad_cat')
  return gw if weight.dtype == torch.float32 else gw.to(weight.dtype)


class _Conv2dIgemm(torch.autograd.Function):
  """slim.conv2d without bias (reference nets.py: the arg_scope's conv2d) --
  forward and data gradient on lsi_conv2d_fwd / lsi_conv2d_bwd_data, the weight
  gradient on lsi_conv3x3_wgrad (3 x 3 stride 1 at >= 200 k pixels: the row-ring
  kernel) or lsi_conv2d_wgrad_cat (every other shape the library takes:
  lsi_conv2d_wgrad_workspace_bytes > 0); aten (MIOpen) only for what is left."""

  @staticmethod
  def forward(ctx, x, weight, stride, pad_t, pad_l, oh, ow, bn_groups=0):
    n, cin, h, w = x.shape
    cout, _, kh, kw = weight.shape
    desc = _conv_desc(n, h, w, cin, oh, ow, cout, kh, kw, stride, pad_t, pad_l)
    ctx.desc = desc
    ctx.save_for_backward(x, weight)
    return _igemm('lsi_conv2d_fwd', desc, x, weight, _empty_cl(n, cout, oh, ow, x.device),
                  bn_groups, True)

  @staticmethod
  def backward(ctx, g):
    x, weight = ctx.saved_tensors
    d = ctx.desc
    if g.dtype != torch.bfloat16:
      g = g.to(torch.bfloat16)
    g = g.contiguous(memory_format=torch.channels_last)
    gx = gw = None
    if ctx.needs_input_grad[1]:   # (first: it may go to the side stream)
      def wgrad():
        if (d.KH == 3 and d.KW == 3 and d.stride == 1 and d.pad_t == 1 and d.pad_l == 1 and
            wgrad_supported(x, d.Cin, d.Cout, 3, 1)):
          return _weight_grad(x, g, weight)    # (the row-ring kernel: full / half resolution)
        if OWN_WGRAD and _igemm_wgrad_bytes(d) > 0:
          return _igemm_wgrad(d, x, g, weight)
        # explicit padding: TF SAME is asymmetric for stride 2 (one more after)
        pb = max((d.OH - 1) * d.stride + d.KH - d.H - d.pad_t, 0)
        pr = max((d.OW - 1) * d.stride + d.KW - d.W - d.pad_l, 0)
        if pb == d.pad_t and pr == d.pad_l:
          xp, pad = x, [d.pad_t, d.pad_l]
        else:
          xp, pad = torch.nn.functional.pad(x, (d.pad_l, pr, d.pad_t, pb)), [0, 0]
        return torch.ops.aten.convolution_backward(
            g, xp, weight.to(g.dtype), None, [d.stride, d.stride], pad, [1, 1], False,
            [0, 0], 1, [False, True, False])[1].to(weight.dtype)
      gw = _wgrad_async(weight, wgrad, x, g)
    if ctx.needs_input_grad[0]:
      gx = _igemm('lsi_conv2d_bwd_data', d, g, weight,
                  _empty_cl(d.N, d.Cin, d.H, d.W, x.device))
    return gx, gw, None, None, None, None, None, None


def conv2d(x, weight, stride, pad_t, pad_l, oh, ow, bn_groups=0):
  """bn_groups > 0: the batch-norm sums of the output (that many sub-batch groups)
  are left for _hip_bn.batch_norm_relu(out, ..., groups, prestat=True), which has
  to be the next batch-norm call on this stream."""
  return _Conv2dIgemm.apply(x, weight, stride, pad_t, pad_l, oh, ow, bn_groups)


CAT_CONV = os.environ.get('LSI_CAT_CONV', '1') != '0'


def cat_supported(x1, x2, cout, k, stride):
  """A convolution over tf.concat([x1, x2], axis=3) read from the two tensors
  (lsi_conv2d_fwd_cat / _bwd_data_cat / _wgrad_cat): both bf16 channels-last on
  the GPU with the same N, H, W; channel counts multiples of 32, x1's also of the
  data-gradient kernel's channel block (64 when the sum is a multiple of 64)."""
  if not (CAT_CONV and OWN_WGRAD and _cl_bf16(x1) and _cl_bf16(x2)):
    return False
  if x1.shape[0] != x2.shape[0] or x1.shape[2:] != x2.shape[2:]:
    return False
  c1, c2 = x1.shape[1], x2.shape[1]
  blk = 64 if (c1 + c2) % 64 == 0 else 32
  if not (c1 % 32 == 0 and c2 % 32 == 0 and c1 % blk == 0 and 1 <= k <= 7 and
          stride in (1, 2) and x1.shape[0] * x1.shape[2] * x1.shape[3] >= IGEMM_MIN_PIXELS):
    return False
  n, _, h, w = x1.shape
  oh, ow = -(-h // stride), -(-w // stride)
  pad = max((oh - 1) * stride + k - h, 0) // 2
  return _desc_supported(n, h, w, c1 + c2, oh, ow, cout, k, stride, pad)


class _Conv2dCatIgemm(torch.autograd.Function):
  """slim.conv2d over a skip connection's concatenation (reference nets.py:104-106,
  300-345) without the concatenated tensor: forward, data gradient (into two
  tensors) and weight gradient read / write the two activations directly."""

  @staticmethod
  def forward(ctx, x1, x2, weight, stride, pad_t, pad_l, oh, ow, bn_groups=0):
    n, c1, h, w = x1.shape
    c2 = x2.shape[1]
    cout, cin, kh, kw = weight.shape
    assert cin == c1 + c2
    desc = _conv_desc(n, h, w, cin, oh, ow, cout, kh, kw, stride, pad_t, pad_l)
    ctx.desc = desc
    ctx.save_for_backward(x1, x2, weight)
    dev = x1.device
    out = _empty_cl(n, cout, oh, ow, dev)
    packed = _packed(desc, 0, weight, True)
    ws_ptr = 0
    if bn_groups:
      from lsi.nnutils import _hip_bn  # pylint: disable=g-import-not-at-top
      ws_ptr = _hip_bn.stats_workspace(tuple(out.shape), dev, 1, bn_groups).data_ptr()
    _run(desc, 0, packed.data_ptr(), dev, x=x1.data_ptr(), x2=x2.data_ptr(), out=out.data_ptr(),
         c1=c1, bn_ws=ws_ptr, groups=bn_groups)
    return out

  @staticmethod
  def backward(ctx, g):
    x1, x2, weight = ctx.saved_tensors
    d = ctx.desc
    dev = x1.device
    c1 = x1.shape[1]
    if g.dtype != torch.bfloat16:
      g = g.to(torch.bfloat16)
    g = g.contiguous(memory_format=torch.channels_last)
    lib = _C.lib()
    gx1 = gx2 = gw = None
    if ctx.needs_input_grad[2]:   # (first: it may go to the side stream)
      gw = _wgrad_async(weight, lambda: _igemm_wgrad(d, x1, g, weight, x2), x1, x2, g)
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
      gx1 = _empty_cl(d.N, c1, d.H, d.W, dev)
      gx2 = _empty_cl(d.N, d.Cin - c1, d.H, d.W, dev)
      packed = _packed(d, 1, weight)
      _run(d, 1, packed.data_ptr(), dev, x=g.data_ptr(), out=gx1.data_ptr(),
           out2=gx2.data_ptr(), c1=c1)
    return gx1, gx2, gw, None, None, None, None, None, None


def conv2d_cat(x1, x2, weight, stride, pad_t, pad_l, oh, ow, bn_groups=0):
  return _Conv2dCatIgemm.apply(x1, x2, weight, stride, pad_t, pad_l, oh, ow, bn_groups)


class _ConvTranspose2dIgemm(torch.autograd.Function):
  """slim.conv2d_transpose 4 x 4 stride 2 (torch ConvTranspose2d(k, stride 2,
  padding p); reference nets.py:100-103, 295-345): the data gradient of the
  forward convolution {2h x 2w x Cout_T -> h x w x Cin_T} with the same weight
  memory -- four parity classes, 2 x 2 taps each (lsi_conv2d_bwd_data); its own
  data gradient is that forward convolution (lsi_conv2d_fwd)."""

  @staticmethod
  def forward(ctx, x, weight, stride, pad, bn_groups=0):
    n, cin_t, h, w = x.shape
    _, cout_t, kh, kw = weight.shape
    desc = _conv_desc(n, stride * h, stride * w, cout_t, h, w, cin_t, kh, kw, stride, pad, pad)
    ctx.desc = desc
    ctx.save_for_backward(x, weight)
    ctx.args = (stride, pad)
    return _igemm('lsi_conv2d_bwd_data', desc, x, weight,
                  _empty_cl(n, cout_t, stride * h, stride * w, x.device), bn_groups,
                  True)

  @staticmethod
  def backward(ctx, g):
    x, weight = ctx.saved_tensors
    d = ctx.desc
    stride, pad = ctx.args
    if g.dtype != torch.bfloat16:
      g = g.to(torch.bfloat16)
    g = g.contiguous(memory_format=torch.channels_last)
    gx = gw = None
    if ctx.needs_input_grad[1]:   # (first: it may go to the side stream)
      def wgrad():
        if OWN_WGRAD and _igemm_wgrad_bytes(d) > 0:
          # the descriptor's convolution: "input" = this layer's output gradient,
          # "output gradient" = this layer's input
          return _igemm_wgrad(d, g, x, weight)
        return torch.ops.aten.convolution_backward(
            g, x, weight.to(g.dtype), None, [stride, stride], [pad, pad], [1, 1], True,
            [0, 0], 1, [False, True, False])[1].to(weight.dtype)
      gw = _wgrad_async(weight, wgrad, x, g)
    if ctx.needs_input_grad[0]:
      gx = _igemm('lsi_conv2d_fwd', d, g, weight,
                  _empty_cl(d.N, d.Cout, d.OH, d.OW, x.device))
    return gx, gw, None, None, None


def conv_transpose2d(x, weight, stride=2, pad=1, bn_groups=0):
  return _ConvTranspose2dIgemm.apply(x, weight, stride, pad, bn_groups)


def convt_supported(x, cin, cout, k, stride):
  if not (_cl_bf16(x) and x.shape[1] == cin and k <= 7 and stride == 2 and
          x.shape[0] * x.shape[2] * x.shape[3] * 4 >= IGEMM_MIN_PIXELS):
    return False
  n, _, h, w = x.shape
  # (the descriptor of _ConvTranspose2dIgemm: the convolution whose data gradient
  # this layer's forward is)
  return _desc_supported(n, stride * h, stride * w, cout, h, w, cin, k, stride, 1)


# ---- the first convolution: 3 image channels (csrc/lsi_conv_first.hip) -------------
FIRST_CONV = os.environ.get('LSI_FIRST_CONV', '1') != '0'


def first_supported(x, cin, cout, k, stride):
  """`cnv1` (reference nets.py:273): the image -- fp32 or bf16, N x C x H x W with
  channels-last strides (= the N x H x W x C tensor the module was handed), C <=
  4 -- through a 7 x 7 stride-2 convolution to 32 channels."""
  if not (FIRST_CONV and x.is_cuda and x.dim() == 4 and
          x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] == cin and cin <= 4):
    return False
  if not x.is_contiguous(memory_format=torch.channels_last):
    return False
  n, _, h, w = x.shape
  oh, ow = -(-h // stride), -(-w // stride)
  pad = max((oh - 1) * stride + k - h, 0) // 2
  key = ('first', n, h, w, cin, oh, ow, cout, k, stride, pad)
  ok = _DESC_OK.get(key)
  if ok is None:
    d = _conv_desc(n, h, w, cin, oh, ow, cout, k, k, stride, pad, pad)
    ok = _DESC_OK[key] = bool(_C.lib().lsi_conv2d_first_supported(ctypes.byref(d)))
  return ok


def _weight_layout(weight):
  """(tensor the kernels read in place, layout bit): fp32 contiguous (0) or
  channels-last strides (2); anything else is copied to fp32 contiguous."""
  w = weight.detach()
  cl = _pack_layout(w)
  if cl is None:
    return w.float().contiguous(), 0
  return w, cl


class _Conv2dFirst(torch.autograd.Function):
  """slim.conv2d(inp_img, 32, [7, 7], stride=2) without bias: forward
  (lsi_conv2d_first_fwd, batch-norm sums in the epilogue) and weight gradient
  (lsi_conv2d_first_wgrad); the image gets no gradient."""

  @staticmethod
  def forward(ctx, x, weight, stride, pad_t, pad_l, oh, ow, bn_groups=0):
    n, cin, h, w = x.shape
    cout, _, kh, kw = weight.shape
    desc = _conv_desc(n, h, w, cin, oh, ow, cout, kh, kw, stride, pad_t, pad_l)
    dev = x.device
    out = _empty_cl(n, cout, oh, ow, dev)
    wsrc, cl = _weight_layout(weight)
    ws_ptr = 0
    if bn_groups:
      from lsi.nnutils import _hip_bn  # pylint: disable=g-import-not-at-top
      ws_ptr = _hip_bn.stats_workspace(tuple(out.shape), dev, 1, bn_groups).data_ptr()
    rc = _C.lib().lsi_conv2d_first_fwd(ctypes.byref(desc), x.data_ptr(),
                                       int(x.dtype == torch.bfloat16), wsrc.data_ptr(), cl,
                                       out.data_ptr(), ws_ptr, int(bn_groups), _C.stream_ptr(dev))
    if rc:
      _C.check(rc, 'lsi_conv2d_first_fwd')
    ctx.desc = desc
    ctx.save_for_backward(x, weight)
    return out

  @staticmethod
  def backward(ctx, g):
    x, weight = ctx.saved_tensors
    d = ctx.desc
    gw = None
    if ctx.needs_input_grad[1]:
      if g.dtype != torch.bfloat16:
        g = g.to(torch.bfloat16)
      g = g.contiguous(memory_format=torch.channels_last)
      dev = x.device
      lib = _C.lib()
      nbytes = int(lib.lsi_conv2d_first_wgrad_workspace_bytes(ctypes.byref(d)))
      ws = _wgrad_workspace(dev, nbytes)
      cl = 2 if (not weight.is_contiguous() and
                 weight.is_contiguous(memory_format=torch.channels_last)) else 0
      gw = torch.empty(tuple(weight.shape), dtype=torch.float32, device=dev,
                       memory_format=torch.channels_last if cl else torch.contiguous_format)
      rc = lib.lsi_conv2d_first_wgrad(ctypes.byref(d), x.data_ptr(),
                                      int(x.dtype == torch.bfloat16), g.data_ptr(), gw.data_ptr(),
                                      cl, ws.data_ptr(), ws.numel() * 4, _C.stream_ptr(dev))
      if rc:
        _C.check(rc, 'lsi_conv2d_first_wgrad')
      if weight.dtype != torch.float32:
        gw = gw.to(weight.dtype)
    return None, gw, None, None, None, None, None, None


def conv2d_first(x, weight, stride, pad_t, pad_l, oh, ow, bn_groups=0):
  return _Conv2dFirst.apply(x, weight, stride, pad_t, pad_l, oh, ow, bn_groups)
