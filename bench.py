"""Headline benchmark: rendered views/sec of the LDI splat+warp hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3]

A "step" is one `forward_splat(compose_layers=True, compute_trg_disp=False,
trg_downsampling=0.5)` (reference ldi.py:71-182: projection + soft z-buffer
weights + 4-corner splat + normalise/compose) over one batch of synthetic LDI
tensors that are already resident in HBM.  Prints ONE JSON line (rank 0).

Workloads (BASELINE.json `configs`):
  cfg3  KITTI, 4-layer, 256x768, batch 32 -- the configuration north_star's
        target is quoted on; default
  cfg2  KITTI stereo, 2-layer, 256x768, batch 4
  cfg4  synthetic 3-layer 256x256, batch 64, general poses
  cfg5  KITTI 4-layer 512x1536, batch 8
Multi-GPU: independent LDIs, NO data-path collective (replicas of an
embarrassingly parallel renderer); RCCL is used only for the barrier and the
max-over-ranks of the elapsed time.  Default `--scaling strong`: the workload's
batch is split B/N over the ranks (SURVEY.md 8e / BASELINE config 3: 32 views
-> 4 per GPU at N = 8; cfg2 is per GPU either way) -- `value` is the rate of
that split, the one north_star's ">= 6x from 1 to 8 GPUs" is about.  The same
run then also times the weak-scaling case (every rank renders the whole batch:
the per-GPU minibatch of a data-parallel training run) and reports it under
`extra.weak`; `--scaling weak` makes that the headline instead.  `--shard-of N`
times one rank's shard of an N-rank split on a single GPU.  `python bench.py --gpus N` from a bare
shell re-launches itself under torch.distributed.run (one rank per GPU).

Inputs smaller than the 256 MiB Infinity Cache would be served from it when one
buffer is replayed, so the timed loop rotates over enough independent input
sets to exceed 1 GiB: the kernel's reads come from HBM.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))

from lsi import _C  # noqa: E402
from lsi.geometry import ldi, projection  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ROTATE_BYTES = 1 << 30   # input sets are rotated until they exceed this

WORKLOADS = {
    # name: (L, H, W, batch, batch_is_per_gpu, cameras, max_disp, bg_layer_disp)
    'cfg2': (2, 256, 768, 4, True, 'kitti', 0.4, 1e-3),
    'cfg3': (4, 256, 768, 32, False, 'kitti', 0.4, 1e-3),
    'cfg4': (3, 256, 256, 64, False, 'synthetic', 1.0, 0.2),
    'cfg5': (4, 512, 1536, 8, False, 'kitti', 0.4, 1e-3),
}
DEFAULT_WORKLOAD = 'cfg3'
ZBUF_SCALE = 50.0
S = 0.5


def algorithmic_bytes(nl, b, h, w):
  """read L*B*H*W*16 B (RGB + disparity) + B*64 B (M); write
  B*Ht*Wt*16 B (RGB + weight), compose mode."""
  return nl * b * h * w * 16 + b * 64 + b * (h // 2) * (w // 2) * 16


def box_blur(x, radius):
  k = 2 * radius + 1
  pad = torch.nn.functional.pad(x, (radius, radius, radius, radius),
                                mode='replicate')
  return torch.nn.functional.avg_pool2d(pad, k, stride=1)


def make_inputs(nl, b, h, w, cams, max_disp, seed, dev, disp_kind='smooth',
                tex_layout='nhwc'):
  """Seeded synthetic LDI + cameras.

  disp_kind:
    'smooth'  the survey's definition: max_disp * sigmoid(field) * (L-l)/L with
              field = box-blur(radius 8) of U[0,1) noise -- neighbouring pixels
              move coherently, layers are ordered front to back;
    'rough'   the same field contrast-stretched to the sigmoid's full range
              (target x-coordinate folds over itself every few pixels);
    'stress'  i.i.d. U[0.01, 1] * max_disp (worst-case scatter locality).
  """
  gen = torch.Generator(device='cpu').manual_seed(seed)
  tex = torch.rand((nl, b, h, w, 3), generator=gen).to(dev)
  if tex_layout == 'planar':  # same logical tensor stored L x B x 3 x H x W
    tex = tex.permute(0, 1, 4, 2, 3).contiguous().permute(0, 1, 3, 4, 2)
  noise = torch.rand((nl * b, 1, h, w), generator=gen).to(dev)
  if disp_kind == 'stress':
    field = (0.01 + 0.99 * noise).reshape(nl, b, h, w, 1)
  else:
    field = box_blur(noise, 8)
    if disp_kind == 'rough':
      lo = field.amin(dim=(2, 3), keepdim=True)
      hi = field.amax(dim=(2, 3), keepdim=True)
      field = torch.sigmoid(4.0 * ((field - lo) / (hi - lo + 1e-12) - 0.5))
    else:
      field = torch.sigmoid(field)
    field = field.reshape(nl, b, h, w, 1)
  scale = torch.tensor([(nl - l) / nl for l in range(nl)],
                       device=dev).view(nl, 1, 1, 1, 1)
  if disp_kind == 'stress':
    scale = torch.ones_like(scale)
  if cams == 'kitti':
    disp = (max_disp * field * scale).contiguous()
    k = torch.tensor([[0.58 * w, 0, w / 2.0], [0, 0.58 * w, h / 2.0],
                      [0, 0, 1.0]])
    k = k.expand(b, 3, 3).contiguous()
    rot = torch.eye(3).expand(b, 3, 3).contiguous()
    t = torch.tensor([[-0.532], [0.0], [0.0]]).expand(b, 3, 1).contiguous()
  else:
    disp = (0.28 + 0.22 * field * scale).contiguous()
    k = torch.tensor([[float(w), 0, w / 2.0], [0, float(h), h / 2.0],
                      [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
    rs = np.random.RandomState(seed)
    rots, ts = [], []
    for _ in range(b):  # look-at pose, syntheticPlanes/data.py:29-52
      cam = np.array([rs.uniform(-.5, .5), rs.uniform(-.5, .5), 0.0])
      at = np.array([rs.uniform(-.5, .5), rs.uniform(-.5, .5),
                     rs.uniform(3.0, 3.5)])
      z = (at - cam) / np.linalg.norm(at - cam)
      x = np.cross([0, 1.0, 0], z)
      x /= np.linalg.norm(x)
      y = np.cross(z, x)
      r = np.stack([x, y, z])
      rots.append(r)
      ts.append((-r @ cam).reshape(3, 1))
    rot = torch.tensor(np.stack(rots), dtype=torch.float32)
    t = torch.tensor(np.stack(ts), dtype=torch.float32)
  mat = projection.forward_projection_matrix(k, k, rot, t)  # host, fp32
  return tex, disp, mat


class Renderer(object):
  """Pre-bound C-ABI call: descriptor, outputs and workspace allocated once.
  `extra_sets` are further (tex, disp) input sets of the same shape; launch()
  walks through all of them round-robin (HBM-resident inputs, see module doc).
  """

  def __init__(self, tex, disp, mat_host, max_disp, bg_layer_disp, path,
               band_rows=0, threads=0, extra_sets=()):
    dev = tex.device
    nl, b, h, w, _ = tex.shape
    ht, wt = h // 2, w // 2
    bg = _C.bg_weight(bg_layer_disp, max_disp, ZBUF_SCALE)
    self.desc = ldi._desc(tex, None, disp, ht, wt, S, max_disp, ZBUF_SCALE, bg,
                          _C.LSI_COMPOSE, 0, band_rows, threads)
    ldi.select_path(self.desc, mat_host, path)
    self.path_name = _C.PATH_NAMES[self.desc.path]
    self.tex, self.disp = tex, disp
    self.sets = [(tex, disp)] + list(extra_sets)
    self.turn = 0
    self.mat = mat_host.to(dev).contiguous()
    self.img = torch.empty((1, b, ht, wt, 3), device=dev)
    self.wts = torch.empty((1, b, ht, wt, 1), device=dev)
    lib = _C.lib()
    self.ws_bytes = int(lib.lsi_splat_workspace_bytes(ctypes.byref(self.desc)))
    self.ws = torch.zeros((max(self.ws_bytes, 16),), dtype=torch.uint8,
                          device=dev)
    self.desc.flags |= _C.LSI_WS_KEEP  # zero-filled once, kept by the library
    # the caller-owned record of the adaptive build choice (LsiSplatDesc.adapt:
    # what lsi.geometry.ldi.forward_splat passes, too; the library keeps no state)
    self.adapt = ldi.stream_adapt(self.desc, dev)
    if self.adapt is not None:
      self.desc.adapt = ctypes.addressof(self.adapt.rec)
    self.fn = lib.lsi_splat_fwd
    self.dev = dev

  def launch(self):
    tex, disp = self.sets[self.turn]
    self.turn = (self.turn + 1) % len(self.sets)
    rc = self.fn(ctypes.byref(self.desc), _C.ptr(tex), _C.ptr(disp),
                 None, _C.ptr(self.mat), _C.ptr(self.img), _C.ptr(self.wts),
                 None, _C.ptr(self.ws), self.ws_bytes, _C.stream_ptr(self.dev))
    if rc != 0:
      _C.check(rc, 'lsi_splat_fwd')


def shard_batch(workload, world, scaling='strong'):
  """Per-rank batch and scaling mode: independent LDIs shard along B with no
  data-path collective.  'strong' (default): the workload's
  batch is split B/N over the ranks (config 3: 32 -> 4 per GPU at N = 8);
  'weak': every rank renders the workload's whole batch (data-parallel
  replicas, what a DDP training run does with its per-GPU minibatch).  A
  workload whose batch is per GPU (cfg2) is 'weak' either way."""
  nl, h, w, batch, per_gpu, cams, max_disp, bg = WORKLOADS[workload]
  if per_gpu or scaling == 'weak':
    return batch, 'weak'
  if batch % world:
    raise SystemExit('batch %d does not split over %d ranks' % (batch, world))
  return batch // world, 'strong'


def rotation_sets(nl, b, h, w):
  """How many independent input sets the timed loop walks through."""
  per_set = nl * b * h * w * 16
  return max(1, min(64, -(-ROTATE_BYTES // per_set)))


def reduce_max(values, dist, device):
  """MAX over ranks of a list of floats (timings); identity without dist."""
  if dist is None:
    return [float(v) for v in values]
  t = torch.tensor(values, device=device, dtype=torch.float64)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return [float(v) for v in t]


class stdout_to_stderr(object):
  """File descriptor 1 points at descriptor 2 inside the block (also for
  native libraries that write to stdout directly)."""

  def __enter__(self):
    sys.stdout.flush()
    self.saved = os.dup(1)
    os.dup2(2, 1)
    return self

  def __exit__(self, *exc):
    sys.stdout.flush()
    os.dup2(self.saved, 1)
    os.close(self.saved)
    return False


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def self_launch_argv(argv, gpus, port=None):
  """Command line that re-runs this script as `gpus` ranks on this node (what
  `python bench.py --gpus N` does when it is not already under a launcher)."""
  return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
          '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
          '--master-port', str(port or free_port()),
          os.path.abspath(__file__)] + list(argv)


def time_backward(r, iters=20):
  """Average time of lsi_splat_bwd (gradient w.r.t. textures and disparities of
  the same launch) on the renderer's inputs, HIP events on the launch stream."""
  lib = _C.lib()
  dev = r.dev
  nl, b, h, w, _ = r.tex.shape
  g_img = torch.rand_like(r.img)
  g_tex = torch.empty((nl, b, h, w, 3), device=dev)
  g_disp = torch.empty((nl, b, h, w, 1), device=dev)
  ws_bytes = int(lib.lsi_splat_bwd_workspace_bytes(ctypes.byref(r.desc)))
  ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
  turn = [0]

  def launch():
    tex, disp = r.sets[turn[0]]
    turn[0] = (turn[0] + 1) % len(r.sets)
    rc = lib.lsi_splat_bwd(ctypes.byref(r.desc), _C.ptr(tex), _C.ptr(disp),
                           None, _C.ptr(r.mat), _C.ptr(r.img), _C.ptr(r.wts),
                           _C.ptr(g_img), None, _C.ptr(g_tex), _C.ptr(g_disp),
                           None, _C.ptr(ws), ws_bytes, _C.stream_ptr(dev))
    if rc != 0:
      _C.check(rc, 'lsi_splat_bwd')

  for _ in range(3):
    launch()
  e0 = torch.cuda.Event(enable_timing=True)
  e1 = torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(iters):
    launch()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / iters


def time_both(r, iters=20):
  """lsi_splat_fwd_both (per-layer + composed outputs from one sweep) and
  lsi_splat_bwd_both on the renderer's inputs; microseconds per launch."""
  lib = _C.lib()
  dev = r.dev
  nl, b, h, w, _ = r.tex.shape
  ht, wt = h // 2, w // 2
  desc = _C.LsiSplatDesc.from_buffer_copy(r.desc)
  desc.flags = 0
  desc.path = r.desc.path
  img = torch.empty((nl, b, ht, wt, 3), device=dev)
  wts = torch.empty((nl, b, ht, wt, 1), device=dev)
  img_c = torch.empty((1, b, ht, wt, 3), device=dev)
  wts_c = torch.empty((1, b, ht, wt, 1), device=dev)
  ws_bytes = int(lib.lsi_splat_workspace_bytes(ctypes.byref(desc)))
  ws = torch.zeros((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
  g_i, g_c = torch.rand_like(img), torch.rand_like(img_c)
  g_tex = torch.empty((nl, b, h, w, 3), device=dev)
  g_disp = torch.empty((nl, b, h, w, 1), device=dev)
  bws_bytes = int(lib.lsi_splat_bwd_workspace_bytes(ctypes.byref(desc)))
  bws = torch.empty((bws_bytes,), dtype=torch.uint8, device=dev)
  turn = [0]

  def fwd():
    tex, disp = r.sets[turn[0]]
    turn[0] = (turn[0] + 1) % len(r.sets)
    rc = lib.lsi_splat_fwd_both(ctypes.byref(desc), _C.ptr(tex), _C.ptr(disp),
                                None, _C.ptr(r.mat), _C.ptr(img), _C.ptr(wts),
                                _C.ptr(img_c), _C.ptr(wts_c), _C.ptr(ws),
                                ws_bytes, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_fwd_both')

  def bwd():
    tex, disp = r.sets[turn[0]]
    rc = lib.lsi_splat_bwd_both(
        ctypes.byref(desc), _C.ptr(tex), _C.ptr(disp), None, _C.ptr(r.mat),
        _C.ptr(img), _C.ptr(wts), _C.ptr(img_c), _C.ptr(wts_c), _C.ptr(g_i), None,
        _C.ptr(g_c), None, _C.ptr(g_tex), _C.ptr(g_disp), None, _C.ptr(bws),
        bws_bytes, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_bwd_both')

  out = []
  for fn in (fwd, bwd):
    for _ in range(3):
      fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 1e3 / iters)
  return out


def time_python_surface(r, workload, iters=300):
  """The Python call surface, eager (no graph): microseconds per
  `ldi.forward_splat(ldi_src, None, k_s, k_t, rot, t, ...)` -- the reference's
  signature, ldi.py:71-83 -- on the renderer's resident inputs.  Per call the
  host computes the projection matrices, picks the kernel, allocates the
  outputs and enqueues one launch; `wall_us` is what an un-graphed caller sees
  (max of host and GPU time per call), `host_us` the time the calls take to
  enqueue.  Cameras on the CPU (the loaders' tensors) and on the GPU (one
  device-to-host copy per call: the kernel choice needs the matrices)."""
  nl, h, w, batch, per_gpu, cams, max_disp, bg = WORKLOADS[workload]
  if cams != 'kitti':
    return None
  b = r.tex.shape[1]
  k = torch.tensor([[0.58 * w, 0, w / 2.0], [0, 0.58 * w, h / 2.0],
                    [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
  rot = torch.eye(3).expand(b, 3, 3).contiguous()
  t = torch.tensor([[-0.532], [0.0], [0.0]]).expand(b, 3, 1).contiguous()
  out = {}
  for where in ('cpu', 'gpu'):
    cam = [x.to(r.dev) if where == 'gpu' else x for x in (k, k, rot, t)]
    turn = [0]

    def call():
      tex, disp = r.sets[turn[0]]
      turn[0] = (turn[0] + 1) % len(r.sets)
      return ldi.forward_splat([tex, None, disp], None, cam[0], cam[1], cam[2],
                               cam[3], compose_layers=True,
                               trg_downsampling=S, bg_layer_disp=bg,
                               max_disp=max_disp, zbuf_scale=ZBUF_SCALE)
    with torch.no_grad():
      for _ in range(10):
        call()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(iters):
        call()
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
    out['cameras_on_' + where] = {'wall_us': (t2 - t0) * 1e6 / iters,
                                  'host_us': (t1 - t0) * 1e6 / iters}
  with torch.no_grad():
    tex, disp = r.sets[0]
    for _ in range(5):
      ldi.forward_splat_both([tex, None, disp], r.mat.cpu(), S, bg, max_disp, ZBUF_SCALE)
    torch.cuda.synchronize()
    mat_host = r.mat.cpu()
    t0 = time.perf_counter()
    for i in range(iters // 3):
      tex, disp = r.sets[i % len(r.sets)]
      ldi.forward_splat_both([tex, None, disp], mat_host, S, bg, max_disp,
                             ZBUF_SCALE, mat_host=mat_host)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
  out['forward_splat_both'] = {'wall_us': (t2 - t0) * 1e6 / (iters // 3),
                               'host_us': (t1 - t0) * 1e6 / (iters // 3)}
  return out


def backward_bytes(nl, b, h, w):
  """the backward re-reads the inputs (16 B / source px), reads
  g_img (12 B) and the saved img/wts (16 B) per target px and writes g_tex and
  g_disp (16 B / source px)."""
  return nl * b * h * w * 32 + b * (h // 2) * (w // 2) * 28


def cpu_baseline(nl, b, h, w, cams, max_disp, bg, budget_s=10.0):
  """CPU baselines on this host's cores, on a bounded sample of the same
  workload (same shapes and cameras, batch capped):
    A  'tf_graph': torch-CPU restatement that executes the reference's own op
       decomposition (per layer x channel x corner scatter into a fresh zero
       canvas + add, every elementwise op its own pass: sampling.py:246-252,
       ldi.py:129-155) -- the stand-in for "the reference's CPU path" (TF1 is
       not installable here);
    B  'fused_port': oracle/lsi_ref_cpu.c, one fused C + OpenMP pass.
  Both are test infrastructure used here only as the reported baseline."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import ref_cpu
  import baseline_tf_graph
  out = {}
  # ---- B: fused C port, canvases allocated once outside the timed loop ------
  bb = min(b, 4)
  tex, disp, mat = make_inputs(nl, bb, h, w, cams, max_disp, 123,
                               torch.device('cpu'))
  tex_n, disp_n, mat_n = tex.numpy(), disp.numpy(), mat.numpy()
  cores = ref_cpu.num_threads()
  ctx = ref_cpu.Context(nl, bb, h, w, h // 2, w // 2)
  ctx.forward_splat(tex_n, None, disp_n, mat_n, S, bg, max_disp, ZBUF_SCALE)
  times = []
  t_end = time.time() + budget_s
  while len(times) < 3 or (time.time() < t_end and len(times) < 200):
    t0 = time.perf_counter()
    ctx.forward_splat(tex_n, None, disp_n, mat_n, S, bg, max_disp, ZBUF_SCALE)
    times.append(time.perf_counter() - t0)
  med = float(np.median(times))
  fused = {
      'value': bb / med, 'unit': 'views/s', 'cores': cores, 'kind': 'port',
      'sample': '%d runs of oracle/lsi_ref_cpu.c (fused C + OpenMP, canvases '
                'pre-allocated) on L=%d B=%d %dx%d, median' %
                (len(times), nl, bb, h, w),
  }
  # ---- A: the reference's op decomposition on torch-CPU ----------------------
  ba = min(b, 2)
  threads = torch.get_num_threads()
  a_tex, a_disp, a_mat = tex[:, :ba].contiguous(), disp[:, :ba].contiguous(), mat[:ba]
  baseline_tf_graph.forward_splat(a_tex, None, a_disp, a_mat, S, bg, max_disp,
                                  ZBUF_SCALE, True)
  times = []
  t_end = time.time() + budget_s
  while len(times) < 3 or (time.time() < t_end and len(times) < 100):
    t0 = time.perf_counter()
    baseline_tf_graph.forward_splat(a_tex, None, a_disp, a_mat, S, bg,
                                    max_disp, ZBUF_SCALE, True)
    times.append(time.perf_counter() - t0)
  med = float(np.median(times))
  out = {
      'value': ba / med, 'unit': 'views/s', 'cores': threads, 'kind': 'port',
      'sample': '%d runs of oracle/baseline_tf_graph.py (torch-CPU, the '
                'reference\'s op-for-op decomposition: 20 scatters per layer '
                'into fresh canvases) on L=%d B=%d %dx%d, median; host has %d '
                'logical cores' % (len(times), nl, ba, h, w, os.cpu_count()),
      'fused_port': fused,
  }
  return out


def measure_traffic(args, timeout_s=150):
  """HBM bytes per launch of the splat kernel from rocprofv3 PMC counters,
  collected by re-running this script (inner mode: a few eager launches) in two
  separate --pmc passes, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
  WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at
  half their size, so traffic = (2*FETCH + WRITE) * 1024.  None when the
  profiler is unavailable or a pass fails."""
  import csv
  import glob
  import shutil
  import tempfile
  exe = shutil.which('rocprofv3')
  if exe is None:
    return None
  vals = {}
  env = dict(os.environ, TMPDIR='/tmp')
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    out = tempfile.mkdtemp(prefix='lsi_pmc_', dir='/tmp')
    cmd = [exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv',
           '-d', out, '-o', 'p', '--', sys.executable, os.path.abspath(__file__),
           '--inner-traffic', '--workload', args.workload, '--path', args.path,
           '--disp', args.disp]
    try:
      subprocess.run(cmd, cwd='/tmp', env=env, timeout=timeout_s,
                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                     check=True)
      got = {}
      for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'),
                         recursive=True):
        for row in csv.DictReader(open(f)):
          name = row['Kernel_Name']
          if ((('splat_' in name and '_kernel' in name) or 'disp_range' in name)
              and 'bwd' not in name and row['Counter_Name'] == counter):
            got.setdefault(name, []).append(float(row['Counter_Value']))
      if not got:
        return None
      # (the any-pose path is two kernels per launch: their averages add up;
      # later launches only: caches in steady state)
      vals[counter] = sum(sum(v[len(v) // 2:]) / len(v[len(v) // 2:])
                          for v in got.values())
    except Exception:  # pylint: disable=broad-except
      return None
    finally:
      shutil.rmtree(out, ignore_errors=True)
  return int((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0)


MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (MI355X_MICROARCH.md)
# forward GFLOP of the network per IMAGE: U-Net + L heads; a
# training sample is two images (source and target view), forward + both gradients
TRAIN_CONFIGS = [
    # name, n_layers, H, W, batch per GPU, forward GFLOP per image
    ('L4_256x768_b4', 4, 256, 768, 4, 113.8),   # BASELINE config 3's model at its per-GPU batch
    ('L2_256x768_b4', 2, 256, 768, 4, 67.8),    # BASELINE config 2
    ('L3_256x256_b16', 3, 256, 256, 16, 30.3),  # BASELINE config 4's model
]


def train_step_extra(dev, budget_s, steps=16):
  """the full training step -- network (bf16 convolutions on
  the own MFMA kernels, fp32 accumulation), both directions' per-layer + composed
  renderings and their backward, the six losses, fused Adam and the re-pack of
  the weights -- as samples/s, eager and captured in a HIP graph, with the MFMA
  roofline beside it: tflops = samples/s x 2 images x 3 (forward + two gradients)
  x the forward GFLOP of SURVEY 8(d), against the dense bf16 peak.  Synthetic
  pairs, random-init weights.  Bounded: a configuration is only started while
  `budget_s` seconds are not used up."""
  import ldi_enc_dec as script  # layered-scene-inference_amd/ldi_enc_dec.py
  out = {'note': 'full train step (U-Net + heads bf16 on own MFMA kernels, 2 x '
                 'forward_splat_both + backward, losses, fused Adam, weight re-pack); '
                 'tflops = samples/s x 6 x forward GFLOP per image (SURVEY 8d); '
                 'peak = %.0f TFLOP/s dense bf16; eager = weight gradients and the per-layer '
                 'decoders on side streams, graph = one stream captured; not part of `value`'
                 % MFMA_PEAK_TFLOPS,
         'configs': {}}
  t_start = time.perf_counter()
  for name, nl, h, w, b, gflop in TRAIN_CONFIGS:
    if time.perf_counter() - t_start > budget_s:
      out['configs'][name] = {'skipped': 'wall-time budget of %.0f s used up' % budget_s}
      continue
    res = {'n_layers': nl, 'hw': [h, w], 'batch': b, 'fwd_gflop_per_image': gflop}
    for mode in ('graph', 'eager'):
      if time.perf_counter() - t_start > budget_s:
        res[mode] = {'skipped': 'wall-time budget used up'}
        continue
      try:
        argv = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', str(b),
                '--n_layers', str(nl), '--img_height', str(h), '--img_width', str(w),
                '--checkpoint_dir', '/tmp/lsi_bench_ckpt', '--save_latest_freq', '1000000',
                '--checkpoint_freq', '1000000', '--log_freq', '1000000', '--bf16', 'true',
                '--hip_graph', 'true' if mode == 'graph' else 'false']
        opts = script.apply_dataset_overrides(script.build_parser().parse_args(argv))
        tr = script.Trainer(opts)
        tr.local_rank = dev.index or 0
        tr.setup()
        for _ in range(tr.GRAPH_WARMUP + 2 if mode == 'graph' else 3):
          total, _ = tr.train_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
          total, _ = tr.train_step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        loss = float(total)
        if not np.isfinite(loss):
          raise RuntimeError('loss is not finite')
        sps = b / dt
        tf = sps * 6.0 * gflop / 1e3
        res[mode] = {'samples_per_s': sps, 'ms_per_step': dt * 1e3, 'tflops': tf,
                     'frac_of_mfma_peak': tf / MFMA_PEAK_TFLOPS, 'loss': loss}
        del tr
      except Exception as e:  # pylint: disable=broad-except
        res[mode] = {'error': '%s: %s' % (type(e).__name__, e)}
      torch.cuda.empty_cache()
    done = [(res[m]['samples_per_s'], m) for m in ('graph', 'eager') if 'samples_per_s' in res[m]]
    if done:
      res['best'] = {'mode': max(done)[1], 'samples_per_s': max(done)[0],
                     'frac_of_mfma_peak': res[max(done)[1]]['frac_of_mfma_peak']}
    out['configs'][name] = res
  out['wall_s'] = time.perf_counter() - t_start
  return out


def build_renderer(workload, b_local, seed, dev, args):
  nl, h, w, batch, per_gpu, cams, max_disp, bg = WORKLOADS[workload]
  tex, disp, mat = make_inputs(nl, b_local, h, w, cams, max_disp, seed, dev,
                               args.disp, args.tex_layout)
  extra = []
  for i in range(1, rotation_sets(nl, b_local, h, w)):
    t2, d2, _ = make_inputs(nl, b_local, h, w, cams, max_disp,
                            seed + 7919 * i, dev, args.disp, args.tex_layout)
    extra.append((t2, d2))
  return Renderer(tex, disp, mat, max_disp, bg, args.path, args.band_rows,
                  args.threads, extra)


SETTLE_S = 0.2  # untimed replays before the timed region (clock ramp)


def timed_region(r, steps, warmup, launch_mode, dist, dev):
  """W untimed steps (plus SETTLE_S seconds of untimed replays: clock ramp),
  then exactly K steps between barrier + synchronize on both sides; HIP events
  on the launch stream bracket the same K launches."""
  stream = torch.cuda.Stream(device=dev)
  with torch.cuda.stream(stream):
    for _ in range(max(warmup, 1)):
      r.launch()
    stream.synchronize()
    graph = None
    if launch_mode == 'graph':
      try:
        graph = torch.cuda.CUDAGraph()
        # (thread_local: the RCCL watchdog thread of an N > 1 run may query its
        # events while this thread captures)
        with torch.cuda.graph(graph, stream=stream,
                              capture_error_mode='thread_local'):
          for _ in range(steps):
            r.launch()
        graph.replay()          # one untimed replay (graph upload)
        stream.synchronize()
      except Exception as e:  # pylint: disable=broad-except
        sys.stderr.write('graph capture failed (%s); eager launches\n' % e)
        graph, launch_mode = None, 'eager'
    # Settle: a GPU that has just left idle runs its first milliseconds at
    # lower clocks (cfg3: 112 us per step with --steps 20 --warmup 5, 98 us
    # once warm).  Untimed replays of the same work for SETTLE_S seconds, so
    # that short runs measure the same steady state as long ones.
    t_settle = time.perf_counter() + SETTLE_S
    while time.perf_counter() < t_settle:
      if graph is not None:
        graph.replay()
      else:
        for _ in range(steps):
          r.launch()
      stream.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    if graph is not None:
      graph.replay()
    else:
      for _ in range(steps):
        r.launch()
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
  return elapsed, ev_ms, launch_mode


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--workload', default=DEFAULT_WORKLOAD,
                  choices=sorted(WORKLOADS))
  ap.add_argument('--path', default='auto', choices=['auto', 'atomic', 'rowband', 'stream', 'tile'])
  ap.add_argument('--launch', default='graph', choices=['graph', 'eager'])
  ap.add_argument('--band-rows', type=int, default=0)
  ap.add_argument('--threads', type=int, default=0)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extra', action='store_true',
                  help='skip the extras (backward, other workloads, traffic)')
  ap.add_argument('--train-step-budget', type=float, default=45.0,
                  help='seconds of wall time for extra.train_step (the full '
                  'training step as samples/s and fraction of the MFMA peak); 0: skip')
  ap.add_argument('--traffic', default='measure', choices=['measure', 'off'],
                  help='roofline.traffic: two rocprofv3 --pmc passes of this '
                  'script (N=1 only), or null')
  ap.add_argument('--inner-traffic', action='store_true',
                  help=argparse.SUPPRESS)
  ap.add_argument('--selftest-backend', default=None, choices=['gloo'],
                  help='CPU check of the N>1 launch/shard/reduce logic (no '
                  'GPU work): tests/test_dist_cpu.py')
  ap.add_argument('--debug-flags', type=int, default=0,
                  help='LsiSplatDesc.reserved: planner experiments (bits 12+) work '
                  'with any build; the kernel timing hooks (bits 0-9) need '
                  'LSI_HIP_LIB=hooks (build.py --hooks)')
  ap.add_argument('--scaling', default='strong', choices=['weak', 'strong'],
                  help='N > 1: strong = the workload\'s batch is split B/N over '
                  'the ranks (default; the weak figure is added under extra.weak), '
                  'weak = every rank renders the whole batch (per-GPU work fixed)')
  ap.add_argument('--shard-of', type=int, default=1,
                  help='N=1 only: time the per-rank shard of an N-rank run (what '
                  'one GPU of --gpus N renders); the JSON line is then about that '
                  'shard, not the workload')
  ap.add_argument('--disp', default='smooth',
                  choices=['smooth', 'rough', 'stress'])
  ap.add_argument('--tex-layout', default='nhwc', choices=['nhwc', 'planar'],
                  help='storage of the textures: the reference\'s channels-last '
                  'L x B x H x W x 3 (default) or planar L x B x 3 x H x W (what '
                  'a conv decoder emits); same logical tensor either way')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus > 1 and 'RANK' not in os.environ:
    # bare `python bench.py --gpus N`: become the launcher (one rank per GPU)
    # (the port was free a moment ago; if another process took it before the
    # rendezvous bound it -- EADDRINUSE -- once more on a fresh one)
    # The ranks' stderr is forwarded line by line as it comes (progress, and
    # whatever a hang or a kill leaves behind, stay visible), and searched for
    # the rendezvous error; stdout -- the one JSON line -- is held back until
    # the attempt is known to be the last.
    rc = 1
    for attempt in range(4):
      proc = subprocess.Popen(self_launch_argv(sys.argv[1:], args.gpus), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, bufsize=1)
      taken = [False]

      def pump(taken=taken, proc=proc):
        for line in proc.stderr:
          taken[0] = taken[0] or 'EADDRINUSE' in line
          sys.stderr.write(line)
          sys.stderr.flush()
      import threading
      th = threading.Thread(target=pump, daemon=True)
      th.start()
      child_out = proc.stdout.read()
      rc = proc.wait()
      th.join(timeout=10)
      if rc == 0 or not taken[0]:
        sys.stdout.write(child_out)
        break
      sys.stderr.write('bench.py: rendezvous port taken (attempt %d), once more on a fresh one\n'
                       % (attempt + 1))
    sys.exit(rc)
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus > 1 and world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  dist = None
  selftest = args.selftest_backend is not None
  if world > 1 or 'RANK' in os.environ:  # launched by torch.distributed.run
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if selftest:
      dist.init_process_group(args.selftest_backend, rank=rank,
                              world_size=world)
    else:
      # (RCCL prints a version banner on stdout when its communicator comes up:
      # stdout is kept for the one JSON line, the banner goes to stderr)
      if os.environ.get('LSI_BENCH_SHARE_GPU') == '1':
        # (tests: all ranks on device 0 of a one-GPU box, rendezvous over gloo
        # -- RCCL refuses two ranks on one device; the N > 1 control flow is
        # the same: split, barriers, max over ranks, the weak leg)
        local_rank = 0
        with stdout_to_stderr():   # (gloo announces its peers on stdout)
          dist.init_process_group('gloo', rank=rank, world_size=world)
          dist.barrier()
      else:
        with stdout_to_stderr():
          dist.init_process_group('nccl', rank=rank, world_size=world,
                                  device_id=torch.device('cuda', local_rank))
          dist.barrier()
          torch.cuda.synchronize()
  nl, h, w, batch, per_gpu, cams, max_disp, bg = WORKLOADS[args.workload]
  # (--shard-of N times what one GPU of a strong-scaling N-rank run renders)
  b_local, scaling = shard_batch(
      args.workload, max(world, args.shard_of),
      'strong' if args.shard_of > 1 else args.scaling)

  if selftest:
    # the rank plumbing without a GPU: shard, barrier, max-over-ranks, one line
    dev = torch.device('cpu')
    dist.barrier()
    fake = 1.0e-3 * (1 + rank)
    elapsed, = reduce_max([fake], dist, dev)
    if rank == 0:
      print(json.dumps({'selftest': True, 'n_gpus': world, 'scaling': scaling,
                        'views_per_step': b_local * world,
                        'elapsed_max': elapsed}))
    dist.destroy_process_group()
    return

  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)

  if args.inner_traffic:  # profiled by measure_traffic(): a few eager launches
    r = build_renderer(args.workload, b_local, 1000, dev, args)
    for _ in range(2 * len(r.sets) + 4):
      r.launch()
    torch.cuda.synchronize()
    return

  r = build_renderer(args.workload, b_local, 1000 + rank, dev, args)
  r.desc.reserved = args.debug_flags
  elapsed, ev_ms, launch_mode = timed_region(r, args.steps, args.warmup,
                                             args.launch, dist, dev)
  ev_ms_rank = ev_ms
  elapsed, ev_ms = reduce_max([elapsed, ev_ms], dist, dev)
  # per-rank average launch time: slowest and fastest rank (N > 1: which GPU
  # holds the job back)
  ev_min = -reduce_max([-ev_ms_rank], dist, dev)[0]
  # N > 1, batch split over the ranks: the weak-scaling figure of the same run
  # (every rank renders the workload's whole batch) goes under extra.weak
  weak = None
  if world > 1 and scaling == 'strong' and args.shard_of == 1:
    del r
    torch.cuda.empty_cache()
    rw = build_renderer(args.workload, batch, 1000 + rank, dev, args)
    rw.desc.reserved = args.debug_flags
    w_elapsed, w_ev, _ = timed_region(rw, args.steps, args.warmup, args.launch,
                                      dist, dev)
    w_elapsed, w_ev = reduce_max([w_elapsed, w_ev], dist, dev)
    weak = {'value': batch * world * args.steps / w_elapsed, 'unit': 'views/s',
            'scaling': 'weak', 'views_per_gpu': batch,
            'ms_per_step': w_elapsed * 1e3 / args.steps,
            'avg_launch_us': w_ev * 1e3 / args.steps,
            'note': 'every rank renders the workload\'s whole batch (per-GPU '
                    'work fixed); `value` above is the B/N split'}
    r = rw

  if rank == 0:
    views = b_local * world * args.steps
    alg = algorithmic_bytes(nl, b_local, h, w)
    kern_s = ev_ms * 1e-3 / args.steps
    achieved = alg / kern_s / 1e9
    out = {
        'metric': 'rendered views/sec (BxLxHxW splat+warp)',
        'value': views / elapsed, 'unit': 'views/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True,
        'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': '%s: %d-layer LDI %dx%d, batch %d per GPU (%d total), '
                        'trg_downsampling 0.5, %s cameras, compose_layers, '
                        '%s disparities' %
                        (args.workload, nl, h, w, b_local, b_local * world,
                         cams, args.disp),
            'kernel_path': r.path_name, 'launch': launch_mode,
            'untimed_settle_s': SETTLE_S,
            'input_sets_rotated': len(r.sets),
            'parallelism': 'batch-sharded replicas x%d (no collective)' % world,
        },
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
            'traffic': None,
            # the any-pose path: disp_range_kernel + splat_sweep_kernel, both
            # inside the measured launch
            'kernel': ('splat_sweep_kernel (+ disp_range_kernel)'
                       if r.path_name == 'tile' else
                       # compose mode, no mask, unit normaliser, channels-last,
                       # rows of whole 256-pixel segments: the compact instance
                       # of any width that is a multiple of 4 pixels
                       'splat_stream2_kernel' if (r.path_name == 'stream' and
                                                  w % 4 == 0 and
                                                  args.tex_layout == 'nhwc')
                       else 'splat_%s_kernel' % r.path_name),
            'algorithmic_bytes_per_launch': alg,
            'avg_launch_us': kern_s * 1e6,
            'avg_launch_us_per_rank': {
                'max': ev_ms * 1e3 / args.steps, 'min': ev_min * 1e3 / args.steps},
        },
    }
    if weak is not None:
      out['extra'] = {'weak': weak}
    if world == 1 and not args.no_extra:
      extra = {}
      bwd_us = time_backward(r)
      balg = backward_bytes(nl, b_local, h, w)
      extra['backward'] = {
          'us_per_launch': bwd_us,
          'frac_of_hbm_peak': balg / (bwd_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
          'algorithmic_bytes': balg,
          'note': 'lsi_splat_bwd on the same inputs (eager launches); not part '
                  'of `value`',
      }
      extra['fwd_bwd'] = {
          'views_per_s': b_local / ((kern_s * 1e6 + bwd_us) * 1e-6),
          'frac_of_hbm_peak': (alg + balg) / ((kern_s * 1e6 + bwd_us) * 1e-6) /
                              1e9 / HBM_PEAK_GBPS,
      }
      try:
        f_us, b_us = time_both(r)
        # read the inputs once, write L + 1 rendered views; backward as above
        # plus the composed view's gradient
        both_alg = (nl * b_local * h * w * 16 +
                    (nl + 1) * b_local * (h // 2) * (w // 2) * 16)
        extra['both_outputs'] = {
            'fwd_us': f_us, 'bwd_us': b_us,
            'fwd_frac_of_hbm_peak': both_alg / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            'note': 'lsi_splat_fwd_both / lsi_splat_bwd_both: the per-layer and '
                    'the composed view of a training step from one sweep',
        }
      except Exception as e:  # pylint: disable=broad-except
        extra['both_outputs'] = {'error': str(e)}
      try:
        ps = time_python_surface(r, args.workload)
        if ps is not None:
          ps['graph_launch_us'] = kern_s * 1e6
          extra['python_surface'] = ps
      except Exception as e:  # pylint: disable=broad-except
        extra['python_surface'] = {'error': str(e)}
      del r
      torch.cuda.empty_cache()
      other = {}
      for wl in sorted(WORKLOADS):
        if wl == args.workload:
          continue
        try:
          o_nl, o_h, o_w = WORKLOADS[wl][:3]
          o_b, _ = shard_batch(wl, 1)
          ro = build_renderer(wl, o_b, 1000, dev, args)
          _, o_ms, _ = timed_region(ro, 50, 5, args.launch, None, dev)
          o_s = o_ms * 1e-3 / 50
          o_alg = algorithmic_bytes(o_nl, o_b, o_h, o_w)
          other[wl] = {
              'us_per_launch': o_s * 1e6, 'views_per_s': o_b / o_s,
              'frac_of_hbm_peak': o_alg / o_s / 1e9 / HBM_PEAK_GBPS,
              'kernel_path': ro.path_name, 'batch': o_b,
          }
          del ro
          torch.cuda.empty_cache()
        except Exception as e:  # pylint: disable=broad-except
          other[wl] = {'error': str(e)}
      extra['other_workloads'] = other
      if args.train_step_budget > 0:
        try:
          extra['train_step'] = train_step_extra(dev, args.train_step_budget)
        except Exception as e:  # pylint: disable=broad-except
          extra['train_step'] = {'error': '%s: %s' % (type(e).__name__, e)}
      out['extra'] = extra
    if world == 1 and args.traffic == 'measure':
      out['roofline']['traffic'] = measure_traffic(args)
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(nl, b_local, h, w, cams, max_disp, bg)
    print(json.dumps(out))
  if dist is not None:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
