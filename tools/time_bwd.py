"""Times lsi_splat_bwd and lsi_splat_fwd_both / lsi_splat_bwd_both on a bench
workload (HIP events; microseconds per call).  LSI_BWD_STREAM=0 selects the
one-thread-per-pixel gather kernel, LSI_HIP_LIB=<name> an experiment build.
  python tools/time_bwd.py [--workload cfg3] [--shard-of N]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='cfg3')
ap.add_argument('--shard-of', type=int, default=1)
ap.add_argument('--disp', default='smooth')
ap.add_argument('--tex-layout', default='nhwc')
ap.add_argument('--path', default='auto')
ap.add_argument('--band-rows', type=int, default=0)
ap.add_argument('--threads', type=int, default=0)
args = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
b_local, _ = bench.shard_batch(args.workload, args.shard_of, 'strong')
r = bench.build_renderer(args.workload, b_local, 1000, dev, args)
r.launch()
torch.cuda.synchronize()
nl, h, w = bench.WORKLOADS[args.workload][:3]
bwd = min(bench.time_backward(r) for _ in range(3))
both = [min(x) for x in zip(*[bench.time_both(r) for _ in range(3)])]

def time_disp(path, indep=False):
  """forward_splat(compose_layers=True, compute_trg_disp=True): the evaluation
  script's call; path = the renderer's own (STREAM: two launches) or TILE.
  indep: compose_layers=False without the disparity output instead."""
  import ctypes
  from lsi import _C
  lib = _C.lib()
  desc = _C.LsiSplatDesc.from_buffer_copy(r.desc)
  desc.flags = r.desc.flags | _C.LSI_WANT_DISP
  desc.path = path
  out_disp = torch.empty_like(r.wts)
  img_o, wts_o = r.img, r.wts
  if indep:
    desc.flags = r.desc.flags & ~(_C.LSI_COMPOSE | _C.LSI_WANT_DISP)
    nl_ = r.tex.shape[0]
    img_o = torch.empty((nl_,) + tuple(r.img.shape[1:]), device=dev)
    wts_o = torch.empty((nl_,) + tuple(r.wts.shape[1:]), device=dev)
  ws_bytes = int(lib.lsi_splat_workspace_bytes(ctypes.byref(desc)))
  ws = torch.zeros((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
  turn = [0]
  def launch():
    tex, disp = r.sets[turn[0]]
    turn[0] = (turn[0] + 1) % len(r.sets)
    rc = lib.lsi_splat_fwd(ctypes.byref(desc), _C.ptr(tex), _C.ptr(disp), None,
                           _C.ptr(r.mat), _C.ptr(img_o), _C.ptr(wts_o),
                           _C.ptr(out_disp), _C.ptr(ws), ws_bytes, _C.stream_ptr(dev))
    _C.check(rc, 'lsi_splat_fwd')
  for _ in range(3):
    launch()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(20):
    launch()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / 20

from lsi import _C as _Cm
disp_stream = time_disp(r.desc.path) if r.desc.path == _Cm.LSI_PATH_STREAM else None
disp_tile = time_disp(_Cm.LSI_PATH_TILE)
indep_us = time_disp(r.desc.path, indep=True)
print(json.dumps({'workload': args.workload, 'batch': b_local,
                  'fwd_per_layer_us': indep_us, 'fwd_with_disp_us': disp_stream, 'fwd_with_disp_tile_us': disp_tile,
                  'lib': os.environ.get('LSI_HIP_LIB', ''),
                  'bwd_stream': os.environ.get('LSI_BWD_STREAM', '1'),
                  'rows': os.environ.get('LSI_BWD_STREAM_ROWS', ''),
                  'bwd_us': bwd, 'fwd_both_us': both[0], 'bwd_both_us': both[1],
                  'bwd_frac': bench.backward_bytes(nl, b_local, h, w) / (bwd * 1e-6) / 1e9 / bench.HBM_PEAK_GBPS}))
