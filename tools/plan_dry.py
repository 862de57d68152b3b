"""Prints the STREAM planner's choice for the bench workloads without a GPU
(the launch itself then fails: no device here).  LSI_STREAM_VERBOSE is set."""
import ctypes, os, sys
os.environ['LSI_STREAM_VERBOSE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import numpy as np, torch
from lsi import _C
from lsi.geometry import projection
lib = ctypes.CDLL(_C.SO_PATH)
for name, (nl, b, h, w) in {'cfg3': (4, 32, 256, 768), 'cfg3/2': (4, 16, 256, 768),
                            'cfg3/4': (4, 8, 256, 768), 'cfg3/8': (4, 4, 256, 768),
                            'cfg2': (2, 4, 256, 768), 'cfg5': (4, 8, 512, 1536),
                            'cfg5/8': (4, 1, 512, 1536), 'B64': (4, 64, 256, 768),
                            'tiny': (1, 1, 64, 256)}.items():
  d = _C.LsiSplatDesc()
  d.L, d.B, d.H, d.W, d.Ht, d.Wt = nl, b, h, w, h // 2, w // 2
  d.tex_sl, d.tex_sb, d.tex_sy, d.tex_sx, d.tex_sc = b * h * w * 3, h * w * 3, w * 3, 3, 1
  d.disp_sl, d.disp_sb, d.disp_sy, d.disp_sx = b * h * w, h * w, w, 1
  d.trg_downsampling, d.max_disp, d.zbuf_scale, d.bg_wt = 0.5, 0.4, 50.0, 1e-11
  d.flags = _C.LSI_COMPOSE | _C.LSI_WS_KEEP
  d.path = _C.LSI_PATH_STREAM
  if len(sys.argv) > 1: d.tune_rows = int(sys.argv[1])
  if len(sys.argv) > 2: d.tune_threads = int(sys.argv[2])
  k = torch.tensor([[0.58 * w, 0, w / 2.0], [0, 0.58 * w, h / 2.0], [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
  rot = torch.eye(3).expand(b, 3, 3).contiguous()
  t = torch.tensor([[-0.532], [0.0], [0.0]]).expand(b, 3, 1).contiguous()
  mat = projection.forward_projection_matrix(k, k, rot, t).contiguous()
  lib.lsi_stream_ok.restype = ctypes.c_int
  d.tune_window = lib.lsi_stream_ok(ctypes.byref(d), ctypes.c_void_p(mat.data_ptr()))
  fake = ctypes.c_void_p(1 << 24)
  sys.stderr.write('%-7s ' % name); sys.stderr.flush()
  lib.lsi_splat_fwd.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_size_t, ctypes.c_void_p]
  rc = lib.lsi_splat_fwd(ctypes.byref(d), fake, fake, None, fake, fake, fake, None, fake, 1 << 30, None)
