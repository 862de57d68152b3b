"""Differential fuzz of the any-pose sweep kernel against the global-atomic
path over random shapes / poses / options.  python tools/fuzz_tile.py [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi.geometry import ldi
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(1234)
worst = 0.0
for it in range(n):
  nl = int(rs.choice([1, 2, 3, 5]))
  b = int(rs.choice([1, 2, 3, 7]))
  s = float(rs.choice([0.5, 1.0, 0.25]))
  q = int(round(1 / s))
  h = int(rs.choice([4, 12, 36, 64, 130, 256])) // q * q or q
  w = int(rs.choice([4, 20, 68, 132, 260, 520, 1100])) // q * q or q
  tex = torch.tensor(rs.rand(nl, b, h, w, 3).astype(np.float32), device=dev)
  disp = torch.tensor((0.05 + rs.rand(nl, b, h, w, 1) * rs.choice([0.3, 1.0])).astype(np.float32), device=dev)
  mask = torch.tensor(rs.rand(nl, b, h, w, 1).astype(np.float32), device=dev) if rs.rand() < 0.5 else None
  if rs.rand() < 0.3:  # planar storage
    tex = tex.permute(0, 1, 4, 2, 3).contiguous().permute(0, 1, 3, 4, 2)
  mats = []
  for _ in range(b):
    ang = rs.uniform(-0.3, 0.3, 3)
    cx, cy, cz = np.cos(ang); sx, sy, sz = np.sin(ang)
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @
         np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
         np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t = rs.uniform(-0.4, 0.4, 3)
    K = np.array([[0.8 * w, 0, w / 2], [0, 0.8 * w, h / 2], [0, 0, 1.0]])
    Kp = np.eye(4); Kp[:3, :3] = K
    Ki = np.eye(4); Ki[:3, :3] = np.linalg.inv(K)
    Rt = np.eye(4); Rt[:3, :3] = R; Rt[:3, 3] = t
    mats.append(Kp @ Rt @ Ki)
  mat = torch.tensor(np.stack(mats).astype(np.float32))
  compose = bool(rs.rand() < 0.6)
  want_disp = bool(rs.rand() < 0.5)
  kw = dict(compose_layers=compose, compute_trg_disp=want_disp, trg_downsampling=s,
            bg_layer_disp=0.05, max_disp=1.0, zbuf_scale=float(rs.choice([10., 50.])))
  a = ldi.forward_splat_matrix([tex, mask, disp], mat, path='tile', **kw)
  r = ldi.forward_splat_matrix([tex, mask, disp], mat, path='atomic', **kw)
  errs = []
  for k, (x, y) in enumerate(zip(a, r)):
    if x is None: continue
    x, y = x.double(), y.double()
    scale = 1.0 if k == 0 else float(y.abs().max()) + 1e-30
    errs.append(float((x - y).abs().max()) / scale)
  worst = max(worst, max(errs))
  flag = '' if max(errs) < 5e-5 else '   <-- MISMATCH'
  print('%2d L=%d B=%d %dx%d s=%.2f mask=%d compose=%d disp=%d  err %s%s' % (
      it, nl, b, h, w, s, mask is not None, compose, want_disp,
      ' '.join('%.1e' % e for e in errs), flag))
print('worst', worst)
