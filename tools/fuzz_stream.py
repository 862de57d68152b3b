"""Differential fuzz of the STREAM kernel (all band decompositions / lock modes
the planner can pick, plus forced ones) against the global-atomic path over
random rectified configurations.  python tools/fuzz_stream.py [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi.geometry import ldi
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(4321)
worst = 0.0
for it in range(n):
  nl = int(rs.choice([1, 2, 4, 6]))
  b = int(rs.choice([1, 2, 4, 9]))
  s = float(rs.choice([0.5, 1.0, 0.25]))
  q = 4 * int(round(1 / s))
  h = int(rs.choice([2, 8, 36, 64, 130, 256, 384]))
  h = max(int(round(1 / s)), h // int(round(1 / s)) * int(round(1 / s)))
  w = max(q, int(rs.choice([8, 64, 256, 260, 520, 768, 1284])) // q * q)
  tex = torch.tensor(rs.rand(nl, b, h, w, 3).astype(np.float32), device=dev)
  dmax = float(rs.choice([0.4, 1.0]))
  dd = rs.rand(nl, b, h, w, 1) * dmax * rs.choice([0.6, 1.0, 1.3]) - rs.choice([0.0, 0.05])
  if rs.rand() < 0.5:  # smooth field instead of noise
    dd = np.broadcast_to((0.1 + 0.8 * rs.rand(nl, b, 1, 1, 1)) * dmax, dd.shape) + 0.02 * dd
  disp = torch.tensor(np.ascontiguousarray(dd).astype(np.float32), device=dev)
  mask = torch.tensor(rs.rand(nl, b, h, w, 1).astype(np.float32), device=dev) if rs.rand() < 0.35 else None
  if rs.rand() < 0.3:
    tex = tex.permute(0, 1, 4, 2, 3).contiguous().permute(0, 1, 3, 4, 2)
  mats = []
  for _ in range(b):
    f = rs.uniform(0.5, 1.0) * w
    K = np.array([[f, 0, w / 2], [0, f * rs.uniform(0.9, 1.1), h / 2], [0, 0, 1.0]])
    Kp = np.eye(4); Kp[:3, :3] = K
    Ki = np.eye(4); Ki[:3, :3] = np.linalg.inv(K)
    Rt = np.eye(4); Rt[0, 3] = rs.uniform(-0.6, 0.6); Rt[1, 3] = 0.0
    if rs.rand() < 0.3:
      Rt[2, 3] = 0.0
    mats.append(Kp @ Rt @ Ki)
  mat = torch.tensor(np.stack(mats).astype(np.float32))
  compose = bool(rs.rand() < 0.7)
  exp = int(rs.choice([0, 0, 1 << 16, 2 << 16, 1 << 18, 2 << 18, (2 << 16) | (2 << 18)]))
  det = bool(rs.rand() < 0.2)
  kw = dict(compose_layers=compose, trg_downsampling=s, bg_layer_disp=0.05 * dmax,
            max_disp=dmax, zbuf_scale=float(rs.choice([10., 50.])))
  try:
    a = ldi.forward_splat_matrix([tex, mask, disp], mat, path='stream', experiment=exp,
                                 deterministic=det, **kw)
  except RuntimeError as e:
    print('%2d stream not applicable: %s' % (it, str(e)[:60]))
    continue
  r = ldi.forward_splat_matrix([tex, mask, disp], mat, path='atomic', **kw)
  errs = []
  for k, (x, y) in enumerate(zip(a, r)):
    if x is None: continue
    x, y = x.double(), y.double()
    scale = 1.0 if k == 0 else float(y.abs().max()) + 1e-30
    errs.append(float((x - y).abs().max()) / scale)
  worst = max(worst, max(errs))
  flag = '' if max(errs) < 5e-5 else '   <-- MISMATCH'
  print('%2d L=%d B=%d %dx%d s=%.2f mask=%d compose=%d exp=%x det=%d  err %s%s' % (
      it, nl, b, h, w, s, mask is not None, compose, exp, det,
      ' '.join('%.1e' % e for e in errs), flag))
print('worst', worst)
