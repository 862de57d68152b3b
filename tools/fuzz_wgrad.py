"""Random shapes through lsi_conv3x3_wgrad and the prediction head's
lsi_conv3x3_pred_bwd (weight + bias gradient, data gradient) against fp32 torch
references of the same bf16 operands.   python tools/fuzz_wgrad.py [n] [seed]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi import _C
dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
lib = _C.lib()
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
worst = dict(wgrad=0.0, pred_w=0.0, pred_b=0.0, pred_x=0.0)
for it in range(n_cases):
  n = int(rs.randint(1, 4)); h = int(rs.randint(1, 80)); w = int(rs.randint(1, 210))
  cin = int(rs.choice([32, 64, 96, 128])); cout = int(rs.choice([32, 64, 96, 128]))
  g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
  x = cl(torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16))
  gy = cl(torch.randn((n, cout, h, w), generator=g).to(dev).to(torch.bfloat16))
  nb = lib.lsi_conv3x3_wgrad_workspace_bytes(n, h, w, cin, cout)
  ws = torch.empty((nb // 4 + 4,), device=dev)
  gw = torch.full((cout, cin, 3, 3), float('nan'), device=dev)
  rc = lib.lsi_conv3x3_wgrad(n, h, w, cin, cout, _C.ptr(x), _C.ptr(gy), _C.ptr(gw), _C.ptr(ws), nb,
                             _C.stream_ptr(dev))
  assert rc == 0, (rc, n, h, w, cin, cout)
  want = torch.nn.grad.conv2d_weight(x.float(), (cout, cin, 3, 3), gy.float(), padding=1)
  e = float((gw - want).abs().max()) / max(float(want.abs().max()), 1e-6)
  assert e <= 2e-5, ('wgrad', e, n, h, w, cin, cout)
  worst['wgrad'] = max(worst['wgrad'], e)
  # the prediction head: width a multiple of 16, 1 .. 4 outputs
  w16 = 16 * int(rs.randint(1, 14)); co = int(rs.randint(1, 5))
  x = cl(torch.randn((n, 32, h, w16), generator=g).to(dev).to(torch.bfloat16))
  gg = cl(torch.randn((n, 4, h, w16), generator=g).to(dev))
  yy = cl(torch.rand((n, 4, h, w16), generator=g).to(dev))
  wt = (torch.randn((co, 32, 3, 3), generator=g) * 0.1).to(dev)
  nb = lib.lsi_conv3x3_pred_bwd_workspace_bytes(n, h, w16)
  ws = torch.empty((nb // 4 + 4,), device=dev)
  gwb = torch.full((co * 288 + co,), float('nan'), device=dev)
  gx = cl(torch.full((n, 32, h, w16), float('nan'), device=dev, dtype=torch.bfloat16))
  rc = lib.lsi_conv3x3_pred_bwd(n, h, w16, co, _C.ptr(gg), _C.ptr(yy), _C.ptr(x), _C.ptr(wt), _C.ptr(gx),
                                _C.ptr(gwb), _C.ptr(ws), nb, _C.stream_ptr(dev))
  assert rc == 0, (rc, n, h, w16, co)
  gz = (gg * yy * (1 - yy))[:, :co]
  want_w = torch.nn.grad.conv2d_weight(x.float(), (co, 32, 3, 3), gz, padding=1)
  want_b = gz.sum(dim=(0, 2, 3))
  want_x = torch.nn.grad.conv2d_input((n, 32, h, w16), wt.to(torch.bfloat16).float(),
                                      gz.to(torch.bfloat16).float(), padding=1)
  # (the bias gradient is a sum with cancellation: measured against sum |gz|)
  sw = max(float(want_w.abs().max()), 1e-6); sb = max(float(gz.abs().sum(dim=(0, 2, 3)).max()), 1e-6)
  sx = max(float(want_x.abs().max()), 1e-6)
  ew = float((gwb[:co * 288].view(co, 32, 3, 3) - want_w).abs().max()) / sw
  eb = float((gwb[co * 288:] - want_b).abs().max()) / sb
  ex = float((gx.float() - want_x).abs().max()) / sx
  assert ew <= 1e-4 and eb <= 1e-5 and ex <= 2.0 ** -7, ('pred', ew, eb, ex, n, h, w16, co)
  worst['pred_w'] = max(worst['pred_w'], ew); worst['pred_b'] = max(worst['pred_b'], eb)
  worst['pred_x'] = max(worst['pred_x'], ex)
print('fuzz_wgrad: %d cases ok; worst relative errors %s' % (n_cases, worst))
