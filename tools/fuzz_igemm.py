"""Random shapes through the implicit-GEMM convolution kernels (lsi_conv2d_fwd /
_bwd_data / _wgrad and the transposed convolution built on them) against fp32
torch references on the same bf16-rounded operands.

  python tools/fuzz_igemm.py [n_cases] [seed]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import numpy as np
import torch
import torch.nn.functional as F
from lsi.nnutils import _hip_conv

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rs = np.random.RandomState(seed)
dev = torch.device('cuda:0')
_hip_conv.IGEMM_WGRAD_MIN_PIXELS = 0
_hip_conv.WGRAD_MIN_PIXELS = 1 << 30
worst = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0, 'convt': 0.0, 'convt_dgrad': 0.0, 'convt_wgrad': 0.0}
own_w = 0


def same(size, k, s):
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return total // 2, total - total // 2, out


def cl(t):
  return t.contiguous(memory_format=torch.channels_last)


for it in range(n_cases):
  n = int(rs.randint(1, 4))
  cin = 32 * int(rs.randint(1, 6))
  cout = 32 * int(rs.randint(1, 6))
  h, w = int(rs.randint(1, 71)), int(rs.randint(1, 71))
  g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
  if rs.rand() < 0.3:
    # transposed convolution, k = 4, stride 2, padding 1 (nets.py:100-103)
    h, w = max(1, h // 2), max(1, w // 2)
    x = cl(torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
    wt = (torch.randn((cin, cout, 4, 4), generator=g) * (2.0 / (cin * 4)) ** 0.5).to(dev).requires_grad_(True)
    got = _hip_conv.conv_transpose2d(x, wt)
    xr = x.detach().float().requires_grad_(True)
    wr = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
    want = F.conv_transpose2d(xr, wr, None, 2, 1)
    tag = ('convt', n, cin, cout, h, w)
    keys = ('convt', 'convt_dgrad', 'convt_wgrad')
    d = _hip_conv._conv_desc(n, 2 * h, 2 * w, cout, h, w, cin, 4, 4, 2, 1, 1)
  else:
    k = int(rs.choice([1, 2, 3, 3, 3, 4, 5, 5, 6, 7]))
    s = int(rs.choice([1, 1, 2]))
    x = cl(torch.randn((n, cin, h, w), generator=g).to(dev).to(torch.bfloat16)).requires_grad_(True)
    wt = (torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev).requires_grad_(True)
    pt, pb, oh = same(h, k, s)
    pl, pr, ow = same(w, k, s)
    got = _hip_conv.conv2d(x, wt, s, pt, pl, oh, ow)
    xr = x.detach().float().requires_grad_(True)
    wr = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
    want = F.conv2d(F.pad(xr, (pl, pr, pt, pb)), wr, None, s)
    tag = ('conv', n, cin, cout, h, w, k, s)
    keys = ('fwd', 'dgrad', 'wgrad')
    d = _hip_conv._conv_desc(n, h, w, cin, oh, ow, cout, k, k, s, pt, pl)
  assert got.shape == want.shape, (tag, got.shape, want.shape)
  err = (got.float() - want).abs()
  e0 = float((err - want.abs() * 2.0 ** -8).max())
  assert e0 <= 2e-3, (tag, 'forward', float(err.max()))
  worst[keys[0]] = max(worst[keys[0]], float(err.max() / (want.abs().max() + 1e-20)))
  c = torch.randn(want.shape, generator=g).to(dev).to(torch.bfloat16)
  (got.float() * c.float()).sum().backward()
  (want * c.float()).sum().backward()
  sc = float(xr.grad.abs().max()) + 1e-20
  e1 = float((x.grad.float() - xr.grad).abs().max()) / sc
  assert e1 <= 2.0 ** -7 + 1e-6, (tag, 'data gradient', e1)
  worst[keys[1]] = max(worst[keys[1]], e1)
  own = _hip_conv._igemm_wgrad_bytes(d) > 0
  own_w += int(own)
  sw = float(wr.grad.abs().max()) + 1e-20
  e2 = float((wt.grad - wr.grad).abs().max()) / sw
  assert e2 <= (1e-4 if own else 2e-2), (tag, 'weight gradient', e2, own)
  if own:
    worst[keys[2]] = max(worst[keys[2]], e2)
print('fuzz_igemm: %d cases ok (%d weight gradients on lsi_conv2d_wgrad); worst relative errors %s'
      % (n_cases, own_w, {k: float('%.3g' % v) for k, v in worst.items()}))
