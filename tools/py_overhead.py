"""Host cost per call of the ctypes-bound ops on small tensors (the GPU side is
a few microseconds: the wall clock per call is the Python + launch overhead).
  python tools/py_overhead.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get('LSI_PKG_ROOT') or os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi.nnutils import _hip_bn, _hip_conv
import torch.nn.functional as F
dev = torch.device('cuda:0')
x = torch.randn(8, 32, 16, 32, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
beta = torch.zeros(32, device=dev)
w = torch.randn(32, 32, 3, 3, device=dev) * 0.1
wb = w.to(torch.bfloat16)
def timeit(name, fn, n=2000):
  for _ in range(50): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n): fn()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print('%-34s host %.1f us per call (with the final sync %.1f)' % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
with torch.no_grad():
  timeit('bn_relu forward (2 launches)', lambda: _hip_bn.batch_norm_relu(x, beta, 1e-3, True, 2))
  timeit('conv3x3_c32 forward (1 launch)', lambda: _hip_conv.conv3x3_c32(x, w))
  timeit('F.conv2d bf16 (aten)', lambda: F.conv2d(x, wb, None, 1, 1))
  timeit('torch relu (1 launch)', lambda: torch.relu(x))
xg = x.clone().requires_grad_(True); bg = beta.clone().requires_grad_(True)
g = torch.randn_like(x)
def fb():
  y = _hip_bn.batch_norm_relu(xg, bg, 1e-3, True, 2)
  y.backward(g)
  xg.grad = None; bg.grad = None
timeit('bn_relu forward + backward', fb, 1000)
wg = w.clone().requires_grad_(True)
def fb2():
  y = _hip_conv.conv3x3_c32(xg, wg)
  y.backward(g)
  xg.grad = None; wg.grad = None
timeit('conv3x3_c32 forward + backward', fb2, 1000)
