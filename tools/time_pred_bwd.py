"""Times the two kernels of lsi_conv3x3_pred_bwd separately (data gradient,
weight + bias gradient) at the training step's head shape (8 x 256 x 768)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi import _C
dev = torch.device('cuda:0')
n, h, w = 8, 256, 768
x = torch.randn(n, 32, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
g = torch.randn(n, 4, h, w, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.rand(n, 4, h, w, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(4, 32, 3, 3, device=dev) * 0.1
gx = torch.empty(n, 32, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
gwb = torch.zeros(4 * 288 + 4, device=dev)
ws = torch.empty(_C.lib().lsi_conv3x3_pred_bwd_workspace_bytes(n, h, w) // 4, device=dev)
lib = _C.lib()
def run(a, b):
  rc = lib.lsi_conv3x3_pred_bwd(n, h, w, 4, _C.ptr(g), _C.ptr(y), _C.ptr(x), _C.ptr(wt),
                                _C.ptr(a), _C.ptr(b), _C.ptr(ws), ws.numel() * 4, _C.stream_ptr(dev))
  assert rc == 0, rc
for name, a, b in (('data', gx, None), ('weight', None, gwb)):
  for _ in range(3): run(a, b)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): run(a, b)
  e1.record(); torch.cuda.synchronize()
  print(name, '%.1f us' % (e0.elapsed_time(e1) * 1e3 / 20))
