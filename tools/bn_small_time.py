"""Per-call GPU time of the fused batch-norm kernels on the U-Net's map sizes
(batch 8 = both views, two groups), inside one captured graph of 40 dependent
calls: forward two-pass, normalise-from-sums, backward (statistics + dx) -- next
to a chain of 40 dependent one-element kernels (what a launch costs here)."""
import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi import _C
from lsi.nnutils import _hip_bn
dev = torch.device('cuda:0'); lib = _C.lib()


def timeit(fn, n=40):
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
      for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


one = torch.zeros((1,), device=dev)
print('dependent one-element kernels: %.2f us each' % timeit(lambda: one.add_(1.0)))
shapes = [(8, 32, 256, 768), (8, 32, 128, 384), (8, 64, 64, 192), (8, 128, 32, 96), (8, 256, 16, 48),
          (8, 512, 8, 24), (8, 512, 4, 12), (8, 512, 2, 6)]
for (n, c, h, w) in shapes:
  groups = 2
  x = torch.randn((n, c, h, w), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  dy = torch.randn_like(x)
  y = torch.empty_like(x); dx = torch.empty_like(x)
  beta = torch.zeros((c,), device=dev); dbeta = torch.empty((c,), device=dev)
  npix = (n // groups) * h * w
  ws = _hip_bn._workspace(dev, _C.stream_ptr(dev), npix, c, 1, groups)
  mr = torch.empty((groups, 2, c), device=dev)
  sp = lambda: _C.stream_ptr(dev)
  fwd = lambda: lib.lsi_bn_relu_fwd(x.data_ptr(), y.data_ptr(), beta.data_ptr(), ws.data_ptr(), mr.data_ptr(), npix, c, 1, 1, 1e-3, groups, sp())
  bwd = lambda: lib.lsi_bn_relu_bwd(x.data_ptr(), dy.data_ptr(), mr.data_ptr(), beta.data_ptr(), dx.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), npix, c, 1, 1, groups, sp())
  fwd(); torch.cuda.synchronize()
  mb = x.numel() * 2 / 1e6
  print('%-18s %6.1f MB  fwd (2 kernels) %6.1f us   bwd (2 kernels) %6.1f us' % ((n, c, h, w), mb, timeit(fwd), timeit(bwd)), flush=True)
