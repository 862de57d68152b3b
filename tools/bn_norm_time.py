import sys, os, ctypes, torch
sys.path.insert(0, '/root/repo/layered-scene-inference_amd')
from lsi import _C
from lsi.nnutils import _hip_bn
dev = torch.device('cuda:0'); lib = _C.lib()
shapes = [(8,32,128,384),(8,64,64,192),(8,128,32,96),(8,256,16,48),(8,512,8,24),(8,512,4,12),(8,512,2,6)]
for (n,c,h,w) in shapes:
  groups = 2
  x = torch.randn((n,c,h,w), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  y = torch.empty_like(x); beta = torch.zeros((c,), device=dev)
  npix = (n//groups)*h*w
  st = _C.stream_ptr(dev)
  ws = _hip_bn._workspace(dev, st, npix, c, 1, groups)
  mr = torch.empty((groups,2,c), device=dev)
  def two(): lib.lsi_bn_relu_fwd(x.data_ptr(), y.data_ptr(), beta.data_ptr(), ws.data_ptr(), mr.data_ptr(), npix, c, 1, 1, 1e-3, groups, _C.stream_ptr(dev))
  def one(): lib.lsi_bn_relu_norm(x.data_ptr(), y.data_ptr(), beta.data_ptr(), ws.data_ptr(), mr.data_ptr(), npix, c, 1, 1, 1e-3, groups, _C.stream_ptr(dev))
  res = []
  for fn in (two, one):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
      fn(); torch.cuda.synchronize()
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g, stream=s):
        for _ in range(40): fn()
      g.replay(); torch.cuda.synchronize()
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      e0.record(); 
      for _ in range(5): g.replay()
      e1.record(); torch.cuda.synchronize()
      res.append(e0.elapsed_time(e1)*1e3/200)
  print((n,c,h,w), 'two-pass %.1f us   norm-from-sums %.1f us' % tuple(res))
