"""lsi_conv3x3_wgrad against aten.convolution_backward (weight gradient only)
at the LDI heads' shapes: time and largest difference.
  python tools/time_wgrad.py [n]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi import _C
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
quick = len(sys.argv) > 2  # second argument: the two training shapes only, no reference
lib = _C.lib()
SHAPES = ((32, 32, 256, 768), (96, 64, 128, 384), (192, 128, 64, 192), (256, 128, 32, 96),
          (128, 128, 32, 96), (512, 256, 16, 48), (32, 96, 70, 100))
for cin, cout, h, w in (SHAPES[:2] if quick else SHAPES):
  x = torch.randn(n, cin, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  g = torch.randn(n, cout, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  gw = torch.empty(cout, cin, 3, 3, device=dev)
  nb = lib.lsi_conv3x3_wgrad_workspace_bytes(n, h, w, cin, cout)
  ws = torch.empty(nb // 4, device=dev)
  def own():
    rc = lib.lsi_conv3x3_wgrad(n, h, w, cin, cout, _C.ptr(x), _C.ptr(g), _C.ptr(gw), _C.ptr(ws),
                               nb, _C.stream_ptr(dev))
    assert rc == 0, rc
  wt = torch.zeros(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  def aten():
    return torch.ops.aten.convolution_backward(g, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1]
  own()
  if quick:
    err = scale = float('nan')
  else:
    want = torch.nn.grad.conv2d_weight(x.float(), (cout, cin, 3, 3), g.float(), padding=1)
    err = float((gw - want).abs().max()); scale = float(want.abs().max())
  res = []
  for fn in (own, aten):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 100)
  fl = 2.0 * n * h * w * cin * cout * 9
  print('%d -> %d  %dx%dx%d: own %.1f us (%.0f TFLOP/s)  aten %.1f us  max err %.3g of %.3g  workspace %.1f MB'
        % (cin, cout, n, h, w, res[0], fl / res[0] * 1e-6, res[1], err, scale, nb / 1e6))
