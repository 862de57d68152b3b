"""Per-shape timing of the fused BN+ReLU kernels (csrc/lsi_bn.hip) on the U-Net's
activation shapes at batch 4, 256 x 768: microseconds and GB/s of the forward
(two passes: 2 reads + 1 write of the activation) and the backward (two passes:
4 reads + 1 write).   python tools/bn_bench.py [--bf16 true]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
from lsi.nnutils import _hip_bn

ap = argparse.ArgumentParser()
ap.add_argument('--bf16', default='true')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--groups', type=int, default=2)
args = ap.parse_args()
dt = torch.bfloat16 if args.bf16 == 'true' else torch.float32
es = 2 if dt == torch.bfloat16 else 4
dev = torch.device('cuda', 0)
SHAPES = [(32, 256, 768), (32, 128, 384), (64, 128, 384), (64, 64, 192), (128, 64, 192),
          (128, 32, 96), (256, 16, 48), (512, 8, 24), (512, 4, 12), (512, 2, 6)]
rows = []
for c, h, w in SHAPES:
  # rotate over enough tensors to exceed the Infinity Cache
  nbuf = max(2, int(600e6 // (args.batch * c * h * w * es)) + 1)
  nbuf = min(nbuf, 64)
  xs = [torch.randn(args.batch, c, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(nbuf)]
  beta = torch.zeros(c, device=dev, requires_grad=True)
  gy = torch.randn(args.batch, c, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
  def fwd(i):
    return _hip_bn.batch_norm_relu(xs[i % nbuf], beta, 1e-3, True, args.groups)
  iters = 20
  side = torch.cuda.Stream(dev)
  side.wait_stream(torch.cuda.current_stream(dev))
  with torch.cuda.stream(side):
    for i in range(3):
      fwd(i).backward(gy)
    side.synchronize()
    # GPU time only: the calls captured into HIP graphs (eager calls are
    # host-bound at these sizes)
    g_f, g_fb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_f, stream=side):
      for i in range(iters):
        fwd(i)
    with torch.cuda.graph(g_fb, stream=side):
      for i in range(iters):
        fwd(i).backward(gy)
    times = []
    for g in (g_f, g_fb):
      g.replay(); side.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(side)
      for _ in range(5):
        g.replay()
      e1.record(side)
      side.synchronize()
      times.append(e0.elapsed_time(e1) * 1e3 / (5 * iters))
  tf, tb = times[0], times[1] - times[0]
  nb = args.batch * c * h * w * es
  rows.append({'C': c, 'hw': [h, w], 'MB': round(nb / 1e6, 2), 'fwd_us': round(tf, 2), 'bwd_us': round(tb, 2),
               'fwd_GBps': round(3 * nb / tf / 1e3), 'bwd_GBps': round(5 * nb / tb / 1e3)})
  print(json.dumps(rows[-1]))
