"""Device copy bandwidth reference (read + write traffic): what a kernel that
writes as much as it reads can expect from this part (backward: DESIGN.md 4.5)."""
import torch
dev = torch.device('cuda:0')
for mb in (512, 1024):
  n = mb * 1024 * 1024 // 4
  src = torch.rand(n, device=dev)
  dst = torch.empty_like(src)
  for _ in range(3):
    dst.copy_(src)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    dst.copy_(src)
  e1.record(); torch.cuda.synchronize()
  t = e0.elapsed_time(e1) / 20 * 1e-3
  print('copy %d MB: %.1f us, %.2f TB/s of read+write traffic' % (mb, t * 1e6, 2 * n * 4 / t / 1e12))
  # read-only and write-only
  e0.record()
  for _ in range(20):
    s = src.sum()
  e1.record(); torch.cuda.synchronize()
  t = e0.elapsed_time(e1) / 20 * 1e-3
  print('  sum (read only): %.2f TB/s' % (n * 4 / t / 1e12))
  e0.record()
  for _ in range(20):
    dst.fill_(1.0)
  e1.record(); torch.cuda.synchronize()
  t = e0.elapsed_time(e1) / 20 * 1e-3
  print('  fill (write only): %.2f TB/s' % (n * 4 / t / 1e12))
