"""Every convolution of the bf16 training step (batch 8 = both views, 256 x 768)
timed alone on the implicit-GEMM kernel and on aten (MIOpen): forward, data
gradient; microseconds, TFLOP/s, share of the 2.5 PF bf16 peak.

  python tools/conv_bench.py [--n 8] [--h 256] [--w 768] [--iters 20]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import torch
import torch.nn.functional as F
from lsi.nnutils import _hip_conv

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=8)
ap.add_argument('--h', type=int, default=256)
ap.add_argument('--w', type=int, default=768)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--out', default='')
args = ap.parse_args()
dev = torch.device('cuda:0')
N, H, W = args.n, args.h, args.w


def same(size, k, s):
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return total // 2, total - total // 2, out


def timeit(fn, iters=args.iters):
  """GPU time per call: `iters` calls captured in one HIP graph (no host gaps)."""
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  st = torch.cuda.Stream()
  st.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
      for _ in range(iters):
        fn()
  torch.cuda.synchronize()
  g.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  g.replay()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / iters


# (name, kind, cin, cout, k, stride, input h, input w)
L = []
h, w, cin = H, W, 3
for name, cout, k, s in [('cnv1', 32, 7, 2), ('cnv1b', 32, 7, 1), ('cnv2', 64, 5, 2),
                         ('cnv2b', 64, 5, 1), ('cnv3', 128, 3, 2), ('cnv3b', 128, 3, 1),
                         ('cnv4', 256, 3, 2), ('cnv4b', 256, 3, 1), ('cnv5', 512, 3, 2),
                         ('cnv5b', 512, 3, 1), ('cnv6', 512, 3, 2), ('cnv6b', 512, 3, 1),
                         ('cnv7', 512, 3, 2), ('cnv7b', 512, 3, 1)]:
  L.append((name, 'conv', cin, cout, k, s, h, w))
  h, w, cin = -(-h // s), -(-w // s), cout
for name, ci, co, skip in [('7', 512, 512, 512), ('6', 512, 512, 512), ('5', 512, 256, 256),
                           ('4', 256, 128, 128)]:
  L.append(('upcnv' + name, 'convt', ci, co, 4, 2, h, w))
  h, w = 2 * h, 2 * w
  L.append(('icnv' + name, 'conv', co + skip, co, 3, 1, h, w))
for name, ci, co, skip in [('3', 128, 128, 64), ('2', 128, 64, 32), ('1', 64, 32, 0)]:
  L.append(('head.upcnv' + name, 'convt', ci, co, 4, 2, h, w))
  h, w = 2 * h, 2 * w
  L.append(('head.upcnv%sb' % name, 'conv', co + skip, co, 3, 1, h, w))

rows = []
tot = {'own_f': 0., 'lib_f': 0., 'own_d': 0., 'lib_d': 0., 'flop': 0.}
for name, kind, cin, cout, k, s, h, w in L:
  if cin % 32 or cout % 32:
    continue
  g = torch.Generator().manual_seed(1)
  x = torch.randn((N, cin, h, w), generator=g).to(dev).to(torch.bfloat16).contiguous(
      memory_format=torch.channels_last)
  if kind == 'conv':
    wt = (torch.randn((cout, cin, k, k), generator=g) * 0.05).to(dev)
    pt, pb, oh = same(h, k, s)
    pl, pr, ow = same(w, k, s)
    flop = 2.0 * N * oh * ow * cin * cout * k * k
    wb = wt.to(torch.bfloat16)
    sym = pt == pb and pl == pr
    own_f = lambda: _hip_conv.conv2d(x, wt, s, pt, pl, oh, ow)
    if sym:
      lib_f = lambda: F.conv2d(x, wb, None, s, (pt, pl))
    else:
      lib_f = lambda: F.conv2d(F.pad(x, (pl, pr, pt, pb)), wb, None, s)
    y = own_f()
    gy = torch.randn_like(y)
    d = _hip_conv._conv_desc(N, h, w, cin, oh, ow, cout, k, k, s, pt, pl)
    own_d = lambda: _hip_conv._igemm('lsi_conv2d_bwd_data', d, gy, wt, torch.empty_like(x))
    xp = x if sym else F.pad(x, (pl, pr, pt, pb))
    lib_d = lambda: torch.ops.aten.convolution_backward(
        gy, xp, wb, None, [s, s], [pt, pl] if sym else [0, 0], [1, 1], False, [0, 0], 1,
        [True, False, False])
    ref = lib_f().float()
    if k == 3 and s == 1 and _hip_conv.wgrad_supported(x, cin, cout, 3, 1):
      own_w = lambda: _hip_conv._weight_grad(x, gy, wt)
    elif _hip_conv._igemm_wgrad_bytes(d) > 0:
      own_w = lambda: _hip_conv._igemm_wgrad(d, x, gy, wt)
    else:
      own_w = None
    lib_w = lambda: torch.ops.aten.convolution_backward(
        gy, xp, wb, None, [s, s], [pt, pl] if sym else [0, 0], [1, 1], False, [0, 0], 1,
        [False, True, False])
    if own_w is not None:
      werr = float((own_w().float() - lib_w()[1].float()).abs().max() / (lib_w()[1].float().abs().max() + 1e-20))
  else:
    wt = (torch.randn((cin, cout, 4, 4), generator=g) * 0.05).to(dev)
    flop = 2.0 * N * (2 * h) * (2 * w) * cin * cout * 4
    wb = wt.to(torch.bfloat16)
    own_f = lambda: _hip_conv.conv_transpose2d(x, wt)
    lib_f = lambda: F.conv_transpose2d(x, wb, None, 2, 1)
    y = own_f()
    gy = torch.randn_like(y)
    d = _hip_conv._conv_desc(N, 2 * h, 2 * w, cout, h, w, cin, 4, 4, 2, 1, 1)
    own_d = lambda: _hip_conv._igemm('lsi_conv2d_fwd', d, gy, wt, torch.empty_like(x))
    lib_d = lambda: torch.ops.aten.convolution_backward(
        gy, x, wb, None, [2, 2], [1, 1], [1, 1], True, [0, 0], 1, [True, False, False])
    ref = lib_f().float()
    own_w = (lambda: _hip_conv._igemm_wgrad(d, gy, x, wt)) if _hip_conv._igemm_wgrad_bytes(d) > 0 else None
    lib_w = lambda: torch.ops.aten.convolution_backward(
        gy, x, wb, None, [2, 2], [1, 1], [1, 1], True, [0, 0], 1, [False, True, False])
    if own_w is not None:
      werr = float((own_w().float() - lib_w()[1].float()).abs().max() / (lib_w()[1].float().abs().max() + 1e-20))
  err = float((y.float() - ref).abs().max())
  r = dict(name=name, kind=kind, cin=cin, cout=cout, k=k, stride=s, h=h, w=w, gflop=flop / 1e9,
           own_fwd_us=timeit(own_f), lib_fwd_us=timeit(lib_f), own_dgrad_us=timeit(own_d),
           lib_dgrad_us=timeit(lib_d), max_diff_vs_lib=err)
  mult = 2 if name.startswith('head.') else 1
  r['lib_wgrad_us'] = timeit(lib_w)
  r['own_wgrad_us'] = timeit(own_w) if own_w is not None else float('nan')
  r['wgrad_rel_diff_vs_lib'] = werr if own_w is not None else float('nan')
  r['own_fwd_util'] = flop / (r['own_fwd_us'] * 1e-6) / 2.5e15
  r['own_dgrad_util'] = flop / (r['own_dgrad_us'] * 1e-6) / 2.5e15
  rows.append(r)
  tot['own_f'] += mult * r['own_fwd_us']; tot['lib_f'] += mult * r['lib_fwd_us']
  tot['own_d'] += mult * r['own_dgrad_us']; tot['lib_d'] += mult * r['lib_dgrad_us']
  tot['flop'] += mult * flop
  tot['own_w'] = tot.get('own_w', 0.) + mult * (r['own_wgrad_us'] if own_w is not None else r['lib_wgrad_us'])
  tot['lib_w'] = tot.get('lib_w', 0.) + mult * r['lib_wgrad_us']
  print('%-14s %-5s %4d->%-4d k%d s%d %3dx%-3d %7.1f GF | fwd own %7.1f us (%4.1f%%) lib %7.1f | dgrad own %7.1f (%4.1f%%) lib %7.1f | wgrad own %7.1f lib %7.1f (rel diff %.1e) | diff %.3g'
        % (name, kind, cin, cout, k, s, h, w, flop / 1e9, r['own_fwd_us'], 100 * r['own_fwd_util'],
           r['lib_fwd_us'], r['own_dgrad_us'], 100 * r['own_dgrad_util'], r['lib_dgrad_us'],
           r['own_wgrad_us'], r['lib_wgrad_us'], r['wgrad_rel_diff_vs_lib'], err),
        flush=True)
print('total (2 heads): fwd own %.0f us lib %.0f us | dgrad own %.0f lib %.0f | wgrad own(+lib where not taken) %.0f lib %.0f | %.1f GFLOP; fwd util own %.1f%% lib %.1f%%'
      % (tot['own_f'], tot['lib_f'], tot['own_d'], tot['lib_d'], tot['own_w'], tot['lib_w'], tot['flop'] / 1e9,
         100 * tot['flop'] / (tot['own_f'] * 1e-6) / 2.5e15,
         100 * tot['flop'] / (tot['lib_f'] * 1e-6) / 2.5e15))
if args.out:
  json.dump({'layers': rows, 'total': tot}, open(args.out, 'w'), indent=1)
