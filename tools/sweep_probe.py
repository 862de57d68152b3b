"""Section totals of the sweep kernel (instrumented build, reserved bit 2):
cycles per wave in issue / projection+weights / left pair / right pair / loop /
prologue.  python build.py --hooks first."""
import os, sys
os.environ['LSI_HIP_LIB'] = 'hooks'
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nl, h, w, batch, per_gpu, cams, max_disp, bg = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
tex, disp, mat = bench.make_inputs(nl, batch, h, w, cams, max_disp, 1000, dev)
r = bench.Renderer(tex, disp, mat, max_disp, bg, 'tile')
r.desc.reserved = 4 | flags
base = 8 * 8 * batch * nl
r.ws = torch.zeros((base + 8 * 6 * 16 * 4096,), dtype=torch.uint8, device=dev)
r.ws_bytes = r.ws.numel()
for _ in range(3):
  r.launch()
torch.cuda.synchronize()
t = r.ws[base:].view(torch.int64).view(-1, 6).cpu().numpy()
t = t[t.sum(1) != 0]
names = ['issue', 'project', 'left', 'right', 'loop', 'prologue']
print('waves', len(t))
for q, nm in ((50, 'median'), (95, 'p95'), (100, 'max')):
  print(nm, {n: int(np.percentile(t[:, i], q)) for i, n in enumerate(names)})
raw = r.ws[base:].view(torch.int64).view(-1, 6).cpu().numpy()
nwg = len(t) // 16
wg = raw[:nwg * 16].reshape(nwg, 16, 6)
loop = wg[:, :, 4]
wmax = loop.max(1); wmed = np.median(loop, 1); wmin = loop.min(1)
print('per-WG max loop: min %d median %d p90 %d max %d' % (wmax.min(), np.median(wmax), np.percentile(wmax, 90), wmax.max()))
print('within-WG spread (max/median of waves): median %.2f max %.2f' % (np.median(wmax / np.maximum(wmed, 1)), (wmax / np.maximum(wmed, 1)).max()))
order = np.argsort(wmax)[::-1][:8]
for i in order[:3]:
  print('WG', i, 'b', i // 4, 'tile', i % 4, 'waves loop', loop[i].tolist())
  print('   px-iters', (wg[i, :, 5] & 0xffffffff).tolist())
  print('   ok lanes', (wg[i, :, 5] >> 32).tolist())
