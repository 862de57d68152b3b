import os, sys
os.environ['LSI_HIP_LIB'] = 'hooks'
os.environ['LSI_SWEEP_BALANCE'] = os.environ.get('LSI_SWEEP_BALANCE', '0')
import numpy as np, torch
ROOT = '/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench
wl = 'cfg4'
nl, h, w, batch, per_gpu, cams, max_disp, bg = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
tex, disp, mat = bench.make_inputs(nl, batch, h, w, cams, max_disp, 1000, dev)
r = bench.Renderer(tex, disp, mat, max_disp, bg, 'tile')
r.desc.reserved = 4
base = 8 * 8 * batch * nl
r.ws = torch.zeros((base + 8 * 6 * 16 * 4096,), dtype=torch.uint8, device=dev)
r.ws_bytes = r.ws.numel()
for _ in range(3):
  r.launch()
torch.cuda.synchronize()
raw = r.ws[base:].view(torch.int64).view(-1, 6).cpu().numpy()
nwg = batch * 4
wg = raw[:nwg * 16].reshape(nwg, 16, 6)
loop = wg[:, :, 4].max(1).reshape(batch, 4)
prol = (wg[:, :, 5] & 0xffffffff).max(1).reshape(batch, 4)
# fraction of source pixels landing inside, exact
ys, xs = torch.meshgrid(torch.arange(h, device=dev) + 0.5, torch.arange(w, device=dev) + 0.5, indexing='ij')
M = mat.to(dev)
frac = []
for b in range(batch):
  f = 0
  for l in range(nl):
    dd = disp[l, b, :, :, 0]
    q = [M[b, j, 0] * xs + M[b, j, 1] * ys + M[b, j, 2] + M[b, j, 3] * dd for j in range(3)]
    X = q[0] / q[2] * 0.5 - 0.5; Y = q[1] / q[2] * 0.5 - 0.5
    inside = (X.floor() >= -1) & (X.floor() < w // 2) & (Y.floor() >= -1) & (Y.floor() < h // 2)
    f += inside.float().mean().item()
  frac.append(f / nl)
frac = np.array(frac)
tot = loop.sum(1)
print('corr(frac, element loop sum)', np.corrcoef(frac, tot)[0, 1])
A = np.stack([np.ones(batch), frac], 1)
co = np.linalg.lstsq(A, tot, rcond=None)[0]
print('fit loop_sum = %.0f + %.0f * frac; frac range %.2f..%.2f; resid rel %.3f' % (co[0], co[1], frac.min(), frac.max(), np.std(tot - A @ co) / tot.mean()))
print('per-WG loop max: min %d med %d max %d; per-element mean-of-4: min %d med %d max %d' % (loop.min(), np.median(loop), loop.max(), (tot/4).min(), np.median(tot/4), (tot/4).max()))
print('within element max/mean of the 4 tiles: med %.3f max %.3f' % (np.median(loop.max(1) / loop.mean(1)), (loop.max(1) / loop.mean(1)).max()))
print('prologue cycles+: med %d max %d' % (np.median(prol), prol.max()))
o = np.argsort(tot)[::-1]
for b in list(o[:4]) + list(o[-3:]):
  print(b, 'frac %.3f' % frac[b], loop[b].tolist())
