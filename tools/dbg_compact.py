import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_cpu
import test_splat_gpu as T
from lsi.geometry import ldi
dev = torch.device('cuda:0')
nl, b, h, w = 4, 1, 24, 768
rs = np.random.RandomState(nl * 1000 + w + h)
tex, disp, mat = T._stream_case(rs, nl, b, h, w, 'smooth')
m03 = float(mat[0, 0, 3])
xx = np.arange(w, dtype=np.float32)
frac = rs.choice(np.array([0.004, 0.0040001, 0.0039999, 0.001333, 0.0013334, 0.0013333, 0.996, 0.9960001, 0.5, 0.25], np.float32), size=(nl, b, h, w))
want_x = np.floor(xx / 2)[None, None, None, :] + frac - 3.0
disp = (((want_x + 0.5) * 2.0 - xx - 0.5) / m03)[..., None].astype(np.float32)
disp = np.clip(disp, 1e-4, 0.5).astype(np.float32)
want = ref_cpu.forward_splat(tex, None, disp, mat, 0.5, 1e-3, 0.4, 50, True)
ldi_src = [torch.tensor(tex, device=dev), None, torch.tensor(disp, device=dev)]
def run(**kw):
  return ldi.forward_splat_matrix(ldi_src, torch.tensor(mat), compose_layers=True, trg_downsampling=0.5,
                                  bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50, path='stream', **kw)
for rows in (0, 1, 2, 4, 16):
  for locks in (1, 2):
    for sub in (0, 1, 2):
      img, wts = run(band_rows=rows, experiment=(locks << 18) | (sub << 12))
      e = np.abs(img.cpu().numpy() - want['img'])
      ew = np.abs(wts.cpu().numpy() - want['wts']) / want['wts']
      bad = np.argwhere(e > 2e-5)
      print('rows', rows, 'locks', locks, 'sub', sub, 'img err %.2e' % e.max(), 'wts rel %.2e' % ew.max(),
            'nbad', len(bad), 'first', bad[:3].tolist())
img, wts = run(experiment=0x40000000)
print('general kernel img err %.2e' % np.abs(img.cpu().numpy() - want['img']).max())
