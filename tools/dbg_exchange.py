import sys, os, torch
sys.path.insert(0, '/root/repo/layered-scene-inference_amd'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
from lsi.geometry import ldi, projection
dev = torch.device('cuda:0')
gen = torch.Generator(device='cpu').manual_seed(11)
nl, b, h, w = 4, 4, 256, 768
tex = torch.rand(nl, b, h, w, 3, generator=gen).to(dev)
disp = (0.4 * torch.rand(nl, b, h, w, 1, generator=gen)).to(dev)
k = torch.tensor([[0.58 * w, 0, w / 2], [0, 0.58 * w, h / 2], [0, 0, 1.0]]).expand(b, 3, 3)
eye, t = torch.eye(3).expand(b, 3, 3), torch.tensor([[-0.532], [0], [0]]).expand(b, 3, 1)
mat = projection.forward_projection_matrix(k, k, eye, t)
kw = dict(trg_downsampling=0.5, bg_layer_disp=1e-3, max_disp=0.4, zbuf_scale=50)
ref, rw = ldi.forward_splat_matrix([tex, None, disp], mat, path='atomic', **kw)
for br in (0, 1, 2, 4, 8):
  img, wts = ldi.forward_splat_matrix([tex, None, disp], mat, path='stream', band_rows=br, **kw)
  bad = ((img - ref).abs() > 1e-4).any(-1)[0]
  idx = bad.nonzero()
  print('band_rows', br, 'bad cells', idx.shape[0])
  if idx.shape[0]:
    print(' batches', idx[:, 0].unique().tolist(), 'rows', idx[:, 1].unique().tolist()[:40], 'cols', idx[:, 2].min().item(), idx[:, 2].max().item())
    i = idx[0]; print(' first', i.tolist(), img[0, i[0], i[1], i[2]].tolist(), ref[0, i[0], i[1], i[2]].tolist(), wts[0, i[0], i[1], i[2]].item(), rw[0, i[0], i[1], i[2]].item())
