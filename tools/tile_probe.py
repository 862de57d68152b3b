"""Phase totals of the tile kernel (instrumented build): ticks per workgroup in
A (project+bin), barrier, B (gather), barrier.  python build.py --hooks first."""
import ctypes, os, sys
os.environ['LSI_HIP_LIB'] = 'hooks'
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench
from lsi import _C
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nl, h, w, batch, per_gpu, cams, max_disp, bg = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
tex, disp, mat = bench.make_inputs(nl, batch, h, w, cams, max_disp, 1000, dev)
r = bench.Renderer(tex, disp, mat, max_disp, bg, 'tile', band_rows=rows)
r.desc.reserved = 4
r.ws = torch.zeros((max(r.ws_bytes, 8 * batch + 8 * 4 * 65536),), dtype=torch.uint8, device=dev)
r.ws_bytes = r.ws.numel()
for _ in range(3):
  r.launch()
torch.cuda.synchronize()
t = r.ws[8 * batch:].view(torch.int64).view(-1, 4).cpu().numpy()
t = t[t.sum(1) != 0]
print('workgroups', len(t), 'median ticks A / barrier / B / barrier:', np.median(t, axis=0).astype(int).tolist(),
      ' total', int(np.median(t.sum(1))))
