"""Per-workgroup phase timestamps of the stream kernel (debug flag 4).
Needs the instrumented library: python layered-scene-inference_amd/build.py --hooks"""
import ctypes, os, sys
os.environ['LSI_HIP_LIB'] = 'hooks'
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench
from lsi import _C
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nl, h, w, batch, per_gpu, cams, max_disp, bg = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
tex, disp, mat = bench.make_inputs(nl, batch, h, w, cams, max_disp, 1000, dev)
r = bench.Renderer(tex, disp, mat, max_disp, bg, 'stream')
r.desc.reserved = 4 | flags
base = (int(_C.lib().lsi_splat_workspace_bytes(ctypes.byref(r.desc))) + 255) // 256 * 256
nwg = 4096 * 8
r.ws = torch.zeros((base + nwg * 32 * 8,), dtype=torch.uint8, device=dev)
r.ws_bytes = r.ws.numel()
for _ in range(3):
  r.launch()
torch.cuda.synchronize()
t = r.ws[base:].view(torch.int64).view(-1, 32).cpu().numpy()
t = t[t[:, 0] != 0]
rel = (t - t[:, :1]).astype(np.float64)
n = int((t[0] != 0).sum())
print('workgroups', len(t), 'stamps', n)
print('median phase stamps (cycles since WG start):', np.median(rel[:, :n], axis=0).astype(int).tolist())
print('max:', rel[:, :n].max(axis=0).astype(int).tolist())
st = t[:, 0].astype(np.float64)
en = t[:, :n].max(axis=1).astype(np.float64)
print('WG start: p0/p50/p100 since first start:', int(st.min() - st.min()), int(np.median(st) - st.min()), int(st.max() - st.min()),
      ' WG end p50/p100:', int(np.median(en) - st.min()), int(en.max() - st.min()))
pw = t[:, 16:32]
pw = np.where(pw != 0, pw - t[:, :1], 0)
print('per-wave x-pass end (median over WGs):', np.median(pw, axis=0).astype(int).tolist())
