"""Per-workgroup phase timestamps of the stream kernel (debug flag 4).
Needs the instrumented library: python layered-scene-inference_amd/build.py --hooks

  python tools/phase_probe.py <workload> [flags] [band_rows] [threads] [shard_of]
Stamps (tid 0): 0 start, 1 tile/windows cleared, 2 wave 0 has the row range and
the task table, 3 task loop starts, 4 wave 0 leaves the loop, 5 closing barrier
passed, 6 tile rows written (epilogue), 7 end.  12+w: wave w leaves the loop.
"""
import ctypes, os, sys
os.environ['LSI_HIP_LIB'] = 'hooks'
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench
from lsi import _C
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nl, h, w, batch, per_gpu, cams, max_disp, bg = bench.WORKLOADS[wl]
batch //= int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device('cuda:0')
tex, disp, mat = bench.make_inputs(nl, batch, h, w, cams, max_disp, 1000, dev,
                                   disp_kind=os.environ.get('LSI_PROBE_DISP', 'smooth'))
r = bench.Renderer(tex, disp, mat, max_disp, bg, 'stream', rows, threads)
r.desc.reserved = 4 | flags
base = (int(_C.lib().lsi_splat_workspace_bytes(ctypes.byref(r.desc))) + 255) // 256 * 256
nwg = 4096 * 8
r.ws = torch.zeros((base + nwg * 160 * 8,), dtype=torch.uint8, device=dev)
r.ws_bytes = r.ws.numel()
for _ in range(3):
  r.launch()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
  r.launch()
e1.record(); torch.cuda.synchronize()
print('instrumented kernel: %.1f us per launch' % (e0.elapsed_time(e1) * 1e3 / 20))
t = r.ws[base:].view(torch.int64).view(-1, 160).cpu().numpy()
t = t[t[:, 0] != 0]
rel = (t - t[:, :1]).astype(np.float64)
n = 8
print('workgroups', len(t))
print('median phase stamps (cycles since WG start):', np.median(rel[:, :n], axis=0).astype(int).tolist())
print('max:', rel[:, :n].max(axis=0).astype(int).tolist())
st = t[:, 0].astype(np.float64)
en = t[:, :n].max(axis=1).astype(np.float64)
print('WG start: p50/p100 since first start:', int(np.median(st) - st.min()), int(st.max() - st.min()),
      ' WG end p50/p100:', int(np.median(en) - st.min()), int(en.max() - st.min()))
pw = t[:, 12:28]
pw = np.where(pw != 0, pw - t[:, :1], 0)
print('per-wave loop end (median over WGs):', np.median(pw, axis=0).astype(int).tolist())
print('per-wave loop end (max over WGs):', pw.max(axis=0).astype(int).tolist())
prof = t[:, 32:160].reshape(len(t), 16, 8)[:, :, :6].astype(np.float64)
act = prof.sum(axis=2) > 0
names = ['load wait', 'projection', 'ticket+load issue', 'window phase', 'merge', 'task setup']
tot = prof[act].sum(axis=1).mean()
print('per-wave task-loop cycles by section (mean over active waves), total %d:' % tot)
for k, nme in enumerate(names):
  print('  %-18s %8d  %5.1f %%' % (nme, prof[act][:, k].mean(), 100 * prof[act][:, k].mean() / tot))
