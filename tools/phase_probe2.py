"""Wall-clock timeline of the compact stream kernel (csrc/lsi_splat_stream2.hip
built with -DS2X_STAMPS: tools/build_variant.sh stamps lsi_splat_stream2.hip -DS2X_STAMPS).

  LSI_HIP_LIB=stamps python tools/phase_probe2.py <workload> [band_rows] [threads] [shard_of]
Per wave (lane 0), 100 MHz wall clock: 0 entry, 1 first loads issued, 2 tile
cleared + table filled, 3 opening barrier passed, 4 task loop left, 5 closing
barrier passed, 6 epilogue done.  Printed relative to the earliest stamp of the
launch, in microseconds.
"""
import ctypes, os, sys
os.environ.setdefault('LSI_HIP_LIB', 'stamps')
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench
from lsi import _C
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 0
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nl, h, w, batch, per_gpu, cams, max_disp, bg = bench.WORKLOADS[wl]
batch //= int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device('cuda:0')
sets = []
for i in range(3):
  tex, disp, mat = bench.make_inputs(nl, batch, h, w, cams, max_disp, 1000 + i, dev,
                                     os.environ.get('LSI_PROBE_DISP', 'smooth'))
  sets.append((tex, disp))
r = bench.Renderer(sets[0][0], sets[0][1], mat, max_disp, bg, 'stream', rows, threads, sets[1:])
r.desc.reserved = 4
# stamps: the kernel writes one KiB per workgroup into the LAST nbands * B KiB
# of the workspace it is handed (its start keeps the renderer's own use: the
# exchange bands' flags and rows); the buffer here has room for 8192 workgroups
nwg_max = 8192
base = (r.ws_bytes + 1023) // 1024 * 1024
r.ws = torch.zeros((base + nwg_max * 1024,), dtype=torch.uint8, device=dev)
r.ws_bytes = r.ws.numel()
for _ in range(6):
  r.launch()
torch.cuda.synchronize()
r.ws[base:].zero_()
torch.cuda.synchronize()
r.launch()
torch.cuda.synchronize()
t = r.ws[base:].view(torch.int64).view(-1, 16, 8).cpu().numpy().astype(np.float64)
wg_used = t[:, 0, 0] != 0
t = t[wg_used]
act = t[:, :, 0] != 0                      # waves that exist
# slot 7: items by route (A: 4 pre-summed cells per lane; B: per pixel, distinct
# cells; B': per pixel, lanes of a cell elected one at a time; C: general)
rt = t[:, :, 7].astype(np.uint64)
routes = [int(((rt >> np.uint64(16 * k)) & np.uint64(0xffff))[act].sum()) for k in range(4)]
print('items by route  A %d  B %d  B\' %d  C %d  (%.1f / %.1f / %.1f / %.1f %%)' % (
    tuple(routes) + tuple(100.0 * r / max(sum(routes), 1) for r in routes)))
t0 = t[:, :, 0][act].min()
us = (t - t0) / 100.0
print('workgroups', len(t), 'waves per WG', int(act.sum(axis=1).max()))
names = ['entry', 'loads issued', 'init done', 'barrier passed', 'loop left',
         'closing barrier', 'epilogue done']
for k, n in enumerate(names):
  v = us[:, :, k][act]
  print('%-16s min %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f' %
        (n, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
loop_end = np.where(act, us[:, :, 4], np.nan)
first = np.nanmin(loop_end, axis=1); last = np.nanmax(loop_end, axis=1)
print('per WG: first wave leaves the loop p50 %.2f, last p50 %.2f; spread p50 %.2f p90 %.2f max %.2f us' %
      (np.median(first), np.median(last), np.median(last - first),
       np.percentile(last - first, 90), (last - first).max()))
start = np.where(act, us[:, :, 0], np.nan)
print('WG entry (first wave): p50 %.2f max %.2f;  WG end p50 %.2f max %.2f' %
      (np.median(np.nanmin(start, axis=1)), np.nanmin(start, axis=1).max(),
       np.median(np.nanmax(np.where(act, us[:, :, 6], np.nan), axis=1)),
       np.nanmax(np.where(act, us[:, :, 6], np.nan))))
# how many waves are still in their task loop at time x
grid = np.linspace(0, np.nanmax(us[:, :, 6][act]), 41)
busy = [(int(((us[:, :, 3] <= x) & (us[:, :, 4] > x) & act).sum())) for x in grid]
print('waves inside the task loop over time (us: count):')
print('  ' + '  '.join('%.0f:%d' % (x, n) for x, n in zip(grid, busy)))

# where the slow workgroups are: by XCD (contiguous runs of workgroups), by band
# position inside the image, by batch element
end = np.nanmax(np.where(act, us[:, :, 6], np.nan), axis=1)
nwg_ = len(end)
if nwg_ % 8 == 0 and nwg_ >= 64:
  per_xcd = end.reshape(8, -1)
  print('WG end by XCD (mean / max):', ' '.join('%.1f/%.1f' % (a.mean(), a.max()) for a in per_xcd))
nb = nwg_ // batch
if nb * batch == nwg_ and nb > 1:
  byband = end.reshape(batch, nb)
  print('WG end by band position (mean over b):', ' '.join('%.1f' % v for v in byband.mean(axis=0)))
  print('WG end by batch element (mean over bands):', ' '.join('%.1f' % v for v in byband.mean(axis=1)))
