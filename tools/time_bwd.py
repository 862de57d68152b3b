"""Times lsi_splat_bwd and lsi_splat_fwd_both / lsi_splat_bwd_both on a bench
workload (HIP events; microseconds per call).  LSI_BWD_STREAM=0 selects the
one-thread-per-pixel gather kernel, LSI_HIP_LIB=<name> an experiment build.
  python tools/time_bwd.py [--workload cfg3] [--shard-of N]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='cfg3')
ap.add_argument('--shard-of', type=int, default=1)
ap.add_argument('--disp', default='smooth')
ap.add_argument('--tex-layout', default='nhwc')
ap.add_argument('--path', default='auto')
ap.add_argument('--band-rows', type=int, default=0)
ap.add_argument('--threads', type=int, default=0)
args = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
b_local, _ = bench.shard_batch(args.workload, args.shard_of, 'strong')
r = bench.build_renderer(args.workload, b_local, 1000, dev, args)
r.launch()
torch.cuda.synchronize()
nl, h, w = bench.WORKLOADS[args.workload][:3]
bwd = min(bench.time_backward(r) for _ in range(3))
both = [min(x) for x in zip(*[bench.time_both(r) for _ in range(3)])]
print(json.dumps({'workload': args.workload, 'batch': b_local,
                  'lib': os.environ.get('LSI_HIP_LIB', ''),
                  'bwd_stream': os.environ.get('LSI_BWD_STREAM', '1'),
                  'rows': os.environ.get('LSI_BWD_STREAM_ROWS', ''),
                  'bwd_us': bwd, 'fwd_both_us': both[0], 'bwd_both_us': both[1],
                  'bwd_frac': bench.backward_bytes(nl, b_local, h, w) / (bwd * 1e-6) / 1e9 / bench.HBM_PEAK_GBPS}))
