import sys, os, torch
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'layered-scene-inference_amd'))
sys.argv=['x']
import bench
from lsi.geometry import ldi
nl,h,w,batch,_,cams,md,bg = bench.WORKLOADS['cfg4']
dev=torch.device('cuda:0')
tex,disp,mat = bench.make_inputs(nl,batch,h,w,cams,md,1,dev)
for exp,name in ((0,'sweep'),(256,'gather')):
  for compose in (True, False):
    f=lambda: ldi.forward_splat_matrix([tex,None,disp], mat, compose_layers=compose, compute_trg_disp=True, trg_downsampling=0.5, bg_layer_disp=bg, max_disp=md, zbuf_scale=50., path='tile', experiment=exp)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(name, 'compose' if compose else 'per-layer', 'with disparity: %.1f us per call (host launch included)' % (e0.elapsed_time(e1)/20*1e3))
