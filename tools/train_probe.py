"""Where a training step's wall time goes: feed / stage / step (eager or graph)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import ldi_enc_dec as script
argv = sys.argv[1:]
base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4', '--n_layers', '2', '--img_height', '256', '--img_width', '768',
        '--checkpoint_dir', '/tmp/lsi_ckpt', '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000', '--log_freq', '1000000']
opts = script.apply_dataset_overrides(script.build_parser().parse_args(base + argv))
tr = script.Trainer(opts); tr.setup()
for _ in range(6): tr.train_step()
torch.cuda.synchronize()
def timeit(fn, n=20):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): r = fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / n, r
t_feed, batch = timeit(tr.feed)
t_stage, (staged, plan) = timeit(lambda: tr.stage(batch))
if tr.use_graph:
  t_step, _ = timeit(lambda: tr._graph_step(staged, plan))
else:
  t_step, _ = timeit(lambda: tr._eager_step(staged))
print('bf16', opts.bf16, 'graph', tr.use_graph, 'feed %.2f ms  stage %.2f ms  step %.2f ms' % (t_feed, t_stage, t_step))
