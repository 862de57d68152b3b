"""Soak test of the side streams (weight gradients, per-layer decoders): N eager
training steps of the 4-layer model at 256 x 768 on a FIXED batch with the streams
on and off -- the two loss trajectories must stay finite, fall, and agree to the
noise of the batch-norm atomics (a race shows as NaN or as a trajectory that
leaves the other).  python tools/stream_soak.py [steps=300]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import ldi_enc_dec as script
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4', '--n_layers', '4',
        '--img_height', '256', '--img_width', '768', '--checkpoint_dir', '/tmp/lsi_soak',
        '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000', '--log_freq', '1000000',
        '--bf16', 'true']
out = {}
for mode in ('1', '0'):
  os.environ['LSI_WGRAD_STREAM'] = mode
  os.environ['LSI_HEAD_STREAMS'] = mode
  torch.manual_seed(0)
  tr = script.Trainer(script.apply_dataset_overrides(script.build_parser().parse_args(base)))
  tr.setup()
  batch = tr.feed()
  tr.feed = lambda: batch
  losses = []
  for i in range(steps):
    total, _ = tr.train_step()
    if i % 10 == 0 or i == steps - 1:
      losses.append(float(total))
  bad = [n for n, p in tr.model.named_parameters() if not bool(torch.isfinite(p).all())]
  out['streams_' + mode] = {'losses_every_10': losses, 'non_finite_parameters': bad,
                            'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9}
  del tr
  torch.cuda.empty_cache()
a, b = out['streams_1']['losses_every_10'], out['streams_0']['losses_every_10']
out['max_rel_diff'] = max(abs(x - y) / max(abs(y), 1e-6) for x, y in zip(a, b))
out['final'] = [a[-1], b[-1]]
print(json.dumps(out))
