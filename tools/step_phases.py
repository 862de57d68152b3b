"""GPU time of the phases of an eager training step (events on the main stream,
recorded from module / tensor hooks): U-Net forward, heads forward, rendering +
losses, heads backward (until the gradient of the U-Net's output is complete),
U-Net backward, optimiser + re-pack.   python tools/step_phases.py [--n_layers 4]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import ldi_enc_dec as script
base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4', '--n_layers', '4',
        '--img_height', '256', '--img_width', '768', '--checkpoint_dir', '/tmp/lsi_ph',
        '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000', '--log_freq', '1000000',
        '--bf16', 'true'] + sys.argv[1:]
tr = script.Trainer(script.apply_dataset_overrides(script.build_parser().parse_args(base)))
tr.setup()
ev = {}
def mark(name):
  e = torch.cuda.Event(enable_timing=True); e.record(); ev[name] = e
def unet_hook(_m, _i, out):
  mark('unet_fwd_end')
  feat = out[1]
  if feat.requires_grad:
    feat.register_hook(lambda g: (mark('heads_bwd_end'), g)[1])
tr.model.enc_dec.register_forward_hook(unet_hook)
tr.model.register_forward_hook(lambda _m, _i, _o: mark('heads_fwd_end'))
acc = {}
N = 30
for it in range(N + 5):
  batch = tr.feed(); staged, plan = tr.stage(batch)
  torch.cuda.synchronize()
  mark('start')
  tr.optim.zero_grad(set_to_none=True)
  total, _ = tr.compute_losses(staged)
  mark('losses_end')
  total.backward()
  mark('bwd_end')
  tr.optim.step(); tr._after_update()
  mark('step_end')
  torch.cuda.synchronize()
  if it >= 5:
    order = ['start', 'unet_fwd_end', 'heads_fwd_end', 'losses_end', 'heads_bwd_end', 'bwd_end', 'step_end']
    for a, b in zip(order[:-1], order[1:]):
      acc[b] = acc.get(b, 0.0) + ev[a].elapsed_time(ev[b])
names = {'unet_fwd_end': 'U-Net forward', 'heads_fwd_end': 'heads forward', 'losses_end': 'rendering + losses (forward)',
         'heads_bwd_end': 'losses / rendering / heads backward', 'bwd_end': 'U-Net backward (+ side-stream join)',
         'step_end': 'optimiser + re-pack'}
tot = sum(acc.values()) / N
print('eager step, synchronised per step: %.2f ms of GPU time on the main stream' % tot)
for k, v in acc.items():
  print('  %-40s %6.2f ms  %4.1f %%' % (names[k], v / N, 100 * v / N / tot))
