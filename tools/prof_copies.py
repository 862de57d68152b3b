import sys, os, collections, torch
ROOT='/root/repo'
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import ldi_enc_dec as script
base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4', '--n_layers', '4',
        '--img_height', '256', '--img_width', '768', '--checkpoint_dir', '/tmp/lsi_ckpt',
        '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000', '--log_freq', '1000000', '--bf16', 'true']
opts = script.apply_dataset_overrides(script.build_parser().parse_args(base))
tr = script.Trainer(opts); tr.setup()
for _ in range(4): tr.train_step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
  tr.train_step(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
  if e.name in ('aten::copy_', 'aten::fill_', 'aten::add', 'aten::add_', 'aten::zeros', 'aten::zero_', 'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::clone'):
    st = [s for s in (e.stack or []) if 'repo' in s][:2]
    cnt[(e.name, str(e.input_shapes)[:60], ' | '.join(s.split('/')[-1][:70] for s in st))] += 1
for k, v in cnt.most_common(45): print(v, k)
