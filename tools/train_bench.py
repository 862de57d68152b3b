"""Training-step throughput (samples/s) of the eager LDI trainer: U-Net + heads
on MIOpen, 4 HIP forward splats + backward, losses, Adam [+ DDP over RCCL].
  python tools/train_bench.py --img_height 256 --img_width 768 --batch_size 4 --n_layers 2
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...
"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import ldi_enc_dec as script

def main():
  argv = sys.argv[1:]
  steps, warm = 20, 5
  if '--steps' in argv:
    i = argv.index('--steps'); steps = int(argv[i + 1]); del argv[i:i + 2]
  base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4', '--n_layers', '2',
          '--img_height', '256', '--img_width', '768', '--checkpoint_dir', '/tmp/lsi_ckpt',
          '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000', '--log_freq', '1000000']
  opts = script.apply_dataset_overrides(script.build_parser().parse_args(base + argv))
  tr = script.Trainer(opts)
  tr.setup()
  for _ in range(warm):
    tr.train_step()
  torch.cuda.synchronize()
  if tr.dist is not None: tr.dist.barrier()
  t0 = time.perf_counter()
  for _ in range(steps):
    tr.train_step()
  torch.cuda.synchronize()
  if tr.dist is not None: tr.dist.barrier()
  dt = time.perf_counter() - t0
  if tr.rank == 0:
    print(json.dumps({'metric': 'training samples/s', 'value': opts.batch_size * tr.world * steps / dt,
                      'ms_per_step': dt * 1e3 / steps, 'n_gpus': tr.world, 'batch_per_gpu': opts.batch_size,
                      'n_layers': opts.n_layers, 'hw': [opts.img_height, opts.img_width],
                      'bf16': bool(opts.bf16), 'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9}))
  if tr.dist is not None: tr.dist.destroy_process_group()

if __name__ == '__main__':
  main()
