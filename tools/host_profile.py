"""cProfile of the eager training step's host side (where the Python time of a
step goes when the GPU is not the bound: the 2-layer step).
  python tools/host_profile.py [--n_layers 2] [trainer flags]"""
import cProfile, pstats, sys, os, io, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'layered-scene-inference_amd'))
import ldi_enc_dec as script
base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4', '--n_layers', '2',
        '--img_height', '256', '--img_width', '768', '--checkpoint_dir', '/tmp/lsi_ckpt',
        '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000', '--log_freq', '1000000', '--bf16', 'true'] + sys.argv[1:]
opts = script.apply_dataset_overrides(script.build_parser().parse_args(base))
tr = script.Trainer(opts); tr.setup()
for _ in range(5): tr.train_step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): tr.train_step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host ms/step (launch side) %.2f, with the final sync %.2f' % ((t1 - t0) * 50, (t2 - t0) * 50))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): tr.train_step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
