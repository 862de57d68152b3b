"""The N>1 path on CPU: two `gloo` ranks exercise the batch sharding, the
per-rank seeding and the barrier + max-over-ranks reduction bench.py uses (on
the GPU box the backend is `nccl` = RCCL; the renderer itself has no
collective: independent LDIs shard along B)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _spawn(fn, args_of_port, nprocs):
  """mp.spawn on a port that was free a moment ago; if another process took it
  before the rendezvous bound it (EADDRINUSE), once more on a fresh one."""
  for attempt in range(4):
    try:
      return mp.spawn(fn, args=args_of_port(_free_port()), nprocs=nprocs, join=True)
    except Exception as e:  # pylint: disable=broad-except
      if 'EADDRINUSE' not in str(e) and 'address already in use' not in str(e).lower():
        raise
      if attempt == 3:
        raise


def _worker(rank, world, port, out):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import bench
  res = {}
  for wl in ('cfg2', 'cfg3', 'cfg5'):
    res[wl] = bench.shard_batch(wl, world)
    res[wl + '_weak'] = bench.shard_batch(wl, world, 'weak')
  # per-rank synthetic shard: same shapes, different content (seed 1000 + rank)
  nl, h, w = 2, 16, 32
  tex, disp, mat = bench.make_inputs(nl, 2, h, w, 'kitti', 0.4, 1000 + rank,
                                     torch.device('cpu'))
  res['tex_sum'] = float(tex.sum())
  res['mat'] = mat.numpy()
  dist.barrier()
  res['tmax'] = bench.reduce_max([1.0 + rank, 10.0 - rank], dist,
                                 torch.device('cpu'))
  # the weak-scaling aggregate: every rank renders its own shard
  views = torch.tensor([float(res['cfg2'][0])])
  dist.all_reduce(views)
  res['views_total'] = float(views)
  out[rank] = res
  dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing_reduction():
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  _spawn(_worker, lambda port: (world, port, out), world)
  r0, r1 = out[0], out[1]
  assert r0['cfg2'] == (4, 'weak') and r1['cfg2'] == (4, 'weak')
  # default: the workload's batch split over the ranks (32 -> B/N) ...
  assert r0['cfg3'] == (16, 'strong') and r0['cfg5'] == (4, 'strong')
  # ... --scaling weak: per-GPU work fixed (every rank renders the whole batch)
  assert r0['cfg3_weak'] == (32, 'weak') and r0['cfg5_weak'] == (8, 'weak')
  assert r0['cfg2_weak'] == (4, 'weak')
  assert r0['views_total'] == 8.0           # 4 views per rank, weak scaling
  assert r0['tmax'] == [2.0, 10.0] == r1['tmax']   # max over ranks
  assert r0['tex_sum'] != r1['tex_sum']      # different data shards
  np.testing.assert_array_equal(r0['mat'], r1['mat'])  # same cameras


def test_bare_shell_multi_gpu_launch_selftest():
  """`python bench.py --gpus 2` from a bare shell re-launches itself under
  torch.distributed.run, one rank per GPU, and rank 0 prints ONE JSON line with
  n_gpus = 2.  Here the ranks use gloo and skip the GPU work
  (--selftest-backend): the launch, rendezvous, shard and max-reduction logic
  is the same code the GPU run goes through."""
  import json
  import subprocess
  env = dict(os.environ)
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
       '--selftest-backend', 'gloo'], env=env, capture_output=True, text=True,
      timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout
  rec = json.loads(lines[0])
  assert rec['n_gpus'] == 2 and rec['scaling'] == 'strong'
  assert rec['views_per_step'] == 32          # cfg3: 32 views split 16 + 16
  assert abs(rec['elapsed_max'] - 2.0e-3) < 1e-9   # max over ranks


def test_self_launch_command_line():
  sys.path.insert(0, ROOT)
  import bench
  argv = bench.self_launch_argv(['--gpus', '8', '--steps', '5'], 8, port=1234)
  assert argv[1:3] == ['-m', 'torch.distributed.run']
  assert '--nproc-per-node' in argv and argv[argv.index('--nproc-per-node') + 1] == '8'
  assert argv[argv.index('--master-addr') + 1] == '127.0.0.1'
  assert argv[-4:] == ['--gpus', '8', '--steps', '5']
  assert bench.DEFAULT_WORKLOAD == 'cfg3'
  # every workload's input sets exceed the Infinity Cache when rotated
  for wl, (nl, h, w, batch, per_gpu, _, _, _) in bench.WORKLOADS.items():
    n = bench.rotation_sets(nl, batch, h, w)
    assert n * nl * batch * h * w * 16 >= bench.ROTATE_BYTES or n == 64


def test_shard_rejects_uneven_split():
  import pytest
  sys.path.insert(0, ROOT)
  import bench
  with pytest.raises(SystemExit):
    bench.shard_batch('cfg5', 3)
  assert bench.shard_batch('cfg3', 8) == (4, 'strong')
  assert bench.shard_batch('cfg5', 8) == (1, 'strong')
  assert bench.shard_batch('cfg3', 8, 'weak') == (32, 'weak')
  assert bench.algorithmic_bytes(2, 4, 256, 768) == 28311808
