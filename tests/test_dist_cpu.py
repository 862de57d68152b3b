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
  mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  r0, r1 = out[0], out[1]
  assert r0['cfg2'] == (4, 'weak') and r1['cfg2'] == (4, 'weak')
  assert r0['cfg3'] == (16, 'strong') and r0['cfg5'] == (4, 'strong')
  assert r0['views_total'] == 8.0           # 4 views per rank, weak scaling
  assert r0['tmax'] == [2.0, 10.0] == r1['tmax']   # max over ranks
  assert r0['tex_sum'] != r1['tex_sum']      # different data shards
  np.testing.assert_array_equal(r0['mat'], r1['mat'])  # same cameras


def test_shard_rejects_uneven_split():
  import pytest
  sys.path.insert(0, ROOT)
  import bench
  with pytest.raises(SystemExit):
    bench.shard_batch('cfg5', 3)
  assert bench.shard_batch('cfg3', 8) == (4, 'strong')
  assert bench.algorithmic_bytes(2, 4, 256, 768) == 28311808
