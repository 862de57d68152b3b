"""bench.py as the driver runs it on a GPU box: one JSON line on stdout, also
when it is a rank of a torch.distributed.run job (RCCL process group alive while
the timed steps are captured into a HIP graph)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _torchrun(nproc, bench_args, env):
  """`python -m torch.distributed.run ... bench.py` on a port that was free a
  moment ago; another process may take it before the rendezvous binds it
  (EADDRINUSE: seen once in a suite run) -- then once more on a fresh port."""
  for _ in range(4):
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1', '--master-port',
         str(_free_port()), os.path.join(ROOT, 'bench.py')] + bench_args,
        env=env, capture_output=True, text=True, timeout=900)
    if out.returncode == 0 or 'EADDRINUSE' not in out.stderr:
      break
  return out


def _one_json_line(out):
  lines = [l for l in out.stdout.splitlines() if l.strip()]
  assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
  return json.loads(lines[0])


def test_bench_line_from_a_bare_shell(built_lib):
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup',
       '2', '--workload', 'cfg2', '--no-cpu-baseline', '--no-extra', '--traffic', 'off'],
      env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  rec = _one_json_line(out)
  assert rec['n_gpus'] == 1 and rec['steps'] == 5 and rec['warmup'] == 2
  assert rec['unit'] == 'views/s' and rec['value'] > 0
  rf = rec['roofline']
  assert rf['bound'] == 'hbm' and rf['kernel'] == 'splat_stream2_kernel'
  assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
  assert rec['config']['launch'] == 'graph'


def test_bench_line_as_a_rank_of_torch_distributed_run(built_lib):
  """The driver's N > 1 command with N = 1 (one GPU here): RCCL comes up, its
  banner stays off stdout, the barriers and the max-over-ranks run, the steps
  are captured into a HIP graph with the process group alive."""
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  out = _torchrun(1, ['--gpus', '1', '--steps', '5', '--warmup', '2', '--workload', 'cfg2',
                      '--no-cpu-baseline', '--no-extra', '--traffic', 'off'], env)
  assert out.returncode == 0, out.stderr[-2000:]
  rec = _one_json_line(out)
  assert rec['n_gpus'] == 1 and rec['scaling'] == 'weak'
  assert rec['config']['launch'] == 'graph'
  per_rank = rec['roofline']['avg_launch_us_per_rank']
  assert per_rank['min'] <= per_rank['max']


def test_two_ranks_split_the_batch_and_add_the_weak_figure(built_lib):
  """The driver's N > 1 line on a one-GPU box: two ranks share device 0
  (LSI_BENCH_SHARE_GPU: gloo rendezvous, RCCL refuses two ranks per device) --
  the batch is split over the ranks (`scaling: strong`), rank 0 prints ONE JSON
  line whose `value` counts both ranks' views and `extra.weak` holds the figure
  of the same run with the whole batch on every rank."""
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  env['LSI_BENCH_SHARE_GPU'] = '1'
  out = _torchrun(2, ['--gpus', '2', '--steps', '5', '--warmup', '2', '--no-cpu-baseline',
                      '--no-extra', '--traffic', 'off'], env)
  assert out.returncode == 0, out.stderr[-3000:]
  rec = _one_json_line(out)
  assert rec['n_gpus'] == 2 and rec['scaling'] == 'strong'
  assert 'batch 16 per GPU (32 total)' in rec['config']['workload']
  assert rec['value'] > 0 and abs(rec['value'] - 32 * 5 / (rec['ms_per_step'] * 5e-3)) < 1e-6 * rec['value']
  weak = rec['extra']['weak']
  assert weak['scaling'] == 'weak' and weak['views_per_gpu'] == 32 and weak['value'] > 0
