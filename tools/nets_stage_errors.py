"""Per-stage bf16 error of the networks against the reference's nets.py values
(tests/golden/nets.npz), own kernels vs the library's bf16 path: the measurement
behind tests/test_nets_golden.py's bf16_stage_budget.

    python tools/nets_stage_errors.py > profiles/r06/nets_stage_errors.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import conftest  # noqa: F401,E402  (puts the package on sys.path)
import test_nets_golden as T  # noqa: E402


def run(own):
  from lsi.nnutils import nets, tf_checkpoint
  nets.IGEMM_CONV = own
  nets.MFMA_CONV = own
  g = T.golden('nets.npz')
  tag = 'unet'
  names, _, tf_vars = T._tf_variables(g, tag)
  model = T._model(tag)
  tf_checkpoint.load_tf_variables(model, tf_vars, strict=True)
  dev = torch.device('cuda:0')
  model = model.to(dev).train()
  prefix = {}
  for tf_name, key, _ in tf_checkpoint.variable_map(model):
    if tf_name.endswith('/weights'):
      prefix[tf_name[:-len('/weights')]] = key.rsplit('.', 2)[0]
  mods = dict(model.named_modules())
  got = {}
  for alias, pfx in prefix.items():
    def hook(_m, _i, out, alias=alias):
      got[alias] = out.detach().float().cpu()
    mods[pfx].register_forward_hook(hook)
  imgs = torch.tensor(T._images(g, tag), device=dev)
  with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
    tex, masks, disps = model.predict(imgs)
  stages = [str(s) for s in g[tag + '_stages']]
  shapes = [tuple(int(d) for d in str(s).split(',')) for s in g[tag + '_stage_shapes']]
  rep = []
  T._check_bf16_stages(g, tag, got, stages, shapes, rep)
  outs = {}
  for name, t in (('tex', tex), ('disp', disps)):
    flat = t.detach().float().cpu().reshape(-1).numpy()
    idx, want = g['%s_ldi_%s_idx' % (tag, name)], g['%s_ldi_%s_val' % (tag, name)]
    err = np.abs(flat[idx] - want)
    outs[name] = (float(err.mean()), float(np.percentile(err, 99)), float(err.max()))
  return rep, outs


def main():
  a, oa = run(True)
  b, ob = run(False)
  print('bf16 autocast, U-Net + 2 heads, 4 x 128 x 128 (tests/golden/nets.npz): RMS error over the')
  print('96 sampled activations of every stage / the stage\'s standard deviation')
  print('%-62s %5s %10s %10s %8s' % ('stage', 'depth', 'own', 'library', 'budget'))
  lib = {r[0]: r[2] for r in b}
  for alias, d, rms, budget in sorted(a, key=lambda r: (r[1], r[0])):
    print('%-62s %5d %10.4f %10.4f %8.3f%s' % (alias, d, rms, lib.get(alias, float('nan')), budget,
                                               '  OVER' if rms > budget else ''))
  print('final sigmoid outputs (mean / p99 / max abs error): own %s' % (oa,))
  print('                                                library %s' % (ob,))


if __name__ == '__main__':
  main()
