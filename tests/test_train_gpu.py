"""Full training step on the GPU: U-Net + LDI heads (MIOpen), four HIP forward
splats with their HIP backward, all six losses, Adam."""
import sys

import numpy as np
import pytest
import torch

from conftest import PKG

pytestmark = pytest.mark.gpu


def _trainer(tmp_path, **kw):
  sys.path.insert(0, PKG)
  import ldi_enc_dec as script
  args = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '2', '--n_layers', '2',
          '--img_height', '128', '--img_width', '256', '--num_iter', '3',
          '--log_freq', '1', '--checkpoint_dir', str(tmp_path)]
  # (the tests that do not say otherwise pin the renderer and the losses with the
  # fp32 network; the product default is --bf16 true)
  kw.setdefault('bf16', 'false')
  for k, v in kw.items():
    args += ['--' + k, str(v)]
  opts = script.apply_dataset_overrides(script.build_parser().parse_args(args))
  tr = script.Trainer(opts)
  tr.setup()
  return tr


@pytest.mark.parametrize('bf16', ['false', 'true'])
def test_full_training_step(tmp_path, built_lib, bf16):
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  tr = _trainer(tmp_path, bf16=bf16)
  batch = tr.feed()
  tr.feed = lambda: batch                     # overfit one batch
  losses = []
  for _ in range(6):
    total, scalars = tr.train_step()
    losses.append(float(total))
    assert all(np.isfinite(float(v)) for v in scalars.values())
  assert all(p.grad is not None and bool(torch.isfinite(p.grad).all())
             for p in tr.model.parameters())
  assert float(scalars['indep_splat_loss']) > 0
  assert float(scalars['compose_splat_loss']) > 0
  assert losses[-1] < losses[0], losses       # the step optimises the objective


def test_splat_loss_gradient_reaches_the_network(tmp_path, built_lib):
  tr = _trainer(tmp_path, self_cons_wt=0, disp_smoothness_wt=0, incr_depth_wt=0)
  total, _ = tr.train_step()
  head = tr.model.ldi_tex_disp.pixelwise_pred.preds[0].conv.weight.grad
  enc = tr.model.enc_dec.encoder.cnv1.conv.weight.grad
  assert float(head.abs().sum()) > 0 and float(enc.abs().sum()) > 0


def test_hip_graph_step_follows_the_eager_trajectory(tmp_path, built_lib):
  """--hip_graph: after 3 eager warm-up steps the whole step (network, four
  HIP splats and their backward, losses, Adam) is one captured graph.  Same
  batch, same seed: the graphed run must reproduce the eager run's losses."""
  runs = {}
  for mode in ('false', 'true', 'flat'):
    # 'flat': the data-parallel form of the graphed step (what --hip_graph runs
    # with more than one rank): gradients in one flat buffer, graph A (forward,
    # losses, backward), the all-reduce (none to do on one rank), graph B (Adam)
    kw = dict(hip_graph='true', flat_grads='true') if mode == 'flat' else \
        dict(hip_graph=mode)
    tr = _trainer(tmp_path / mode, **kw)
    batch = tr.feed()
    tr.feed = lambda batch=batch: batch
    runs[mode] = [float(tr.train_step()[0]) for _ in range(8)]
    if mode != 'false':
      assert tr._graph is not None              # steps 4.. were replays
    if mode == 'flat':
      assert tr._graph_b is not None
      assert all(p.grad.data_ptr() >= tr._flat.data_ptr() for p in tr.model.parameters())
  eager = runs['false']
  for graphed in (runs['true'], runs['flat']):
    assert graphed[-1] < graphed[0]
    for a, b in zip(eager, graphed):
      assert abs(a - b) <= 2e-2 * abs(a), (eager, graphed)  # MIOpen wrw is not
      # run-to-run deterministic; the trajectories agree to a fraction of a step


def test_debug_synth_texture_sanity_check(tmp_path):
  """The reference's end-to-end renderer check (ldi_enc_dec.py:52-55, 223-225):
  on procedural box-room scenes with the ground-truth (fg, bg) disparities fed
  in place of the predictions, one training step runs through the planar
  renderer (HIP bilinear + compose), the LDI splat (HIP) and every loss, and all
  six scalars are finite."""
  import ldi_enc_dec as script
  argv = ['--dataset', 'synthetic', '--synth_scene', 'planes',
          '--debug_synth_texture', 'true', '--batch_size', '1', '--n_layers', '2',
          '--img_height', '128', '--img_width', '128', '--n_obj_max', '2',
          '--checkpoint_dir', str(tmp_path), '--log_freq', '1000000',
          '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000']
  opts = script.apply_dataset_overrides(script.build_parser().parse_args(argv))
  tr = script.Trainer(opts)
  tr.setup()
  total, scalars = tr.train_step()
  assert tr.gt_disps is not None and tr.gt_disps[0].shape == (2, 1, 128, 128, 1)
  assert np.isfinite(float(total))
  for k, v in scalars.items():
    assert np.isfinite(float(v)), k
  # (textures still come from the untrained net: only finiteness is asserted
  # here; the geometric consistency of the scenes with their ground-truth
  # disparities is tests/test_sampling_gpu.py::test_scene_generator_views_...)


def test_eval_script_writes_results(tmp_path):
  """ldi_pred_eval.py (the Tester of test_utils.py:182-255): two evaluation
  iterations on procedural planar scenes with ground truth; results.txt holds
  sum(metric) / sum(norm) for every metric of ldi_pred_eval.py:297-548."""
  import ldi_pred_eval as ev
  argv = ['--dataset', 'synthetic', '--synth_scene', 'planes', '--batch_size', '1',
          '--n_layers', '2', '--img_height', '128', '--img_width', '128',
          '--n_obj_max', '2', '--num_eval_iter', '2', '--random_weights', 'true',
          '--checkpoint_dir', str(tmp_path)]
  opts = script_overrides(ev, argv)
  tester = ev.Tester(opts)
  results = tester.test()
  for k in ('compose_splat_loss', 'compose_splat_loss_disocc', 'depth_splat_loss',
            'fg_tex_error', 'fg_disp_error', 'bg_tex_error', 'bg_disp_error', 'psnr'):
    assert k in results and np.isfinite(results[k]), k
  assert 0 <= results['compose_splat_loss'] <= 1
  import os
  text = open(os.path.join(opts.checkpoint_dir, 'results', 'results.txt')).read()
  assert 'fg_disp_error' in text


def script_overrides(ev, argv):
  import ldi_enc_dec as script
  opts = script.apply_dataset_overrides(ev.build_parser().parse_args(argv))
  opts.debug_synth_texture = False
  opts.synth_dl_eval_data = True
  return opts


def test_compute_losses_six_scalars_match_the_oracle(tmp_path, built_lib):
  """Trainer.compute_losses itself (ldi_enc_dec.py:265-410) on a FIXED pair of
  LDIs (the network is bypassed): both view-synthesis directions through
  forward_splat_both, the two self-consistency terms, the regularisers and the
  weighting (/ max_disp, / max_disp^2) against the NumPy oracle."""
  import lsi_oracle as O
  tr = _trainer(tmp_path, img_height=64, img_width=256, self_cons_wt=10,
                incr_depth_wt=7, disp_smoothness_wt=0.3, compose_splat_wt=1.5,
                indep_splat_wt=0.5)
  o = tr.opts
  rs = np.random.RandomState(11)
  nl, b, h, w = o.n_layers, o.batch_size, o.img_height, o.img_width
  f32 = lambda a: np.asarray(a, np.float32)
  img_s, img_t = f32(rs.rand(b, h, w, 3)), f32(rs.rand(b, h, w, 3))
  ldis = []
  for _ in range(2):
    tex = f32(rs.rand(nl, b, h, w, 3))
    disp = f32(o.max_disp * rs.rand(nl, b, h, w, 1))
    ldis.append((tex, disp))
  k = f32([[0.58 * w, 0, w / 2.0], [0, 0.58 * w, h / 2.0], [0, 0, 1.0]])
  k_s = np.broadcast_to(k, (b, 3, 3)).copy()
  k_t = k_s.copy(); k_t[:, 0, 2] += 1.5                    # principal point moves
  rot = np.broadcast_to(np.eye(3, dtype=np.float32), (b, 3, 3)).copy()
  t = np.broadcast_to(f32([[-0.532], [0.0], [0.0]]), (b, 3, 1)).copy()
  T = torch.tensor
  staged, _ = tr.stage((T(img_s), T(img_t), T(k_s), T(k_t), T(rot), T(t)))
  dev = tr.device
  fixed = [[T(tex).to(dev), None, T(disp).to(dev)] for tex, disp in ldis]
  tr.train_model = lambda a, b_: (fixed[0], fixed[1])
  total, got = tr.compute_losses(staged)

  md, bg, zb, s = o.max_disp, o.bg_layer_disp, o.zbuf_scale, o.trg_splat_downsampling
  ones = np.ones((nl, b, h, w, 1), np.float32)
  m_trg = O.forward_projection_matrix(k_s, k_t, rot, t)
  rot_inv = np.swapaxes(rot, -1, -2)
  m_src = O.forward_projection_matrix(k_t, k_s, rot_inv,
                                      -np.matmul(rot_inv, t).astype(np.float32))
  want = {k_: 0.0 for k_ in ('self_cons_loss', 'compose_splat_loss',
                             'indep_splat_loss', 'incr_depth_loss',
                             'disp_smoothness_loss')}
  for (tex, disp), own, other, mat in ((ldis[0], img_s, img_t, m_trg),
                                       (ldis[1], img_t, img_s, m_src)):
    want['self_cons_loss'] += O.zbuffer_composition_loss(tex, ones, disp, own,
                                                         bg, md, zb)
    for compose, key in ((True, 'compose_splat_loss'), (False, 'indep_splat_loss')):
      r = O.forward_splat(tex, ones, disp, mat, s, bg, md, zb, compose)
      want[key] += O.view_synthesis_loss(r['img'], other, o.splat_bdry_ignore)
    want['incr_depth_loss'] += O.decreasing_disp_loss(disp)
    want['disp_smoothness_loss'] += O.disp_smoothness_loss(disp)
  want['total_loss'] = (o.self_cons_wt * want['self_cons_loss'] +
                        o.compose_splat_wt * want['compose_splat_loss'] +
                        o.indep_splat_wt * want['indep_splat_loss'] +
                        o.incr_depth_wt / md * want['incr_depth_loss'] +
                        o.disp_smoothness_wt / (md * md) * want['disp_smoothness_loss'])
  assert set(got) == set(want)
  for key, v in want.items():
    assert abs(float(got[key]) - float(v)) <= 2e-5 * abs(float(v)), (
        key, float(got[key]), float(v))
  assert abs(float(total) - float(want['total_loss'])) <= 2e-5 * abs(want['total_loss'])


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('shape', [(4, 32, 64, 96), (2, 512, 2, 6), (2, 64, 1, 1),
                                   (3, 128, 17, 23), (2, 256, 8, 24),
                                   (4, 32, 256, 384)])
def test_fused_batch_norm_relu_matches_torch(shape, dtype, built_lib):
  """csrc/lsi_bn.hip (batch statistics, beta, ReLU; slim.batch_norm defaults,
  nets.py:44-67) against torch.nn.functional.batch_norm + relu in fp32 (fp64 for
  the gradient reference): forward, dx and dbeta."""
  from lsi.nnutils import _hip_bn
  dev = torch.device('cuda:0')
  dt = getattr(torch, dtype)
  gen = torch.Generator(device='cpu').manual_seed(sum(shape))
  n, c, h, w = shape
  x = (torch.randn(shape, generator=gen) * 1.7 + 0.4).to(dev).to(dt)
  x = x.contiguous(memory_format=torch.channels_last)
  beta = (torch.randn(c, generator=gen) * 0.3).to(dev)
  g = torch.randn(shape, generator=gen).to(dev).to(dt).contiguous(
      memory_format=torch.channels_last)
  assert _hip_bn.supported(x)
  xq = x.detach().clone().requires_grad_(True)
  bq = beta.detach().clone().requires_grad_(True)
  y = _hip_bn.batch_norm_relu(xq, bq, 1e-3, True)
  assert y.dtype == dt and y.is_contiguous(memory_format=torch.channels_last)
  y.backward(g)
  # reference on the SAME (possibly bf16-rounded) inputs, in fp64
  xr = x.double().detach().requires_grad_(True)
  br = beta.double().detach().requires_grad_(True)
  yr = torch.relu(torch.nn.functional.batch_norm(
      xr, None, None, torch.ones(c, dtype=torch.float64, device=dev), br, True,
      0.0, 1e-3))
  yr.backward(g.double())
  if dtype == 'float32':
    fwd_tol, grad_tol = 2e-5, 2e-4
  else:     # outputs rounded to bf16 (8 bits): half an ulp of the largest value
    fwd_tol, grad_tol = 2e-2, 3e-2
  scale = float(yr.abs().max()) + 1e-6
  assert float((y.double() - yr).abs().max()) <= fwd_tol * scale
  # ReLU masks agree except within rounding of zero
  differ = ((y > 0) != (yr > 0))
  assert float(differ.float().mean()) < 1e-3
  # (gradients: away from the ReLU's kink -- an element whose pre-activation is
  # within rounding of 0 has its mask decided by the last bit of the batch
  # statistics, in fp32 here and in fp64 there)
  with torch.no_grad():
    mean = x.double().mean(dim=(0, 2, 3), keepdim=True)
    var = x.double().var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    z = (x.double() - mean) / torch.sqrt(var + 1e-3) + beta.double().view(1, -1, 1, 1)
    kink = z.abs() < (1e-5 if dtype == 'float32' else 2e-2)
  assert float(kink.float().mean()) < 2e-2
  gs = float(xr.grad.abs().max()) + 1e-12
  err = (xq.grad.double() - xr.grad).abs()
  assert float(err[~kink].max()) <= grad_tol * gs, dtype
  # dbeta: the kink elements' gradients may be counted or not
  slack = (g.double().abs() * kink).sum(dim=(0, 2, 3))
  bs = float(br.grad.abs().max()) + 1e-12
  assert bool(((bq.grad.double() - br.grad).abs() <= grad_tol * bs + slack).all())
  # run to run: the cross-workgroup sums are fp32 atomics (arrival order), the
  # statistics agree to rounding
  y2 = _hip_bn.batch_norm_relu(x, beta, 1e-3, True)
  assert float((y2.double() - y.detach().double()).abs().max()) <= fwd_tol * scale


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('shape', [(4, 32, 64, 96), (2, 512, 2, 6), (6, 64, 16, 48),
                                   (8, 32, 256, 384)])
def test_fused_batch_norm_groups(shape, dtype, built_lib):
  """groups: sub-batches along N normalised with their own statistics in one
  launch (what nets.bn_groups asks for: source and target views in one batch)
  == the op applied to each sub-batch."""
  from lsi.nnutils import _hip_bn
  dev = torch.device('cuda:0')
  dt = getattr(torch, dtype)
  gen = torch.Generator(device='cpu').manual_seed(sum(shape) + 1)
  n, c, h, w = shape
  groups = 2
  x = torch.randn(shape, generator=gen)
  x[n // 2:] = x[n // 2:] * 3.0 - 1.0        # the halves differ in mean and spread
  x = x.to(dev).to(dt).contiguous(memory_format=torch.channels_last)
  beta = (torch.randn(c, generator=gen) * 0.3).to(dev)
  g = torch.randn(shape, generator=gen).to(dev).to(dt).contiguous(
      memory_format=torch.channels_last)
  assert _hip_bn.supported(x, groups)
  xq = x.detach().clone().requires_grad_(True)
  bq = beta.detach().clone().requires_grad_(True)
  y = _hip_bn.batch_norm_relu(xq, bq, 1e-3, True, groups)
  y.backward(g)
  ys, dxs, dbs = [], [], 0
  for xc, gc in zip(x.chunk(groups, 0), g.chunk(groups, 0)):
    xc = xc.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bc = beta.detach().clone().requires_grad_(True)
    yc = _hip_bn.batch_norm_relu(xc, bc, 1e-3, True, 1)
    yc.backward(gc.contiguous(memory_format=torch.channels_last))
    ys.append(yc.detach()); dxs.append(xc.grad); dbs = dbs + bc.grad
  tol = 1e-5 if dtype == 'float32' else 2e-2
  yr, dxr = torch.cat(ys, 0).double(), torch.cat(dxs, 0).double()
  assert float((y.double() - yr).abs().max()) <= tol * (float(yr.abs().max()) + 1e-6)
  # (the statistics are fp32 atomic sums: run to run they differ in the last
  # bit, and an element within rounding of the ReLU's kink may change sides)
  off = (xq.grad.double() - dxr).abs() > tol * (float(dxr.abs().max()) + 1e-12)
  assert int(off.sum()) <= max(2, int(2e-6 * off.numel())), int(off.sum())
  slack = 4.0 * float(g.float().abs().max()) * max(2, int(2e-6 * off.numel()))
  assert float((bq.grad - dbs).abs().max()) <= 1e-4 * (float(dbs.abs().max()) + 1e-12) + (
      slack if int(off.sum()) else 0.0)


def test_fused_batch_norm_is_what_the_network_runs(tmp_path, built_lib):
  """The conv layers go through the fused kernels (channels-last, training
  mode), and the step they produce equals the MIOpen-batch-norm step."""
  import lsi.nnutils.nets as nets
  from lsi.nnutils import _hip_bn
  calls = []
  orig = _hip_bn.batch_norm_relu
  _hip_bn.batch_norm_relu = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
  try:
    tr = _trainer(tmp_path / 'a')
    batch = tr.feed()
    tr.feed = lambda: batch
    fused, _ = tr.train_step()
  finally:
    _hip_bn.batch_norm_relu = orig
  assert len(calls) >= 30, len(calls)   # every conv layer, once (src and trg in one pass)
  nets.FUSED_BN = False
  try:
    tr2 = _trainer(tmp_path / 'b')
    tr2.feed = lambda: batch
    plain, _ = tr2.train_step()
  finally:
    nets.FUSED_BN = True
  assert abs(float(fused) - float(plain)) <= 2e-3 * abs(float(plain)), (
      float(fused), float(plain))


def test_batched_pair_pass_on_the_gpu_and_rgbd_output_layout(tmp_path, built_lib):
  """One pass over [src; trg] with per-view batch-norm statistics (fused HIP
  batch norm, groups = 2) equals two passes; and what the network hands to the
  renderer is one buffer of RGBD pixels (textures and scaled disparities are
  views of it), which the descriptor recognises (LSI_PACKED_RGBD)."""
  from lsi import _C
  from lsi.geometry import ldi
  tr = _trainer(tmp_path)
  net, dev = tr.model, tr.device
  g = torch.Generator().manual_seed(2)
  src = torch.rand(2, 128, 256, 3, generator=g).to(dev)
  trg = torch.rand(2, 128, 256, 3, generator=g).to(dev)
  outs, grads = {}, {}
  for mode in (True, False):
    net.batched_pairs = mode
    net.zero_grad()
    a, b = net(src, trg)
    outs[mode] = [a[0], a[2], b[0], b[2]]
    (a[0].sum() + 2 * a[2].sum() + 3 * b[0].sum() + 4 * b[2].sum()).backward()
    grads[mode] = [p.grad.clone() for p in net.parameters()]
    tex, disp = a[0], a[2]
    d = ldi._desc(tex, None, disp, 64, 128, 0.5, 0.4, 50.0, 0.1, 0, 0)
    assert d.flags & _C.LSI_PACKED_RGBD, (mode, tex.stride(), disp.stride())
  for x, y in zip(outs[True], outs[False]):
    torch.testing.assert_close(x, y, rtol=1e-3, atol=1e-4)
  num = sum(float(((x - y) ** 2).sum()) for x, y in zip(grads[True], grads[False]))
  den = sum(float((y ** 2).sum()) for y in grads[False])
  assert num <= 1e-4 * den, num / den       # (MIOpen picks other kernels at batch 4 / 2)


def test_paired_splat_equals_one_call_per_direction(tmp_path, built_lib):
  """--paired_splat (default): both directions of a pair rendered by ONE
  forward_splat_both call on the 2 B LDIs of the batched network pass
  (reference ldi_enc_dec.py:302-334 makes one call per direction).  On a FIXED
  buffer of 2 B RGBD LDIs (the network is bypassed, as in the six-scalar test):
  the scalars and the gradient with respect to every LDI value equal those of
  one call per direction."""
  tr = _trainer(tmp_path)
  o = tr.opts
  nl, b, h, w = o.n_layers, o.batch_size, o.img_height, o.img_width
  batch = tr.feed()
  staged, _ = tr.stage(batch)
  g = torch.Generator().manual_seed(21)
  base = torch.rand((nl, 2 * b, h, w, 4), generator=g)
  base[..., 3] *= o.max_disp
  res = {}
  for paired in (True, False):
    pred = base.clone().to(tr.device).requires_grad_(True)
    tex, disp = pred[..., 0:3], pred[..., 3:4]
    half = lambda t, k: t[:, k * b:(k + 1) * b]
    src, trg = [half(tex, 0), None, half(disp, 0)], [half(tex, 1), None, half(disp, 1)]

    def fake(_a, _b, src=src, trg=trg, full=[tex, None, disp], paired=paired):
      tr.model.pair_ldi = full if paired else None
      return src, trg
    tr.train_model = fake
    total, scalars = tr.compute_losses(staged)
    total.backward()
    res[paired] = ({k: float(v) for k, v in scalars.items()}, pred.grad.clone())
  (sa, ga), (sb, gb) = res[True], res[False]
  for k in sa:
    assert abs(sa[k] - sb[k]) <= 1e-5 * abs(sb[k]) + 1e-9, (k, sa[k], sb[k])
  scale = float(gb.abs().max())
  assert scale > 0
  assert float((ga - gb).abs().max()) <= 2e-5 * scale, float((ga - gb).abs().max()) / scale


@pytest.mark.parametrize('graph', ['false', 'true'])
def test_convolutions_run_on_the_weights_of_the_step(tmp_path, built_lib, graph):
  """The implicit-GEMM layers keep their weights in operand order
  (lsi_conv2d_pack); after every optimiser step the packs have to be those of
  the updated parameters.  (Round 5 found them stale: the trainer's parameters
  have channels-last strides, which the re-pack skipped, and torch's fused Adam
  does not move a parameter's version counter -- every such layer ran on its
  initial weights, silently.)  Checked after every step against a fresh pack of
  the present weights, eager and as a captured graph."""
  import ctypes
  from lsi import _C
  from lsi.nnutils import _hip_conv
  tr = _trainer(tmp_path, bf16='true', hip_graph=graph)
  batch = tr.feed()
  tr.feed = lambda: batch
  lib = _C.lib()

  def fresh(e, w):
    buf = torch.empty_like(e.buf)
    rc = lib.lsi_conv2d_pack(ctypes.byref(e.desc), e.mode, _C.ptr(w.float().contiguous()),
                             _C.ptr(buf), buf.numel() * 2, _C.stream_ptr(w.device))
    assert rc == 0
    return buf

  mine = lambda: {k: e for k, e in _hip_conv._PACKED.items()
                  if e.wref() is not None and any(e.wref() is p for p in tr.model.parameters())}
  first, checked = None, 0
  for step in range(6):
    tr.train_step()
    torch.cuda.synchronize()
    # the trainer re-packs right after the optimiser's update: what the next
    # step's forward and backward will read is the pack of the present weights
    for k, e in mine().items():
      assert torch.equal(fresh(e, e.wref().detach()), e.buf), (step, k)
      checked += 1
    first = first or {k: e.wref().detach().clone() for k, e in mine().items()}
  assert checked >= 6 * 40, checked     # (dozens of layers, both directions)
  # ... and the parameters did move
  assert all(float((e.wref().detach() - first[k]).abs().max()) > 0
             for k, e in mine().items() if k in first)


def test_own_convolutions_follow_the_librarys_training_trajectory(tmp_path, built_lib, monkeypatch):
  """Eight steps on one batch with the implicit-GEMM kernels (packed weights,
  statistics in the epilogue, two-tensor skip connections) against the same
  steps with every convolution on the library (MIOpen, no packing at all): the
  losses must agree to bf16 noise step by step -- a layer that does not see its
  updated weights shows up here, whatever hides it."""
  from lsi.nnutils import nets
  runs = {}
  for own in (True, False):
    monkeypatch.setattr(nets, 'IGEMM_CONV', own)
    tr = _trainer(tmp_path / ('own' if own else 'lib'), bf16='true')
    batch = tr.feed()
    tr.feed = lambda batch=batch: batch
    runs[own] = [float(tr.train_step()[0]) for _ in range(8)]
  a, b = runs[True], runs[False]
  assert a[-1] < a[0] and b[-1] < b[0]
  for x, y in zip(a, b):
    assert abs(x - y) <= 2e-2 * abs(y), (a, b)   # (the tolerance of the graph-vs-eager test)


def test_network_forward_full_size_own_kernels_against_the_fp32_library(tmp_path, built_lib):
  """BASELINE config 3's model at full size (4 layers, 256 x 768, batch 4 = 8
  images per pass with two batch-norm groups): the forward pass through the own
  kernels (bf16 MFMA convolutions incl. the first layer, the split bottleneck
  layers, the two-tensor skip convolutions, fused batch norms with the statistics
  from the convolutions' epilogue, the heads' decoders on their own streams) against
  the same weights in fp32 on the library.  bf16 through ~30 stages: the sigmoid
  outputs agree to 2e-2 in the mean, 0.12 at the 99th percentile (the bar of the
  reference-pinned 128 x 128 test, tests/test_nets_golden.py); a structural error
  at this size -- a tile edge, a row block past the image, a class of a transposed
  convolution -- is O(1) over a region."""
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  from lsi.nnutils import nets
  tr = _trainer(tmp_path, bf16='true', batch_size=4, n_layers=4, img_height=256, img_width=768)
  model = tr.model.train()
  g = torch.Generator(device=tr.device).manual_seed(3)
  # (the reference's layout: B x H x W x 3)
  src = torch.rand((4, 256, 768, 3), generator=g, device=tr.device)
  trg = torch.rand((4, 256, 768, 3), generator=g, device=tr.device)
  cl = lambda t: t
  old = nets.enable_head_streams(True)
  try:
    with torch.no_grad():
      with torch.autocast('cuda', dtype=torch.bfloat16):
        own = model(cl(src), cl(trg))
      own = [[None if t is None else t.float().clone() for t in ldi] for ldi in own]
      nets.enable_head_streams(False)
      ref = model(cl(src), cl(trg))          # fp32: every convolution on the library
  finally:
    nets.enable_head_streams(old)
  for ldi_o, ldi_r in zip(own, ref):
    for name, a, b in zip(('tex', 'mask', 'disp'), ldi_o, ldi_r):
      if a is None:
        assert b is None
        continue
      assert a.shape == b.shape and a.shape[0] == 4 and tuple(a.shape[2:4]) == (256, 768)
      assert bool(torch.isfinite(a).all())
      scale = float(tr.opts.max_disp) if name == 'disp' else 1.0
      err = (a - b.float()).abs() / scale
      q99 = float(torch.quantile(err.flatten()[::97], 0.99))
      assert float(err.mean()) <= 2e-2 and q99 <= 0.12, (name, float(err.mean()), q99,
                                                         float(err.max()))


def test_training_step_at_baseline_config_4_shape(tmp_path, built_lib):
  """BASELINE config 4's network half -- 3-layer LDI, 256 x 256, bf16 convolutions
  + fp32 splat -- as whole training steps (batch 8 here; bench.py's
  extra.train_step times batch 16): finite losses, every parameter gets a finite
  gradient, and the objective falls on a repeated batch."""
  if not torch.cuda.is_available():
    pytest.fail('gpu test selected but no ROCm device is visible')
  tr = _trainer(tmp_path, bf16='true', batch_size=8, n_layers=3, img_height=256, img_width=256)
  batch = tr.feed()
  tr.feed = lambda: batch
  losses = []
  for _ in range(6):
    total, scalars = tr.train_step()
    losses.append(float(total))
    assert all(np.isfinite(float(v)) for v in scalars.values())
  assert all(p.grad is not None and bool(torch.isfinite(p.grad).all())
             for p in tr.model.parameters() if p.requires_grad)
  assert losses[-1] < losses[0], losses
