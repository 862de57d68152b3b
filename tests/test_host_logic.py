"""Host-side mirror functions (torch elementwise / small-matrix helpers) against
the reference-generated goldens, on CPU tensors.  The HIP-backed ops
(forward_splat, splat, bilinear, scatter_add, the losses, compose) are covered
by the -m gpu tests; their torch restatements (oracle/lsi_torch_ref.py, the
gradient oracles of those tests) are pinned to the same goldens here."""
import numpy as np
import pytest
import torch

import lsi_oracle as O

from conftest import golden
import lsi_torch_ref as TR
from lsi.geometry import homography, ldi, projection
from lsi.loss import loss
from lsi.nnutils import helpers

T = torch.tensor


def close(a, b, rtol=1e-5, atol=1e-6):
  np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=rtol, atol=atol)


def test_helpers_known_answers():
  g = golden('known_answers.npz')
  close(helpers.zbuffer_weights(T(g['zbuf_in']), 50), g['zbuf_out_50'], 2e-6, 0)
  close(helpers.divide_safe(T(g['divsafe_num']), T(g['divsafe_den'])),
        g['divsafe_out'], 0, 0)
  np.testing.assert_array_equal(helpers.pixel_coords(2, 3, 5).numpy(),
                                g['pixel_coords_2_3_5'])
  assert float(helpers.zbuffer_weights(1e-3 / 0.4, scale=50)) > 0
  m = torch.rand(2, 3, 4)
  assert helpers.transpose(m).shape == (2, 4, 3)


def test_projection_matrices_match_reference_bitwise():
  for name in ('fs_kitti_L2_s05.npz', 'fs_cfg1_synth_L1_64.npz',
               'fs_general_L3_s05.npz', 'fs_kitti_inv_L1_s1.npz'):
    g = golden(name)
    m = projection.forward_projection_matrix(T(g['k_s']), T(g['k_t']),
                                             T(g['rot']), T(g['t']))
    close(m, g['M'], 1e-6, 1e-6)
  g = golden('layers.npz')
  close(projection.forward_projection_matrix(T(g['p_k_s']), T(g['p_k_t']),
                                             T(g['p_rot']), T(g['p_t'])),
        g['fwd_mat'], 1e-5, 1e-5)
  close(projection.inverse_projection_matrix(T(g['p_k_s']), T(g['p_k_t']),
                                             T(g['p_rot']), T(g['p_t'])),
        g['inv_mat'], 1e-5, 1e-5)
  k = torch.rand(2, 3, 3)
  assert projection.pad_intrinsic(k)[0, 3, 3] == 1
  assert projection.pad_extrinsic(k, torch.rand(2, 3, 1)).shape == (2, 4, 4)


def test_torch_oracle_compose_and_soft_z():
  g = golden('layers.npz')
  imgs, masks, dmaps = T(g['imgs']), T(g['masks']), T(g['dmaps'])
  close(TR.compose(imgs, masks, dmaps), g['compose_hard'])
  close(TR.compose(imgs, masks, dmaps, soft=True, min_disp=1e-3,
                       depth_softmax_temp=0.4), g['compose_soft'], 1e-4, 1e-6)
  close(TR.compose_depth(masks, dmaps), g['compose_depth'])
  close(TR.compose_depth(masks, dmaps, bg_layer=True, min_disp=1e-3,
                             depth_softmax_temp=0.4), g['compose_depth_bg'])
  close(helpers.soft_z_buffering(masks, dmaps, 0.4), g['soft_z'], 1e-4, 1e-7)
  np.testing.assert_array_equal(helpers.enforce_bg_occupied(masks).numpy(),
                                g['enforce_bg'])


def test_homography_algebra():
  g = golden('layers.npz')
  args = [T(g[k]) for k in ('p_k_s', 'p_k_t', 'p_rot', 'p_t')]
  n_hat, a = T(g['p_n_hat'][0]), T(g['p_a'][0])
  close(homography.inv_homography(*args, n_hat, a), g['inv_hom'], 1e-4, 1e-5)
  close(homography.inv_homography_dmat(args[1], args[2], args[3], n_hat, a),
        g['inv_hom_dmat'], 1e-4, 1e-6)
  nt, at = homography.transform_plane_eqns(args[2], args[3], n_hat, a)
  close(nt, g['plane_n_t'], 1e-5, 1e-6)
  close(at, g['plane_a_t'], 1e-5, 1e-6)
  pc = helpers.pixel_coords(2, 16, 20)
  pc = pc.unsqueeze(0).expand(2, 2, 16, 20, 3)
  rep = [x.unsqueeze(0).expand((2,) + tuple(x.shape)) for x in args]
  td = homography.trg_disp_maps(pc, rep[1], rep[2], rep[3], T(g['p_n_hat']),
                                T(g['p_a']))
  close(td, g['p_out_dmaps'], 1e-4, 1e-6)
  pts = torch.rand(2, 5, 3) + 0.5
  close(homography.normalize_homogeneous(pts),
        (pts[..., :2] / pts[..., 2:]).numpy())


def test_torch_oracle_losses():
  g = golden('losses.npz')
  imgs, masks, disps, trg = (T(g[k]) for k in ('imgs', 'masks', 'disps', 'trg'))
  got = TR.zbuffer_composition_loss(imgs, masks, disps, trg,
                                    bg_layer_disp=1e-3, max_disp=0.4,
                                    zbuf_scale=50)
  assert abs(float(got) - float(g['zbuf_comp_loss'])) < 1e-5 * float(g['zbuf_comp_loss'])
  assert abs(float(TR.decreasing_disp_loss(disps)) - float(g['decr_disp_loss'])) < 1e-6
  assert TR.decreasing_disp_loss(disps[:1]) == 0
  assert loss.decreasing_disp_loss(disps[:1]) == 0        # (L = 1: no kernel)
  assert abs(float(TR.disp_smoothness_loss(disps)) - float(g['smooth_loss'])) < 1e-5
  dx, dy = ldi.gradient(disps)
  np.testing.assert_array_equal(dx.numpy(), g['grad_dx'])
  np.testing.assert_array_equal(dy.numpy(), g['grad_dy'])
  probs, esc = loss.event_prob(masks)
  assert probs.shape == masks.shape and esc.shape[0] == 1
  assert float((probs.sum(0, keepdim=True) + esc - 1).abs().max()) < 1e-5


def test_torch_oracle_view_synthesis_loss():
  import lsi_oracle as O
  rs = np.random.RandomState(3)
  tgt = rs.rand(2, 16, 24, 3).astype(np.float32)
  recon = rs.rand(3, 2, 8, 12, 3).astype(np.float32)
  want = O.view_synthesis_loss(recon, tgt, 0.05)
  got = TR.view_synthesis_loss(T(recon), T(tgt), 0.05)
  assert abs(float(got) - float(want)) < 1e-6
  # decreasing_disp_loss stops the gradient of the nearer layer (loss.py:58-60)
  d = torch.rand(3, 1, 4, 4, 1, requires_grad=True)
  TR.decreasing_disp_loss(d).backward()
  assert float(d.grad[0].abs().sum()) == 0


def test_bilinear_taps_variant_has_no_cpu_path():
  # compose=False is a HIP kernel too (lsi_bilinear_taps; pinned on the GPU by
  # tests/test_sampling_gpu.py::test_bilinear_goldens): CPU tensors are an error
  from lsi.geometry import sampling
  g = golden('bilinear.npz')
  with pytest.raises(RuntimeError):
    sampling.bilinear(T(g['imgs']), T(g['coords']), compose=False)


@pytest.mark.parametrize('tag', ['compose', 'indep', 'full'])
def test_view_synthesis_loss_matches_the_reference_script_lines(tag):
  """tests/golden/view_synthesis.npz holds the output of the reference's own
  statements (ldi_enc_dec.py:337-351, executed in place by
  oracle/make_goldens.py): AREA resize, L1, mean over channels, min over
  layers, py2-rounded border crop.  The oracle's NumPy and torch restatements
  must both reproduce it."""
  import torch
  from lsi.loss import loss
  g = golden('view_synthesis.npz')
  recons, target = g[tag + '_recons'], g[tag + '_target']
  bdry = float(g[tag + '_bdry'])
  want = float(g[tag + '_loss'])
  got_oracle = float(O.view_synthesis_loss(recons, target, bdry))
  assert abs(got_oracle - want) <= 1e-6 * abs(want)
  got = float(TR.view_synthesis_loss(torch.tensor(recons), torch.tensor(target),
                                     bdry))
  assert abs(got - want) <= 1e-6 * abs(want)
  assert g[tag + '_pwise'].shape[1:] == (
      recons.shape[2] - 2 * loss._py2_round(recons.shape[2] * bdry),
      recons.shape[3] - 2 * loss._py2_round(recons.shape[3] * bdry))


def test_host_projection_matrices_equal_the_torch_ops_bit_for_bit(built_lib):
  """lsi_projection_matrices (csrc/lsi_host.hip; what forward_splat calls for
  host cameras) against the torch restatement of projection.py:71-106 (fp64
  adjugate inverse rounded once, sequential-k products): identical bits, both
  directions, intrinsics-like and general matrices."""
  from lsi.geometry import projection
  gen = torch.Generator().manual_seed(5)
  for trial in range(120):
    b = 1 + trial % 5
    f = torch.rand(b, generator=gen) * 900 + 100
    k = torch.zeros(b, 3, 3)
    k[:, 0, 0] = f
    k[:, 1, 1] = f * (0.9 + 0.2 * torch.rand(b, generator=gen))
    k[:, 0, 2] = torch.rand(b, generator=gen) * 800
    k[:, 1, 2] = torch.rand(b, generator=gen) * 300
    k[:, 2, 2] = 1
    if trial % 3 == 0:
      k = k + 0.1 * torch.randn(b, 3, 3, generator=gen)
    k2 = k * (0.8 + 0.4 * float(torch.rand(1, generator=gen)))
    rot = torch.linalg.qr(torch.randn(b, 3, 3, generator=gen))[0].contiguous()
    t = torch.randn(b, 3, 1, generator=gen)
    for fn in (projection.forward_projection_matrix,
               projection.inverse_projection_matrix):
      fast = fn(k, k2, rot, t)                       # B x 3 x 3 on the host: C
      slow = fn(k[None], k2[None], rot[None], t[None])[0]   # other ranks: torch
      assert fast.shape == (b, 4, 4) and torch.equal(fast, slow)
  assert projection._host_matrices(k.double(), k2, rot, t, False) is None


def test_source_coordinates_are_compared_with_the_whole_pixel_grid():
  """reference ldi.py:134 renders the caller's pixel_coords_src; the fused
  kernels generate the grid, so a tensor is compared with it as a whole (an
  interior that is not the grid must not pass on its corners) and anything else
  goes to the generic route."""
  disps = torch.zeros(2, 3, 4, 6, 1)
  grid = helpers.pixel_coords(3, 4, 6)
  assert ldi._is_pixel_grid(None, disps)
  assert ldi._is_pixel_grid(grid, disps)
  assert ldi._is_pixel_grid(grid, disps)          # (second call: remembered)
  with pytest.raises(ValueError):
    ldi._is_pixel_grid(helpers.pixel_coords(3, 4, 5), disps)
  assert not ldi._is_pixel_grid(grid + 0.25, disps)
  moved = grid.clone()
  assert ldi._is_pixel_grid(moved, disps)
  moved[1, 2, 3, 0] += 1.0     # an INTERIOR pixel, in place: the version moves
  assert not ldi._is_pixel_grid(moved, disps)
  third = grid.clone()
  third[..., 2] = 2.0          # the homogeneous coordinate counts as well
  assert not ldi._is_pixel_grid(third, disps)


def test_host_copy_of_cameras_follows_the_tensor_not_its_address():
  k = torch.eye(3).expand(2, 3, 3).contiguous()
  a = ldi._host_copy(k)
  assert a.dtype == torch.float32 and torch.equal(a, k)
  k64 = k.double()
  assert ldi._host_copy(k64).dtype == torch.float32
  # (the cache is only used for GPU tensors; CPU tensors are converted per call)
  assert not ldi._HOST_COPIES


def test_pack_layout_of_parameters():
  """Which parameter memory layouts lsi_conv2d_pack reads in place (the
  trainer's model is channels_last: its k x k parameters have channels-last
  strides, which round 5 once skipped) and which need a copy."""
  import torch
  from lsi.nnutils import _hip_conv
  w = torch.zeros((8, 4, 3, 3))
  assert _hip_conv._pack_layout(w) == 0
  wc = w.contiguous(memory_format=torch.channels_last)
  assert not wc.is_contiguous() and _hip_conv._pack_layout(wc) == 2
  assert _hip_conv._pack_layout(w.to(torch.bfloat16)) is None
  assert _hip_conv._pack_layout(w[:, :2]) is None            # a slice: neither layout
  assert _hip_conv._pack_layout(torch.zeros((8, 4, 1, 1)).contiguous(
      memory_format=torch.channels_last)) == 0               # 1 x 1: both at once
  src, cl = _hip_conv._pack_source(w[:, :2])
  assert cl == 0 and src.is_contiguous() and src.dtype == torch.float32
