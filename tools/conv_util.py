"""MFMA utilisation of the convolution path (SURVEY 8d "Convs"):
every convolution / transposed convolution the training step executes, timed on
its own at the step's shapes (forward; backward = data + weight gradient), with

    utilisation = MFMA FLOPs / (time x peak),   peak = 2.5 PF bf16 | 157.3 TF fp32

and the same figure for the whole training step (conv FLOPs of forward +
backward over the measured step time, tools/train_bench.py).

  python tools/conv_util.py [--bf16 true] [--batch_size 4] [--n_layers 2]
                            [--img_height 256 --img_width 768] [--step_ms T]

FLOPs are counted analytically: conv 2 * N * Hout * Wout * Cout * Cin * k * k;
transposed conv (4x4, stride 2) 2 * N * Hin * Win * Cin * Cout * 16; backward =
2x forward (data gradient + weight gradient; the first layer has no data
gradient).  One JSON document on stdout.
"""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'layered-scene-inference_amd'))
import ldi_enc_dec as script
from lsi.nnutils import nets, _hip_conv

PEAK = {True: 2.5e15, False: 157.3e12}


def main():
  argv = sys.argv[1:]
  step_ms = None
  if '--step_ms' in argv:
    i = argv.index('--step_ms'); step_ms = float(argv[i + 1]); del argv[i:i + 2]
  base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4',
          '--n_layers', '2', '--img_height', '256', '--img_width', '768',
          '--checkpoint_dir', '/tmp/lsi_ckpt']
  opts = script.apply_dataset_overrides(script.build_parser().parse_args(base + argv))
  dev = torch.device('cuda', 0)
  bf16 = bool(opts.bf16)
  dt = torch.bfloat16 if bf16 else torch.float32
  net = script.LdiNet(opts).to(dev).to(memory_format=torch.channels_last)
  # ---- every conv call of one forward pass of the step (both views in one
  # batch: --batched_pairs, the trainer's default): module, input shape --------
  calls = []

  def hook(mod, inp, out):
    calls.append((mod, tuple(inp[0].shape)))
  hs = [m.register_forward_hook(hook) for m in net.modules()
        if isinstance(m, (nets.SlimConv2d, nets.SlimConvTranspose2d))]
  names = {m: n for n, m in net.named_modules()}
  b = opts.batch_size * 2                       # source and target images
  imgs = torch.rand(b, opts.img_height, opts.img_width, 3, device=dev)
  with torch.no_grad(), torch.autocast('cuda', dtype=dt, enabled=bf16):
    net.predict(imgs)
  for h in hs:
    h.remove()

  def bench(fn, iters=20):
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters

  rows, tot_f, tot_b, t_f, t_b = [], 0.0, 0.0, 0.0, 0.0
  for idx, (mod, ishape) in enumerate(calls):
    n, cin, hi, wi = ishape
    transposed = isinstance(mod, nets.SlimConvTranspose2d)
    conv = mod.conv
    k = conv.kernel_size[0]
    x = torch.randn(ishape, device=dev, dtype=dt).contiguous(
        memory_format=torch.channels_last).requires_grad_(idx > 0)
    w = conv.weight.detach().to(dt).requires_grad_(True)
    if transposed:
      fwd = lambda: F.conv_transpose2d(x, w, None, conv.stride, conv.padding)
    else:
      # slim's SAME padding (nets.SlimConv2d.forward): symmetric for stride 1,
      # one more pixel after for stride 2
      ph = nets._same_pad(hi, k, mod.stride)
      pw = nets._same_pad(wi, k, mod.stride)
      if ph[0] == ph[1] and pw[0] == pw[1]:
        fwd = lambda: F.conv2d(x, w, None, mod.stride, (ph[0], pw[0]))
      else:
        fwd = lambda: F.conv2d(F.pad(x, (pw[0], pw[1], ph[0], ph[1])), w, None, mod.stride)
    with torch.no_grad():
      oshape = tuple(fwd().shape)
    _, cout, ho, wo = oshape
    flops = (2.0 * n * hi * wi * cin * cout * k * k if transposed else
             2.0 * n * ho * wo * cout * cin * k * k)
    y = fwd()
    g = torch.randn_like(y)
    tf = bench(lambda: fwd())

    def fb():
      yy = fwd()
      yy.backward(g)
      x.grad = None; w.grad = None
    tfb = bench(fb)
    tb = max(tfb - tf, 1e-9)
    bflops = flops * (2.0 if idx > 0 else 1.0)
    # the hand-written MFMA kernel (csrc/lsi_conv.hip) where it applies
    own = {}
    head = isinstance(mod, nets.SlimConv2d) and mod.bn is None and mod.activation == 'sigmoid'
    if (not transposed and bf16 and
        _hip_conv.supported(x.detach(), cin, cout, k, getattr(mod, 'stride', 1), head)):
      xd = x.detach()
      if head:
        bias = torch.zeros(cout, device=dev)
        t_own = bench(lambda: _hip_conv.conv3x3_c32_sigmoid(xd, w.detach(), bias))
      else:
        t_own = bench(lambda: _hip_conv.conv3x3_c32(xd, w.detach()))
      xg = x.detach().requires_grad_(True)
      def fb_own():
        yy = (_hip_conv.conv3x3_c32_sigmoid(xg, w, bias) if head
              else _hip_conv.conv3x3_c32(xg, w))
        yy.backward(g.to(yy.dtype))
        xg.grad = None; w.grad = None
      t_own_fb = bench(fb_own)
      own = {'own_fwd_us': t_own * 1e6, 'own_fwd_tflops': flops / t_own / 1e12,
             'own_fwd_mfma_util': flops / t_own / PEAK[bf16],
             'own_bwd_us': max(t_own_fb - t_own, 1e-9) * 1e6}
    elif (not transposed and bf16 and isinstance(mod, nets.SlimConv2d) and mod.bn is not None and
          _hip_conv.wgrad_supported(x.detach(), cin, cout, k, mod.stride)):
      # forward and data gradient on the library, weight gradient on lsi_conv3x3_wgrad
      xg = x.detach().requires_grad_(True)
      def fb_own():
        yy = _hip_conv.conv3x3_lib_own_wgrad(xg, w)
        yy.backward(g)
        xg.grad = None; w.grad = None
      t_own_fb = bench(fb_own)
      own = {'own_bwd_us': max(t_own_fb - tf, 1e-9) * 1e6, 'own': 'weight gradient only'}
    rows.append({**own, **{
        'layer': names.get(mod, '?'), 'in': list(ishape), 'out': list(oshape),
        'k': k, 'transposed': transposed, 'gflop_fwd': flops / 1e9,
        'fwd_us': tf * 1e6, 'fwd_tflops': flops / tf / 1e12,
        'fwd_mfma_util': flops / tf / PEAK[bf16],
        'bwd_us': tb * 1e6, 'bwd_tflops': bflops / tb / 1e12,
        'bwd_mfma_util': bflops / tb / PEAK[bf16]}})
    tot_f += flops; tot_b += bflops; t_f += tf; t_b += tb
  out = {
      'dtype': 'bf16' if bf16 else 'fp32', 'peak_flops': PEAK[bf16],
      'batch_per_pass': b, 'hw': [opts.img_height, opts.img_width],
      'n_layers': opts.n_layers, 'layers': rows,
      'sum': {'gflop_fwd': tot_f / 1e9, 'gflop_bwd': tot_b / 1e9,
              'conv_fwd_ms': t_f * 1e3, 'conv_bwd_ms': t_b * 1e3,
              'fwd_mfma_util': tot_f / t_f / PEAK[bf16],
              'bwd_mfma_util': tot_b / t_b / PEAK[bf16],
              'fwd_bwd_mfma_util': (tot_f + tot_b) / (t_f + t_b) / PEAK[bf16]}}
  if step_ms:
    out['step'] = {'ms': step_ms,
                   'mfma_util': (tot_f + tot_b) / (step_ms * 1e-3) / PEAK[bf16],
                   'conv_share_of_step': (t_f + t_b) * 1e3 / step_ms}
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
