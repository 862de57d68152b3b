"""Are the packed weights of the implicit-GEMM layers current in a training run?
Prints, per step, a layer's parameter version and whether its packed operand
equals a fresh pack of the parameter's present values."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'layered-scene-inference_amd'))
import ldi_enc_dec as script
from lsi.nnutils import _hip_conv
from lsi import _C
base = ['--dataset', 'kitti', '--kitti_procedural', 'true', '--batch_size', '4', '--n_layers', '2',
        '--img_height', '256', '--img_width', '768', '--checkpoint_dir', '/tmp/lsi_ckpt',
        '--save_latest_freq', '1000000', '--checkpoint_freq', '1000000', '--log_freq', '1000000', '--bf16', 'true'] + sys.argv[1:]
opts = script.apply_dataset_overrides(script.build_parser().parse_args(base))
tr = script.Trainer(opts); tr.setup()
def fresh(e, w):
  lib = _C.lib()
  buf = torch.empty_like(e.buf)
  rc = lib.lsi_conv2d_pack(ctypes.byref(e.desc), e.mode, _C.ptr(w.detach().float().contiguous()), _C.ptr(buf),
                           buf.numel() * 2, _C.stream_ptr(w.device))
  assert rc == 0
  return buf
before = {}
for i in range(4):
  tr.train_step()
  torch.cuda.synchronize()
  stale = 0
  for k, e in _hip_conv._PACKED.items():
    w = e.wref()
    # the operand this step's forward used against a pack of the weights as they
    # were BEFORE this step's update
    # (the trainer re-packs after the optimiser's update: the packs are those of
    # the present weights)
    if w is not None and not torch.equal(fresh(e, w), e.buf):
      stale += 1
  before = {k: e.wref().detach().clone() for k, e in _hip_conv._PACKED.items() if e.wref() is not None}
  e = next(iter(_hip_conv._PACKED.values())); w = e.wref()
  print('step', i, 'version', w._version, 'packed version', e.version, 'contiguous', w.is_contiguous(),
        'layers whose pack is not of the present weights: %d of %d' % (stale, len(_hip_conv._PACKED)))
