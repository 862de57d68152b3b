"""Eager training harness (counterpart of the reference's
lsi/nnutils/train_utils.py `Trainer` template, without TF sessions).

Keeps the reference's template methods (define_data_loader, define_pred_graph
-> build_model, define_loss_graph -> compute_losses, feed, train, save), its
default flags, optimiser (Adam lr 1e-4, beta1 0.9: train_utils.py:107-117) and
checkpoint cadence (`latest` every save_latest_freq, numbered every
checkpoint_freq, keep 10, auto-resume: train_utils.py:172-232).  Multi-GPU is
plain data parallelism: one process per GPU, the minibatch sharded, gradients
all-reduced by DistributedDataParallel over RCCL -- and nothing else (batch norm
stays per replica, like the reference's single-GPU batch-4 statistics).
"""
import glob
import json
import os
import time

import torch


def define_default_flags(parser):
  """The reference's default trainer flags (train_utils.py:31-57)."""
  a = parser.add_argument
  a('--checkpoint_dir', default='cachedir/snapshots/')
  a('--pretrain_name', default='')
  a('--pretrain_iter', type=int, default=100000)
  a('--batch_size', type=int, default=2)
  a('--num_iter', type=int, default=100000)
  a('--img_height', type=int, default=256)
  a('--img_width', type=int, default=256)
  a('--log_freq', type=int, default=5)
  a('--checkpoint_freq', type=int, default=50000)
  a('--save_latest_freq', type=int, default=2000)
  a('--learning_rate', type=float, default=0.0001)
  a('--beta1', type=float, default=0.9)
  return parser


class Trainer(object):
  """Template-method trainer; subclasses provide data, model and losses."""

  def __init__(self, opts):
    self.opts = opts
    self.world = int(os.environ.get('WORLD_SIZE', '1'))
    self.rank = int(os.environ.get('RANK', '0'))
    self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    self.dist = None
    self.device = torch.device('cpu')
    self.global_step = 0

  # ---- to be provided by the experiment -----------------------------------
  def define_data_loader(self):
    raise NotImplementedError

  def build_model(self):
    """Returns the torch.nn.Module holding every trainable parameter."""
    raise NotImplementedError

  def compute_losses(self, batch):
    """Returns (total_loss, dict of named scalar losses)."""
    raise NotImplementedError

  def feed(self):
    raise NotImplementedError

  def stage(self, batch):
    """Moves one batch to the device.  Returns (staged, plan): `staged` is the
    list of device tensors compute_losses reads; `plan` is any hashable
    host-side decision derived from the batch (e.g. which kernel family renders
    it) that a captured HIP graph bakes in -- a change forces a re-capture."""
    return [t.to(self.device) for t in batch], None

  # ---- harness ---------------------------------------------------------------
  def setup(self, backend=None):
    opts = self.opts
    use_gpu = torch.cuda.is_available() and not getattr(opts, 'cpu', False)
    if use_gpu:
      torch.cuda.set_device(self.local_rank)
      self.device = torch.device('cuda', self.local_rank)
      # --miopen_search: let MIOpen time every solver once per layer (static
      # shapes; the step is bound by the convolutions: 16 % faster at batch 4,
      # 256 x 768).  Off by default: with ROCm 7.2 one of the candidate
      # igemm_bwd kernels the search tries was seen to fault (GPU memory access
      # fault inside MIOpen's search, depending on where buffers happen to lie).
      torch.backends.cudnn.benchmark = bool(getattr(opts, 'miopen_search', False))
      # --miopen_tune: MIOpen tunes its solvers for this run's layer shapes
      # (MIOPEN_FIND_ENFORCE=3: search, results kept in the user perf-db under
      # ~/.config/miopen) before the first step: 50-100 s once per machine, then
      # bf16 297 -> 373 samples/s (391 with --hip_graph), fp32 155 -> 182 at
      # batch 4, 256 x 768 -- the shipped perf-db has no entries for these
      # shapes.  Must be in the environment before the first convolution.
      if getattr(opts, 'miopen_tune', False):
        os.environ.setdefault('MIOPEN_FIND_ENFORCE', '3')
    # (a one-rank job launched by torch.distributed.run gets its process group
    # too: the collectives of the data-parallel step then run -- over one rank
    # -- which is how the step is exercised on a single-GPU box)
    if self.world > 1 or (getattr(opts, 'flat_grads', False) and
                          'RANK' in os.environ and 'MASTER_ADDR' in os.environ):
      import torch.distributed as dist
      if not dist.is_initialized():
        dist.init_process_group(backend or ('nccl' if use_gpu else 'gloo'))
      self.dist = dist
    torch.manual_seed(0)  # same initial weights on every rank
    self.define_data_loader()
    self.model = self.build_model().to(self.device)
    if getattr(opts, 'channels_last', False):
      self.model = self.model.to(memory_format=torch.channels_last)
      if use_gpu:
        from lsi.nnutils import nets  # pylint: disable=g-import-not-at-top
        nets.own_kernel_param_layouts(self.model)
    self.train_model = self.model
    # --hip_graph: the step (about 1500 kernels eagerly) is captured once into
    # HIP graphs and replayed.  One process: one graph (forward, losses,
    # backward, Adam).  Data parallel: torch's DDP hooks its bucketed
    # all-reduces into the backward, which a captured graph cannot hold; the
    # graphed step therefore keeps every gradient in ONE flat buffer
    # (`flat_grads`: the parameters' .grad are views of it), replays graph A
    # (zero, forward, losses, backward), all-reduces the buffer with one RCCL
    # call, replays graph B (Adam).  --flat_grads alone gives the same step
    # eagerly (no DDP wrapper, no overlap of the reduction with the backward:
    # 150 MB of gradients are ~1 ms over xGMI against a 19 ms step).
    self.use_graph = bool(getattr(opts, 'hip_graph', False)) and use_gpu
    self.flat_grads = bool(getattr(opts, 'flat_grads', False)) or \
        (self.use_graph and self.world > 1)
    if self.world > 1 and not self.flat_grads:
      from torch.nn.parallel import DistributedDataParallel as DDP
      self.train_model = DDP(
          self.model, device_ids=[self.local_rank] if use_gpu else None,
          bucket_cap_mb=25, gradient_as_bucket_view=True)
    self._graph, self._graph_plan, self._static = None, None, None
    self._graph_b = None
    self._graph_warm = 0
    # (fused: one multi-tensor kernel per step instead of the foreach chain --
    # 12 launches, 0.5 ms of a 14 ms bf16 step; same update rule,
    # train_utils.py:109-112)
    self.optim = torch.optim.Adam(self.model.parameters(),
                                  lr=opts.learning_rate,
                                  betas=(opts.beta1, 0.999), eps=1e-8,
                                  capturable=self.use_graph,
                                  fused=bool(use_gpu) and bool(getattr(opts, 'fused_adam', True)))
    if self.use_graph:
      self._stream = torch.cuda.Stream(self.device)
    if use_gpu:
      # weight gradients on a second stream, joined at the end of the backward
      # pass (_hip_conv.enable_wgrad_stream): single process, gradients stolen
      # by autograd -- not with DDP's reducer hooks or the flat gradient buffer,
      # which read a gradient on the main stream as soon as it is returned --
      # and eager launches: inside a captured HIP graph the fork / join edges
      # cost more than the overlap wins (4 layers, batch 4, 256 x 768: eager
      # 324 -> 336 samples/s, graphed 326 -> 319; profiles/r06/train_ab_*.txt).
      # LSI_WGRAD_STREAM=0 / 1 forces it off / on.
      from lsi.nnutils import _hip_conv  # pylint: disable=g-import-not-at-top
      env = os.environ.get('LSI_WGRAD_STREAM', '')
      _hip_conv.enable_wgrad_stream(
          self.world == 1 and not self.flat_grads and
          (env == '1' or (env != '0' and not self.use_graph)))
      # ... and the heads' per-layer decoders on a stream each (nets.HEAD_STREAMS)
      from lsi.nnutils import nets  # pylint: disable=g-import-not-at-top
      env = os.environ.get('LSI_HEAD_STREAMS', '')
      nets.enable_head_streams(
          self.world == 1 and not self.flat_grads and
          (env == '1' or (env != '0' and not self.use_graph)))
    self.resume()
    if self.flat_grads:
      self._setup_flat_grads()

  def _setup_flat_grads(self):
    """One fp32 buffer for every gradient; each parameter's .grad is a view of
    it (autograd accumulates in place into a defined .grad).  Parameters start
    identical on every rank (same seed / same checkpoint); rank 0's are
    broadcast anyway."""
    params = [p for p in self.model.parameters() if p.requires_grad]
    if self.dist is not None and self.world > 1:
      for p in params:
        self.dist.broadcast(p.data, src=0)
    n = sum(p.numel() for p in params)
    self._flat = torch.zeros((n,), dtype=params[0].dtype, device=self.device)
    off = 0
    for p in params:
      # (same strides as the parameter -- channels-last conv weights are dense
      # permuted blocks --: autograd accumulates in place, Adam stays fused)
      p.grad = torch.as_strided(self._flat, p.size(), p.stride(),
                                storage_offset=off)
      off += p.numel()

  def _reduce_flat(self):
    if self.dist is not None:
      self.dist.all_reduce(self._flat)
      if self.world > 1:
        self._flat.div_(self.world)

  @staticmethod
  def latest_checkpoint(checkpoint_dir):
    """The most recently written of model.latest / model-<step> (what
    tf.train.latest_checkpoint returns for the reference's Saver), or None."""
    cands = glob.glob(os.path.join(checkpoint_dir, 'model-*'))
    latest = os.path.join(checkpoint_dir, 'model.latest')
    if os.path.exists(latest):
      cands.append(latest)
    return max(cands, key=os.path.getmtime) if cands else None

  @staticmethod
  def optimistic_restore(model, state_dict):
    """helpers.optimistic_restorer (reference helpers.py:27-62): variables of the
    saved model that exist in this one with the same shape are restored, the
    others are skipped.  Returns the names restored."""
    own = model.state_dict()
    picked = {k: v for k, v in state_dict.items()
              if k in own and tuple(own[k].shape) == tuple(v.shape)}
    model.load_state_dict(picked, strict=False)
    return sorted(picked)

  @staticmethod
  def strict_restore(model, state_dict):
    """load_state_dict(strict=True), except that the constant moving statistics
    of the fully-connected stacks (SlimFC.moving_mean / moving_variance: buffers
    added to mirror the reference's variable list, always 0 / 1, never updated)
    may be missing from checkpoints written before they existed."""
    from lsi.nnutils import nets
    const = set()
    for name, mod in model.named_modules():
      if isinstance(mod, nets.SlimFC):
        const.update((name + '.moving_mean', name + '.moving_variance'))
    res = model.load_state_dict(state_dict, strict=False)
    missing = [k for k in res.missing_keys if k not in const]
    if missing or res.unexpected_keys:
      raise RuntimeError('checkpoint does not match the model: missing %s, '
                         'unexpected %s' % (missing, list(res.unexpected_keys)))

  def resume(self):
    """Latest checkpoint in checkpoint_dir if any; else, when --pretrain_name
    is given, a shape-tolerant warm start from
    checkpoint_dir/../<pretrain_name>/model-<pretrain_iter>
    (train_utils.py:176-200).  Like the reference, only model variables +
    global_step are saved: Adam's moments restart."""
    opts = self.opts
    path = self.latest_checkpoint(opts.checkpoint_dir)
    if path is not None:
      state = torch.load(path, map_location=self.device)
      self.strict_restore(self.model, state['model'])
      self.global_step = int(state['global_step'])
      return
    if getattr(opts, 'pretrain_name', ''):
      pre = os.path.join(os.path.dirname(os.path.normpath(opts.checkpoint_dir)),
                         opts.pretrain_name, 'model-%d' % opts.pretrain_iter)
      if not os.path.exists(pre):
        raise FileNotFoundError('--pretrain_name given but %s does not exist'
                                % pre)
      state = torch.load(pre, map_location=self.device)
      self.optimistic_restore(self.model, state['model'])

  def save(self, name):
    if self.rank != 0:
      return
    os.makedirs(self.opts.checkpoint_dir, exist_ok=True)
    path = os.path.join(self.opts.checkpoint_dir, name)
    torch.save({'model': self.model.state_dict(),
                'global_step': self.global_step}, path)
    numbered = sorted(glob.glob(os.path.join(self.opts.checkpoint_dir,
                                             'model-*')),
                      key=os.path.getmtime)
    for old in numbered[:-10]:  # max_to_keep=10
      os.remove(old)

  def _grad_part(self, staged):
    """Zero the gradients, forward, losses, backward."""
    if self.flat_grads:
      self._flat.zero_()
    else:
      self.optim.zero_grad(set_to_none=True)
    total, scalars = self.compute_losses(staged)
    total.backward()
    return total, scalars

  def _eager_step(self, staged):
    out = self._grad_part(staged)
    if self.flat_grads:
      self._reduce_flat()
    self.optim.step()
    self._after_update()
    return out

  def _after_update(self):
    """Nothing to do: the convolution kernels' packed weights follow the
    parameters through torch.optim's global post-step hook
    (_hip_conv._install_optimizer_hook: one lsi_conv2d_pack_many launch right
    after `optim.step()`, eagerly or inside the captured graph) -- any training
    loop gets that, not only this one."""

  def train_step(self):
    batch = self.feed()
    staged, plan = self.stage(batch)
    if not self.use_graph:
      out = self._eager_step(staged)
    else:
      out = self._graph_step(staged, plan)
    self.global_step += 1
    return out

  GRAPH_WARMUP = 3  # eager steps (MIOpen find, allocator) before capturing

  def _graph_step(self, staged, plan):
    """Eager for the first steps, then one captured HIP graph per plan."""
    cur = torch.cuda.current_stream(self.device)
    if self._graph is not None and plan == self._graph_plan:
      for dst, src in zip(self._static, staged):
        dst.copy_(src)
      return self._replay()
    self._stream.wait_stream(cur)
    with torch.cuda.stream(self._stream):
      if self._graph_warm < self.GRAPH_WARMUP:
        out = self._eager_step(staged)
        self._graph_warm += 1
      else:
        # capture: records the step, nothing runs until the replay below
        self._graph = None
        self._static = [t.clone() for t in staged]
        self._graph_plan = plan
        if not self.flat_grads:
          self.optim.zero_grad(set_to_none=True)
          graph = torch.cuda.CUDAGraph()
          with torch.cuda.graph(graph, stream=self._stream,
                                capture_error_mode='thread_local'):
            self._graph_out = self._eager_step(self._static)
        else:
          # graph A: gradients into the flat buffer; graph B: Adam.  The
          # all-reduce between them stays an ordinary RCCL call.
          graph = torch.cuda.CUDAGraph()
          # (thread_local: RCCL's watchdog thread queries its events meanwhile)
          with torch.cuda.graph(graph, stream=self._stream,
                                capture_error_mode='thread_local'):
            self._graph_out = self._grad_part(self._static)
          self._graph_b = torch.cuda.CUDAGraph()
          with torch.cuda.graph(self._graph_b, stream=self._stream,
                                pool=graph.pool(),
                                capture_error_mode='thread_local'):
            self.optim.step()
            self._after_update()
        self._graph = graph
        out = None
    cur.wait_stream(self._stream)
    if out is not None:
      return out
    return self._replay()

  def _replay(self):
    self._graph.replay()
    if self.flat_grads:
      self._reduce_flat()
      self._graph_b.replay()
    return self._graph_out

  def train(self, log_file=None):
    opts = self.opts
    if not hasattr(self, 'model'):
      self.setup()
    t0 = time.time()
    while self.global_step < opts.num_iter:
      total, scalars = self.train_step()
      step = self.global_step
      if step % opts.log_freq == 0 and self.rank == 0:
        rec = {'iter': step, 'total_loss': float(total),
               'time': time.time() - t0}
        rec.update({k: float(v) for k, v in scalars.items()})
        line = json.dumps(rec)
        print(line, flush=True)
        if log_file:
          with open(log_file, 'a') as f:
            f.write(line + '\n')
      if step % opts.save_latest_freq == 0:
        self.save('model.latest')
      if step % opts.checkpoint_freq == 0:
        self.save('model-%d' % step)
    return self
