"""Flat parameter/gradient buffers and the one-launch Adam update.

The reference calls `torch.optim.Adam(module.parameters(), lr=...)` with default betas/eps
(dqn_controller.py:39, ddpg_controller.py:73,80, a2c_controller.py:14, ppo_controller.py:15).
Here every parameter of a module is a view into ONE contiguous float32 buffer and every
`.grad` a view into a second one, so that
  * zero_grad is one memset,
  * the data-parallel all-reduce is one collective on one buffer (parallel.py),
  * clip_grad_value_ + Adam are one HIP kernel launch (csrc/trl_loss.hip: k_adam),
  * hard target sync / Polyak averaging are one launch over the flat buffers.
The module keeps working as an ordinary nn.Module (state_dict, load_state_dict, parameters).
"""
import torch

from . import _lib
from . import functional as F


class FlatParams:
  """Re-homes `module`'s parameters (in `parameters()` order) into flat buffers."""

  def __init__(self, module, fresh_grads=False):
    """fresh_grads: zero_grad() detaches the parameters' .grad instead of zeroing the flat buffer,
    autograd then hands each parameter a fresh gradient tensor (no per-parameter accumulate
    kernel), and collect_grads() moves them into the flat buffer with ONE multi-tensor copy before
    the all-reduce / update.  For small-batch learners, where a step is bound by its number of
    launches (a 6-parameter network saves 6 adds + 1 fill - 1 copy per backward).  Not for stacks
    whose fused backward writes .grad in place (direct_grads)."""
    self.module = module
    self.fresh_grads = bool(fresh_grads)
    self._pending = False
    self.params = [p for p in module.parameters()]
    if not self.params:
      raise ValueError('module has no parameters')
    dev = self.params[0].device
    for p in self.params:
      if p.dtype != torch.float32 or p.device != dev:
        raise TypeError('FlatParams needs float32 parameters on one device')
    # every parameter starts on a 16 B boundary (kernels then take 16 B loads of weight rows);
    # the padding floats carry zero weights / zero gradients and never move
    self.offsets, n = [], 0
    for p in self.params:
      self.offsets.append(n)
      n += (p.numel() + 3) // 4 * 4
    self.data = torch.zeros(n, dtype=torch.float32, device=dev)
    self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
    with torch.no_grad():
      for p, off in zip(self.params, self.offsets):
        k = p.numel()
        self.data[off:off + k].copy_(p.data.reshape(-1))
        p.data = self.data[off:off + k].view(p.shape)
        p.grad = self.grad[off:off + k].view(p.shape)
    self.numel = n
    self.device = dev

  def dense(self, grad=False):
    """The parameters (or gradients) concatenated without the alignment padding, in
    `module.parameters()` order: what torch.cat over the reference's parameters gives."""
    buf = self.grad if grad else self.data
    return torch.cat([buf[off:off + p.numel()] for p, off in zip(self.params, self.offsets)])

  def zero_grad(self):
    """Zero in place: the .grad views must survive (autograd then accumulates into them)."""
    if self.fresh_grads:
      for p in self.params:
        p.grad = None
      self._pending = True
      return
    self.grad.zero_()
    for p, off in zip(self.params, self.offsets):   # re-attach a view a caller dropped or replaced
      if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
        p.grad = self.grad[off:off + p.numel()].view(p.shape)

  def collect_grads(self):
    """fresh_grads mode: the gradients autograd produced since zero_grad() -> the flat buffer
    (one launch; a parameter that got none is zeroed), and the .grad views are re-attached."""
    if not self._pending:
      return
    self._pending = False
    views = [self.grad[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets)]
    have = [(v, p.grad) for v, p in zip(views, self.params) if p.grad is not None]
    for v, p in zip(views, self.params):
      if p.grad is None:
        v.zero_()
    if have:
      torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
    for v, p in zip(views, self.params):
      p.grad = v

  def copy_from(self, other):
    """Hard sync: load_state_dict of an identically shaped module (dqn_controller.py:86)."""
    self.data.copy_(other.data)

  def polyak_from(self, source, tau):
    F.polyak_(self.data, source.data, tau)


class FlatAdam:
  """torch.optim.Adam(lr=lr) semantics (betas (0.9, 0.999), eps 1e-8, no weight decay, no
  amsgrad) over a FlatParams, as one kernel launch per step.

  grad_scale / clip_value are applied to the gradient inside the same launch (1/world_size
  after the all-reduce; torch.nn.utils.clip_grad_value_).  With device_state=True the step
  count and hyper-parameters live in a device struct (trl_adam_state) and the launch derives
  the bias corrections there, which is what a captured HIP graph needs (kernel arguments are
  frozen at capture, the step count is not)."""

  def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, clip_value=0.0,
               device_state=False):
    self.flat = flat
    self.lr, self.betas, self.eps = lr, betas, eps
    self.grad_scale, self.clip_value = grad_scale, clip_value
    self.exp_avg = torch.zeros_like(flat.data)
    self.exp_avg_sq = torch.zeros_like(flat.data)
    self.step_count = 0
    self.device_state = device_state
    self._state_dev = None

  def zero_grad(self):
    self.flat.zero_grad()

  def _push_state(self):
    """(Re)write the device struct from the host-side values."""
    st = _lib.AdamState(step=self.step_count, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                        eps=self.eps, grad_scale=self.grad_scale, clip_value=self.clip_value)
    host = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8)
    if self._state_dev is None:
      self._state_dev = torch.empty(host.numel() + 8, dtype=torch.uint8, device=self.flat.device)
      off = (-self._state_dev.data_ptr()) % 8
      self._state_dev = self._state_dev[off:off + host.numel()]
    self._state_dev.copy_(host)

  def state_tensors(self):
    """Device tensors a step mutates (for GraphedStep's warm-up restore)."""
    out = [self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq]
    if self.device_state:
      if self._state_dev is None:
        self._push_state()
      out.append(self._state_dev)
    return out

  def step(self):
    """Device work only (safe inside a captured graph); see note_step for the host mirror."""
    if not self.flat.data.is_cuda:
      raise _lib.TrlError('FlatAdam.step needs a HIP device (no CPU update path)')
    self.flat.collect_grads()
    if self.device_state:
      if self._state_dev is None:
        self._push_state()
      with torch.cuda.device(self.flat.device):
        _lib.check(_lib.lib().trl_adam_step_dev(self.flat.data.data_ptr(), self.flat.grad.data_ptr(),
                                                self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                self.flat.numel, self._state_dev.data_ptr(),
                                                _lib.stream_ptr(self.flat.device)))
    else:
      self.step_count += 1
      F.adam_step_(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.step_count,
                   self.lr, self.betas, self.eps, self.grad_scale, self.clip_value)

  def fused_args(self, update=True, allow_device_state=False):
    """trl_adam_args for a kernel that applies this update itself (trl_ppo_learn_fused,
    trl_ddpg_learn_fused, trl_dqn_learn_fused): advances the host step count, like step().  Not for
    device_state (graph) mode.  update=False: the block with step = 0, which those kernels read as
    "gradients only" (the caller all-reduces and steps afterwards); the count is not advanced."""
    dev = self.device_state and update
    if dev and not allow_device_state:
      raise _lib.TrlError('this fused update is for the eager step (graph mode keeps its step count on the device)')
    if dev and self._state_dev is None:
      self._push_state()
    if update and not dev:
      self.step_count += 1
    a = getattr(self, '_fused_args', None)
    if a is None:
      a = self._fused_args = _lib.AdamArgs()
    # device_state: the chain advances the DEVICE step count itself (note_step mirrors it on the host)
    a.dev_state = self._state_dev.data_ptr() if dev else None
    a.param, a.grad = self.flat.data.data_ptr(), self.flat.grad.data_ptr()
    a.exp_avg, a.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
    a.n = self.flat.numel
    a.lr, a.beta1, a.beta2, a.eps = self.lr, self.betas[0], self.betas[1], self.eps
    a.step, a.grad_scale, a.clip_value = (self.step_count if update and not dev else 0), self.grad_scale, self.clip_value
    return a

  def note_step(self):
    """Host mirror of the device step counter: call once per optimizer step when
    device_state=True (the step itself may have been a graph replay)."""
    if self.device_state:
      self.step_count += 1

  def state_dict(self):
    return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                lr=self.lr, betas=self.betas, eps=self.eps)

  def load_state_dict(self, sd):
    self.step_count = int(sd['step'])
    self.exp_avg.copy_(sd['exp_avg'])
    self.exp_avg_sq.copy_(sd['exp_avg_sq'])
    self.lr, self.betas, self.eps = sd['lr'], tuple(sd['betas']), sd['eps']
    if self.device_state:
      self._push_state()
