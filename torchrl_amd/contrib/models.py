"""Policy / value networks of the four controllers (reference: torchrl/contrib/models.py).

These stay plain torch.nn modules: their matmuls run on PyTorch-ROCm (hipBLASLt / MFMA),
the one place of the learner where matrix cores are used.  What has to hold for drop-in
parity is (1) the parameter names, so a reference state_dict loads, (2) the order in which
parameters are created and re-initialised, so that under the same torch seed the weights
are bit-identical to the reference's, and (3) the forward arithmetic.  Hidden sizes are
arguments here (an int keeps the reference's depth, a tuple stacks more layers, e.g. the
"MLP 64x64" of the PPO benchmark); the defaults are the reference shapes.
"""
import os

import torch
from torch import nn
from torch.distributions import Normal, Categorical

from .. import functional as _F


class _SplitKLinearFn(torch.autograd.Function):
  """y = x W^T + b with a weight gradient that PyTorch-ROCm computes efficiently.

  dW[out, in] = dY[B, out]^T X[B, in] is a skinny GEMM whose reduction dimension is the batch.
  For B in the tens of thousands the BLAS picks a kernel without split-K and one workgroup
  walks the whole batch (measured on MI355X, B = 65536, 64x64: 187 us, against 17 us for
  the same product as a batched matmul over 64 slabs of the batch plus a sum).  Same
  arithmetic up to the order of the fp32 additions."""

  @staticmethod
  def forward(ctx, x, weight, bias):
    ctx.save_for_backward(x, weight)
    return torch.addmm(bias, x, weight.t())

  @staticmethod
  def backward(ctx, gy):
    x, weight = ctx.saved_tensors
    gy = gy.contiguous()
    B, n_out = gy.shape
    gx = gy @ weight if ctx.needs_input_grad[0] else None
    slabs = 64
    while slabs > 1 and (B % slabs or B // slabs < 256):
      slabs //= 2
    if slabs > 1:
      gw = torch.bmm(gy.view(slabs, B // slabs, n_out).transpose(1, 2),
                     x.reshape(slabs, B // slabs, x.shape[1])).sum(0)
    else:
      gw = gy.t() @ x
    return gx, gw, gy.sum(0)


class _LinearReluFn(torch.autograd.Function):
  """relu(x W^T + b) as ONE BLAS call (bias and ReLU in the GEMM's epilogue); the backward is
  what autograd runs for Linear + ReLU (mask, two products, bias sum).  Same values."""

  @staticmethod
  def forward(ctx, x, weight, bias):
    y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
    ctx.save_for_backward(x, weight, y)
    return y

  @staticmethod
  def backward(ctx, gy):
    x, weight, y = ctx.saved_tensors
    g = torch.ops.aten.threshold_backward(gy, y, 0.0)
    gx = g @ weight if ctx.needs_input_grad[0] else None
    gw = g.t() @ x if ctx.needs_input_grad[1] else None
    gb = g.sum(0) if ctx.needs_input_grad[2] else None
    return gx, gw, gb


class SkinnyLinear(nn.Linear):
  """nn.Linear (same parameters, same init, same forward values) whose backward splits the
  batch reduction of the weight gradient; used for large device batches only."""

  SPLIT_MIN_BATCH = 2048

  def forward(self, x):
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= self.SPLIT_MIN_BATCH and torch.is_grad_enabled() \
        and self.bias is not None:
      return _SplitKLinearFn.apply(x, self.weight, self.bias)
    return super().forward(x)


def _as_sizes(hidden, depth):
  """int -> `depth` hidden layers of that width; tuple/list -> as given."""
  if isinstance(hidden, (tuple, list)):
    return [int(h) for h in hidden]
  return [int(hidden)] * depth


class _Mlp(nn.Sequential):
  """nn.Sequential of Linear / ReLU (/ tail) modules -- same children, same state_dict keys --
  whose forward hands a Linear(+ReLU) pair to the fused HIP layer kernels (csrc/trl_mlp.hip)
  where those win: float32 device batches of at least FUSED_MIN_BATCH rows and layers up to
  FUSED_MAX_WIDTH wide (measured on MI355X at B = 65536: 64x64 forward 16 us vs 23 us, backward
  45 us vs ~55 us for torch with the split-K weight gradient; at 128 and above, and for small
  batches, torch's kernels are as fast or faster, so those layers stay on the torch modules).
  CPU tensors, fan-out > 256 and config.fused_mlp = False always take the plain modules."""

  FUSED_MIN_BATCH = 8192
  FUSED_MAX_WIDTH = 64

  def _plain(self, x):
    """The modules in order; on the device a Linear followed by ReLU is ONE BLAS call with the bias
    and the ReLU in its epilogue (same values, one launch less per layer: the small-batch
    learners are bound by their launch count)."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] < SkinnyLinear.SPLIT_MIN_BATCH
            and _fused_mlp_enabled()):
      return super().forward(x)
    mods, i = list(self), 0
    while i < len(mods):
      m = mods[i]
      if type(m) in (nn.Linear, SkinnyLinear) and m.bias is not None and i + 1 < len(mods) and \
          isinstance(mods[i + 1], nn.ReLU):
        x = _LinearReluFn.apply(x, m.weight, m.bias)
        i += 2
      else:
        x = m(x)
        i += 1
    return x

  def _u8_head(self, x):
    """uint8 input (frames the replay ring keeps uint8; the reference holds the same values as
    float32, storage.py:65-67): the first Linear(+ReLU) reads them as uint8 where the HIP layer
    takes the shape (csrc/trl_linear_u8.hip), the rest of the stack follows as usual.  Any other
    uint8 batch is cast first."""
    mods = list(self)
    m = mods[0] if mods else None
    if type(m) in (nn.Linear, SkinnyLinear) and m.bias is not None and _fused_mlp_enabled() and \
        _F.linear_u8_supported(x, m.weight):
      relu = len(mods) > 1 and isinstance(mods[1], nn.ReLU)
      h = _F.linear_u8(x, m.weight, m.bias, relu=relu)
      rest = mods[2:] if relu else mods[1:]
      return _Mlp(*rest)(h) if rest else h
    return self.forward(x.float())

  def forward(self, x):
    if x.dtype == torch.uint8:
      return self._u8_head(x)
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= self.FUSED_MIN_BATCH
            and _fused_mlp_enabled()):
      return self._plain(x)
    mods = list(self)
    whole = self._whole_net(mods, x)
    if whole is not None:
      return whole
    i = 0
    while i < len(mods):
      m = mods[i]
      if isinstance(m, nn.Linear) and m.bias is not None and \
          max(m.in_features, m.out_features) <= self.FUSED_MAX_WIDTH:
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        x = _F.linear_act(x, m.weight, m.bias, relu=relu)
        i += 2 if relu else 1
      else:
        x = m(x)
        i += 1
    return x


def _fusable_stack(mods):
  """(linear layers, trailing modules) if `mods` is Linear/ReLU alternation of <= 3 Linear layers,
  all <= 64 wide, followed only by non-Linear, non-ReLU modules (Tanh, Softmax); else None."""
  lin, i = [], 0
  while i < len(mods) and isinstance(mods[i], nn.Linear):
    lin.append(mods[i])
    i += 1
    if i < len(mods) and isinstance(mods[i], nn.ReLU) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.Linear):
      i += 1
    else:
      break
  tail = mods[i:]
  if not (1 <= len(lin) <= 3) or any(isinstance(m, (nn.Linear, nn.ReLU)) for m in tail):
    return None
  if any(m.bias is None or max(m.in_features, m.out_features) > 64 for m in lin):
    return None
  return lin, tail


def _direct_ok(stack, lin):
  return getattr(stack, 'direct_grads', False) and torch.is_grad_enabled() and \
      all(m.weight.grad is not None and m.bias.grad is not None for m in lin)


def _whole_net(self, mods, x):
  """One fused forward launch (and one backward launch) for the whole stack when it qualifies
  and its input needs no gradient; trailing modules are applied by torch afterwards."""
  fs = None if x.requires_grad else _fusable_stack(mods)
  if fs is None:
    return None
  lin, tail = fs
  y = _F.mlp64(x, [m.weight for m in lin], [m.bias for m in lin], direct_grads=_direct_ok(self, lin))
  for m in tail:
    y = m(y)
  return y


def run_stacks(stacks, xs):
  """Outputs of several _Mlp stacks on same-sized batches.  When every one qualifies for the
  whole-network kernels they share ONE forward launch and ONE backward launch (the critic and
  the actor of an actor-critic, online and target Q-networks); otherwise each runs on its own."""
  x0 = xs[0]
  ok = (len(stacks) <= 4 and _fused_mlp_enabled() and x0.is_cuda and x0.dim() == 2 and x0.dtype == torch.float32
        and x0.shape[0] >= _Mlp.FUSED_MIN_BATCH
        and all(x.shape[0] == x0.shape[0] and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
                for x in xs))
  fss = [(_fusable_stack(list(s)) if ok and isinstance(s, _Mlp) else None) for s in stacks]
  if not ok or any(f is None for f in fss):
    return [s(x) for s, x in zip(stacks, xs)]
  direct = [_direct_ok(s, f[0]) for s, f in zip(stacks, fss)]
  if any(direct) and not all(direct):      # mixed: keep it simple, one launch pair each
    return [s(x) for s, x in zip(stacks, xs)]
  ys = _F.mlp64_group(xs, [([m.weight for m in lin], [m.bias for m in lin]) for lin, _ in fss],
                      direct_grads=all(direct))
  out = []
  for y, (_, tail) in zip(ys, fss):
    for m in tail:
      y = m(y)
    out.append(y)
  return out


_Mlp._whole_net = _whole_net


def _fused_mlp_enabled():
  from .. import config
  return config.fused_mlp


def _mlp(n_in, hidden_sizes, n_out, tail=None):
  """Linear/ReLU stack whose Linear layers sit at even Sequential indices (0, 2, 4, ...)."""
  layers, width = [], n_in
  for h in hidden_sizes:
    layers += [SkinnyLinear(width, h), nn.ReLU()]
    width = h
  layers.append(SkinnyLinear(width, n_out))
  if tail is not None:
    layers.append(tail)
  return _Mlp(*layers)


def _last_linear(seq):
  return [m for m in seq if isinstance(m, nn.Linear)][-1]


class QNet(nn.Module):
  """obs -> Q(s, .): in -> 128 -> 128 -> out (models.py:6-24)."""

  def __init__(self, input_size, output_size, hidden=128):
    super().__init__()
    self._input_size, self._output_size = input_size, output_size
    self._hidden_sizes = _as_sizes(hidden, 2)
    self._hidden_size = self._hidden_sizes[0]
    self.net = _mlp(input_size, self._hidden_sizes, output_size)

  def forward(self, obs):
    return self.net(obs)


class DDPGActorNet(nn.Module):
  """obs -> tanh action; last layer U(-3e-3, 3e-3) (models.py:27-53)."""

  def __init__(self, state_size, action_size, hidden_size):
    super().__init__()
    self.state_size, self.action_size, self.hidden_size = state_size, action_size, hidden_size
    self._weight_init = 3e-3
    self.net = _mlp(state_size, _as_sizes(hidden_size, 2), action_size, tail=nn.Tanh())
    out = _last_linear(self.net)
    nn.init.uniform_(out.weight, -self._weight_init, self._weight_init)
    nn.init.uniform_(out.bias, -self._weight_init, self._weight_init)

  def forward(self, obs):
    return self.net(obs)


class DDPGCriticNet(nn.Module):
  """(obs, action) -> Q; last layer U(-3e-3, 3e-3) (models.py:56-81)."""

  def __init__(self, state_size, action_size, hidden_size):
    super().__init__()
    self.state_size, self.action_size, self.hidden_size = state_size, action_size, hidden_size
    self._weight_init = 3e-3
    self.net = _mlp(state_size + action_size, _as_sizes(hidden_size, 2), 1)
    out = _last_linear(self.net)
    nn.init.uniform_(out.weight, -self._weight_init, self._weight_init)
    nn.init.uniform_(out.bias, -self._weight_init, self._weight_init)

  def forward(self, obs, action):
    return self.net(torch.cat([obs, action], dim=1))


class A2CNet(nn.Module):
  """Separate critic and softmax actor, one hidden layer each (models.py:84-109).

  `forward` returns (value, Categorical); `heads` returns the raw (value, probs) pair that
  the fused A2C loss kernel consumes.
  """

  def __init__(self, input_size, output_size, hidden_size):
    super().__init__()
    self._input_size, self._output_size, self._hidden_size = input_size, output_size, hidden_size
    sizes = _as_sizes(hidden_size, 1)
    self.critic = _mlp(input_size, sizes, 1)                                   # created first
    self.actor = _mlp(input_size, sizes, output_size, tail=nn.Softmax(dim=1))

  def heads(self, obs):
    value, probs = run_stacks([self.critic, self.actor], [obs, obs])
    return value, probs

  def forward(self, obs):
    value, probs = self.heads(obs)
    return value, Categorical(probs)


class ActorCriticNet(nn.Module):
  """Gaussian policy + critic; every Linear W ~ N(0, 0.1), b = 0.1 (models.py:112-147).

  `heads` returns (value, mean, log_std) for the fused PPO kernels.
  """

  def __init__(self, input_size, output_size, hidden_size, std=0.0):
    super().__init__()
    self._input_size, self._output_size, self._hidden_size = input_size, output_size, hidden_size
    sizes = _as_sizes(hidden_size, 1)
    self.critic = _mlp(input_size, sizes, 1)
    self.actor = _mlp(input_size, sizes, output_size)
    self.log_std = nn.Parameter(torch.ones(1, output_size) * std)
    self.apply(self.init_weights)   # module order = creation order: critic.*, then actor.*

  def heads(self, obs):
    value, mean = run_stacks([self.critic, self.actor], [obs, obs])
    return value, mean, self.log_std

  def forward(self, obs):
    value, mean, log_std = self.heads(obs)
    return value, Normal(mean, log_std.exp().expand_as(mean))

  @staticmethod
  def init_weights(module):
    if isinstance(module, nn.Linear):
      nn.init.normal_(module.weight, mean=0., std=0.1)
      nn.init.constant_(module.bias, 0.1)
