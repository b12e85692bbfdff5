"""Torch-facing wrappers of the HIP kernels (C ABI: include/trl_hip.h).

Every function here takes device tensors and enqueues kernels on torch's current stream.
There is no CPU implementation behind any of them: a CPU tensor raises TrlError.
The loss functions are autograd Functions whose forward runs the fused kernel (loss AND
its backward seeds in one pass) and whose backward only scales the saved seeds.
"""
import torch

from . import _lib
from ._lib import TRL_GAE_MASKED, TRL_GAE_REFERENCE, TRL_LAYOUT_NT, TRL_LAYOUT_TN, TRL_LOSS_HUBER, TRL_LOSS_MSE

_workspaces = {}


def _ws(device):
  """Per-device reduction scratch of the loss kernels (allocated once)."""
  key = (device.index if device.index is not None else torch.cuda.current_device())
  ws = _workspaces.get(key)
  if ws is None:
    ws = torch.empty(_lib.lib().trl_loss_workspace_bytes(), dtype=torch.uint8, device=device)
    _workspaces[key] = ws
  return ws


def _f32c(t, what):
  _lib.require_device(t, what)
  if t.dtype != torch.float32:
    raise TypeError('%s must be float32, got %s' % (what, t.dtype))
  return t.contiguous()


# ------------------------------------------------------------------------------ GAE
def gae(reward, value, bootstrap, gamma, lmbda, done=None, masked=False, env_major_out=False,
        want_adv=True, out=None):
  """Reverse-scan GAE over [T, N] (a2c_controller.py:38-43, ppo_controller.py:41-46).

  reward, value: [T, N] float32 on the device; bootstrap: [N] = V(next_obs[T-1]) or 0 where
  the last step was terminal.  Returns (returns, adv) shaped [T, N], or [N, T] with
  env_major_out (the concatenation order of examples/ppo.py:61-64).  `masked=True` applies
  (1 - done) at every step (extension; the reference ignores mid-trajectory dones).
  """
  reward = _f32c(reward, 'reward')
  value = _f32c(value, 'value')
  bootstrap = _f32c(bootstrap, 'bootstrap')
  if reward.dim() != 2 or value.shape != reward.shape or bootstrap.numel() != reward.shape[1]:
    raise ValueError('expected reward/value [T, N] and bootstrap [N], got %s %s %s' %
                     (tuple(reward.shape), tuple(value.shape), tuple(bootstrap.shape)))
  T, N = reward.shape
  if masked:
    if done is None:
      raise ValueError('masked GAE needs done')
    done = _f32c(done, 'done')
    if done.shape != reward.shape:
      raise ValueError('done must be [T, N]')
  shape = (N, T) if env_major_out else (T, N)
  if out is not None:           # caller-owned outputs (no allocation on the hot path)
    returns, adv = out
    for o in (returns, adv):
      if o is not None and (o.shape != shape or o.dtype != torch.float32 or not o.is_contiguous()
                            or o.device != reward.device):
        raise ValueError('gae: out tensors must be contiguous float32 %s on %s' % (shape, reward.device))
  else:
    returns = torch.empty(shape, dtype=torch.float32, device=reward.device)
    adv = torch.empty(shape, dtype=torch.float32, device=reward.device) if want_adv else None
  with torch.cuda.device(reward.device):
    _lib.check(_lib.lib().trl_gae_reverse_scan(
        reward.data_ptr(), value.data_ptr(), _lib.ptr(done) if masked else None, bootstrap.data_ptr(),
        T, N, float(gamma), float(lmbda), TRL_GAE_MASKED if masked else TRL_GAE_REFERENCE,
        TRL_LAYOUT_NT if env_major_out else TRL_LAYOUT_TN, returns.data_ptr(), _lib.ptr(adv),
        _lib.stream_ptr(reward.device)))
  return returns, adv


def _blocked_plane(t, what):
  """A [G, T, E] float32 device plane whose env groups are contiguous [T, E] runs: strides
  (t_stride * E, E, 1) with t_stride >= T (a [:, :T] slice of a [G, capacity, E] buffer qualifies).
  Returns t_stride."""
  _lib.require_device(t, what)
  if t.dtype != torch.float32 or t.dim() != 3:
    raise TypeError('%s must be a float32 [N / E, T, E] tensor, got %s %s' % (what, t.dtype, tuple(t.shape)))
  G, T, E = t.shape
  if E not in (8, 16):
    raise ValueError('%s: the blocked layout holds 8 or 16 envs per group, got %d' % (what, E))
  if G * T == 0:
    return T
  # (the stride of a size-1 dimension is arbitrary in torch)
  if t.stride(2) != 1 or (T > 1 and t.stride(1) != E) or (G > 1 and (t.stride(0) % E or t.stride(0) < T * E)):
    raise ValueError('%s: expected strides (t_stride * E, E, 1), got %s' % (what, t.stride()))
  return t.stride(0) // E if G > 1 else T


def gae_blocked(reward, value, bootstrap, gamma, lmbda, done=None, masked=False, env_major_out=False,
                want_adv=True, out=None):
  """`gae` over a rollout kept in the blocked layout (C ABI trl_gae_reverse_scan_blocked).

  reward, value (and done when masked): [N / E, T, E] float32, E = 8 or 16 consecutive envs per group,
  group g's [T, E] run contiguous (RolloutBuffer(layout='blocked').tensors(), or the value kernel's
  output over blocked observations); bootstrap: [N].  Same arithmetic and bit-identical results as
  `gae` on the [T, N] transposition.  Returns (returns, adv): [N / E, T, E], or env-major [N, T] with
  env_major_out."""
  ts = _blocked_plane(reward, 'reward')
  if value.shape != reward.shape or _blocked_plane(value, 'value') != ts:
    raise ValueError('gae_blocked: value must have reward\'s shape and strides')
  G, T, E = reward.shape
  N = G * E
  bootstrap = _f32c(bootstrap, 'bootstrap')
  if bootstrap.numel() != N:
    raise ValueError('gae_blocked: bootstrap must be [%d]' % N)
  if masked:
    if done is None:
      raise ValueError('masked GAE needs done')
    if done.shape != reward.shape or _blocked_plane(done, 'done') != ts:
      raise ValueError('gae_blocked: done must have reward\'s shape and strides')
  shape = (N, T) if env_major_out else (G, T, E)
  if out is not None:
    returns, adv = out
    for o in (returns, adv):
      if o is not None and (o.shape != shape or o.dtype != torch.float32 or not o.is_contiguous()
                            or o.device != reward.device):
        raise ValueError('gae_blocked: out tensors must be contiguous float32 %s on %s' % (shape, reward.device))
  else:
    returns = torch.empty(shape, dtype=torch.float32, device=reward.device)
    adv = torch.empty(shape, dtype=torch.float32, device=reward.device) if want_adv else None
  with torch.cuda.device(reward.device):
    _lib.check(_lib.lib().trl_gae_reverse_scan_blocked(
        reward.data_ptr(), value.data_ptr(), _lib.ptr(done) if masked else None, bootstrap.data_ptr(),
        T, N, E, ts, float(gamma), float(lmbda), TRL_GAE_MASKED if masked else TRL_GAE_REFERENCE,
        TRL_LAYOUT_NT if env_major_out else _lib.TRL_LAYOUT_BLOCKED, 0, returns.data_ptr(), _lib.ptr(adv),
        _lib.stream_ptr(reward.device)))
  return returns, adv


def to_blocked(x, block_envs=16):
  """[T, N, ...] -> the blocked layout [N / E, T, E, ...] (a copy)."""
  T, N = x.shape[:2]
  if N % block_envs:
    raise ValueError('N=%d is not a multiple of block_envs=%d' % (N, block_envs))
  rest = tuple(x.shape[2:])
  perm = (1, 0, 2) + tuple(range(3, 3 + len(rest)))
  return x.reshape((T, N // block_envs, block_envs) + rest).permute(perm).contiguous()


def from_blocked(x):
  """[N / E, T, E, ...] -> [T, N, ...] (a copy)."""
  G, T, E = x.shape[:3]
  rest = tuple(x.shape[3:])
  perm = (1, 0, 2) + tuple(range(3, 3 + len(rest)))
  return x.permute(perm).reshape((T, G * E) + rest).contiguous()


# ------------------------------------------------------------------------------ exploration on the device
class ActRng:
  """Seed + call counter of a controller's on-device exploration (csrc/trl_act.hip: Philox keyed by the seed,
  counter = (call number, row)); the seed defaults to torch.initial_seed() at the first draw, mixed with a stream id
  that is different for every ActRng of the process and for every rank of a data-parallel job -- two agents, or N
  ranks seeded alike, must not explore in lock-step (the reference's consumers draw one after the other from one
  numpy generator and so never see the same numbers either).  An explicit `seed` is used as given."""
  _instances = 0

  def __init__(self, seed=None, rank=None):
    self.seed, self.calls = seed, 0
    self._rank = rank
    self._instance = ActRng._instances
    ActRng._instances += 1

  @staticmethod
  def _mix(x):      # splitmix64 finalizer: distinct inputs -> well-separated Philox keys
    m = 2 ** 64 - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)

  def next(self):
    if self.seed is None:
      rank = self._rank
      if rank is None:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
      self.seed = self._mix(int(torch.initial_seed()) ^ self._mix((rank << 32) | self._instance))
    self.calls += 1
    return self.seed & (2 ** 64 - 1), self.calls - 1


def act_epsilon_greedy(q, eps, rng):
  """q [n, A] float32 -> int64 actions [n]: greedy w.p. 1 - eps + eps / A, any other action w.p. eps / A
  (the categorical of dqn_controller.py:16-22), one launch, no host round trip."""
  q = _f32c(q, 'q')
  n, A = q.shape
  out = torch.empty(n, dtype=torch.int64, device=q.device)
  seed, call = rng.next()
  with torch.cuda.device(q.device):
    _lib.check(_lib.lib().trl_act_epsilon_greedy(q.data_ptr(), n, A, float(eps), seed, call, out.data_ptr(),
                                                 _lib.stream_ptr(q.device)))
  return out


def act_normal(mean, log_std, rng):
  """mean [n, A], log_std [A] -> mean + exp(log_std) * N(0, 1) (ActorCriticNet's Normal(...).sample())."""
  mean = _f32c(mean, 'mean')
  n, A = mean.shape
  log_std = _f32c(log_std, 'log_std').view(-1)
  if log_std.numel() != A:
    raise ValueError('act_normal: log_std must hold %d values' % A)
  out = torch.empty_like(mean)
  seed, call = rng.next()
  with torch.cuda.device(mean.device):
    _lib.check(_lib.lib().trl_act_normal(mean.data_ptr(), log_std.data_ptr(), n, A, seed, call, out.data_ptr(),
                                         _lib.stream_ptr(mean.device)))
  return out


def act_categorical(probs, rng):
  """probs [n, A] -> int64 actions [n] ~ Categorical(probs) (A2CNet's policy head)."""
  probs = _f32c(probs, 'probs')
  n, A = probs.shape
  out = torch.empty(n, dtype=torch.int64, device=probs.device)
  seed, call = rng.next()
  with torch.cuda.device(probs.device):
    _lib.check(_lib.lib().trl_act_categorical(probs.data_ptr(), n, A, seed, call, out.data_ptr(),
                                              _lib.stream_ptr(probs.device)))
  return out


def act_ou_noise(actor_out, state_in, state_out, theta, mu, sigma, low, high, rng):
  """One Ornstein-Uhlenbeck step of the shared [A] state (state_in -> state_out) and
  clip(actor_out + state, low, high) (ddpg_controller.py:29-36, 93), one launch."""
  a = _f32c(actor_out, 'actor_out')
  n, A = a.shape
  for t, nm in ((state_in, 'state_in'), (state_out, 'state_out')):
    _lib.require_device(t, nm)
    if t.dtype != torch.float32 or t.numel() != A or not t.is_contiguous():
      raise ValueError('act_ou_noise: %s must be a contiguous float32 [%d]' % (nm, A))
  out = torch.empty_like(a)
  seed, call = rng.next()
  with torch.cuda.device(a.device):
    _lib.check(_lib.lib().trl_act_ou_noise(a.data_ptr(), n, A, state_in.data_ptr(), state_out.data_ptr(), float(theta),
                                           float(mu), float(sigma), float(low), float(high), seed, call, out.data_ptr(),
                                           _lib.stream_ptr(a.device)))
  return out


# ------------------------------------------------------------------------------ TD losses
class _TdLoss(torch.autograd.Function):
  @staticmethod
  def forward(ctx, q, q_next_target, q_next_online, action, reward, done, gamma, kind, delta):
    q = _f32c(q, 'q')
    B, A = q.shape
    qnt = _f32c(q_next_target, 'q_next_target')
    qno = _f32c(q_next_online, 'q_next_online') if q_next_online is not None else None
    action, reward, done = (_f32c(t, n).view(-1) for t, n in
                            ((action, 'action'), (reward, 'reward'), (done, 'done')))
    if qnt.shape != q.shape or (qno is not None and qno.shape != q.shape) or \
        action.numel() != B or reward.numel() != B or done.numel() != B:
      raise ValueError('td_loss: inconsistent shapes')
    loss = torch.empty((), dtype=torch.float32, device=q.device)
    need = ctx.needs_input_grad[0]
    dq = torch.empty_like(q) if need else None
    with torch.cuda.device(q.device):
      _lib.check(_lib.lib().trl_td_loss(q.data_ptr(), qnt.data_ptr(), _lib.ptr(qno), action.data_ptr(),
                                        reward.data_ptr(), done.data_ptr(), B, A, float(gamma), kind,
                                        float(delta), loss.data_ptr(), _lib.ptr(dq), _ws(q.device).data_ptr(),
                                        _lib.stream_ptr(q.device)))
    if need:
      ctx.save_for_backward(dq)
    return loss

  @staticmethod
  def backward(ctx, g):
    (dq,) = ctx.saved_tensors
    return (dq * g,) + (None,) * 8


def td_loss(q, q_next_target, action, reward, done, gamma, q_next_online=None, huber_delta=None):
  """DQN / Double-DQN loss (dqn_controller.py:64-78) from the three Q tables.

  q: Q_online(obs) [B, A] (differentiable); q_next_target: Q_target(next_obs);
  q_next_online: Q_online(next_obs) for Double-DQN, else None.  MSE like the reference unless
  huber_delta is given."""
  kind = TRL_LOSS_MSE if huber_delta is None else TRL_LOSS_HUBER
  return _TdLoss.apply(q, q_next_target, q_next_online, action, reward, done, gamma, kind,
                       0.0 if huber_delta is None else huber_delta)


class _DdpgCriticLoss(torch.autograd.Function):
  @staticmethod
  def forward(ctx, cur_q, next_q, reward, done, gamma, kind, delta):
    shape = cur_q.shape
    cur = _f32c(cur_q, 'current_q').view(-1)
    B = cur.numel()
    nxt, reward, done = (_f32c(t, n).view(-1) for t, n in
                         ((next_q, 'next_q'), (reward, 'reward'), (done, 'done')))
    if nxt.numel() != B or reward.numel() != B or done.numel() != B:
      raise ValueError('ddpg_critic_loss: inconsistent shapes')
    loss = torch.empty((), dtype=torch.float32, device=cur.device)
    need = ctx.needs_input_grad[0]
    d = torch.empty_like(cur) if need else None
    with torch.cuda.device(cur.device):
      _lib.check(_lib.lib().trl_ddpg_critic_loss(cur.data_ptr(), nxt.data_ptr(), reward.data_ptr(),
                                                 done.data_ptr(), B, float(gamma), kind, float(delta),
                                                 loss.data_ptr(), _lib.ptr(d), _ws(cur.device).data_ptr(),
                                                 _lib.stream_ptr(cur.device)))
    if need:
      ctx.save_for_backward(d.view(shape))
    return loss

  @staticmethod
  def backward(ctx, g):
    (d,) = ctx.saved_tensors
    return (d * g,) + (None,) * 6


def ddpg_critic_loss(current_q, next_q, reward, done, gamma, huber_delta=None):
  """mse_loss(current_q, reward + (1 - done) * gamma * next_q) (ddpg_controller.py:107-112)."""
  kind = TRL_LOSS_MSE if huber_delta is None else TRL_LOSS_HUBER
  return _DdpgCriticLoss.apply(current_q, next_q, reward, done, gamma, kind,
                               0.0 if huber_delta is None else huber_delta)


# ------------------------------------------------------------------------------ PPO
def normal_log_prob(mean, log_std, action):
  """Normal(mean, exp(log_std)).log_prob(action), per action dimension (ppo_controller.py:48)."""
  mean = _f32c(mean, 'mean')
  B, A = mean.shape
  log_std = _f32c(log_std, 'log_std').view(-1)
  action = _f32c(action, 'action')
  if log_std.numel() != A or action.shape != mean.shape:
    raise ValueError('normal_log_prob: inconsistent shapes')
  out = torch.empty_like(mean)
  with torch.cuda.device(mean.device):
    _lib.check(_lib.lib().trl_normal_logp(mean.data_ptr(), log_std.data_ptr(), action.data_ptr(), B, A,
                                          out.data_ptr(), _lib.stream_ptr(mean.device)))
  return out


class _PpoLoss(torch.autograd.Function):
  @staticmethod
  def forward(ctx, mean, log_std, value, action, old_logp, adv, returns, clip, alpha, beta):
    mean = _f32c(mean, 'mean')
    B, A = mean.shape
    ls_shape, v_shape = log_std.shape, value.shape
    log_std = _f32c(log_std, 'log_std').view(-1)
    value = _f32c(value, 'value').view(-1)
    action = _f32c(action, 'action')
    old_logp = _f32c(old_logp, 'old_log_probs')
    adv = _f32c(adv, 'advantages').view(-1)
    returns = _f32c(returns, 'returns').view(-1)
    if (log_std.numel() != A or value.numel() != B or action.shape != mean.shape or
        old_logp.shape != mean.shape or adv.numel() != B or returns.numel() != B):
      raise ValueError('ppo_loss: inconsistent shapes')
    losses = torch.empty(4, dtype=torch.float32, device=mean.device)
    need = any(ctx.needs_input_grad[:3])
    dmean = torch.empty_like(mean) if need else None
    dls = torch.empty_like(log_std) if need else None
    dv = torch.empty_like(value) if need else None
    with torch.cuda.device(mean.device):
      _lib.check(_lib.lib().trl_ppo_loss(mean.data_ptr(), log_std.data_ptr(), value.data_ptr(),
                                         action.data_ptr(), old_logp.data_ptr(), adv.data_ptr(),
                                         returns.data_ptr(), B, A, float(clip), float(alpha), float(beta),
                                         losses.data_ptr(), _lib.ptr(dmean), _lib.ptr(dls), _lib.ptr(dv),
                                         _ws(mean.device).data_ptr(), _lib.stream_ptr(mean.device)))
    if need:
      ctx.save_for_backward(dmean, dls.view(ls_shape), dv.view(v_shape))
    total, parts = losses[3], losses[:3]
    ctx.mark_non_differentiable(parts)
    return total, parts

  @staticmethod
  def backward(ctx, g, _gparts):
    dmean, dls, dv = ctx.saved_tensors
    return (dmean * g, dls * g, dv * g) + (None,) * 7


def ppo_loss(mean, log_std, value, action, old_log_probs, advantages, returns, clip_ratio, alpha, beta):
  """PPO clipped-surrogate loss (ppo_controller.py:55-68).

  Returns (loss, parts): `loss = actor + alpha * critic - beta * entropy` is differentiable
  w.r.t. mean, log_std and value; `parts` = [actor_loss, critic_loss, entropy_loss] detached."""
  return _PpoLoss.apply(mean, log_std, value, action, old_log_probs, advantages, returns,
                        clip_ratio, alpha, beta)


# ------------------------------------------------------------------------------ A2C
class _A2cLoss(torch.autograd.Function):
  @staticmethod
  def forward(ctx, probs, value, action, returns, alpha, beta):
    probs = _f32c(probs, 'probs')
    B, A = probs.shape
    v_shape = value.shape
    value = _f32c(value, 'value').view(-1)
    action = _f32c(action, 'action').view(-1)
    returns = _f32c(returns, 'returns').view(-1)
    if value.numel() != B or action.numel() != B or returns.numel() != B:
      raise ValueError('a2c_loss: inconsistent shapes')
    losses = torch.empty(4, dtype=torch.float32, device=probs.device)
    need = any(ctx.needs_input_grad[:2])
    dp = torch.empty_like(probs) if need else None
    dv = torch.empty_like(value) if need else None
    with torch.cuda.device(probs.device):
      _lib.check(_lib.lib().trl_a2c_loss(probs.data_ptr(), value.data_ptr(), action.data_ptr(),
                                         returns.data_ptr(), B, A, float(alpha), float(beta),
                                         losses.data_ptr(), _lib.ptr(dp), _lib.ptr(dv),
                                         _ws(probs.device).data_ptr(), _lib.stream_ptr(probs.device)))
    if need:
      ctx.save_for_backward(dp, dv.view(v_shape))
    total, parts = losses[3], losses[:3]
    ctx.mark_non_differentiable(parts)
    return total, parts

  @staticmethod
  def backward(ctx, g, _gparts):
    dp, dv = ctx.saved_tensors
    return (dp * g, dv * g) + (None,) * 4


def a2c_loss(probs, value, action, returns, alpha, beta):
  """A2C loss (a2c_controller.py:48-59) from the actor's softmax output and the critic's value."""
  return _A2cLoss.apply(probs, value, action, returns, alpha, beta)


# ------------------------------------------------------------------------------ update
def adam_step_(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
               grad_scale=1.0, clip_value=0.0):
  """In-place torch.optim.Adam default update on flat float32 buffers (+ optional gradient
  scale and elementwise clamp, torch.nn.utils.clip_grad_value_ at ppo_controller.py:72-73)."""
  for t, n in ((param, 'param'), (grad, 'grad'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
    _lib.require_device(t, n)
    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != param.numel():
      raise ValueError('adam_step_: %s must be a contiguous float32 buffer of the parameter size' % n)
  with torch.cuda.device(param.device):
    _lib.check(_lib.lib().trl_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(),
                                        exp_avg_sq.data_ptr(), param.numel(), float(lr), float(betas[0]),
                                        float(betas[1]), float(eps), int(step), float(grad_scale),
                                        float(clip_value), _lib.stream_ptr(param.device)))


def polyak_(target, source, tau):
  """target <- tau * source + (1 - tau) * target on flat buffers (ddpg_controller.py:42-57)."""
  for t, n in ((target, 'target'), (source, 'source')):
    _lib.require_device(t, n)
    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != target.numel():
      raise ValueError('polyak_: %s must be a contiguous float32 buffer' % n)
  with torch.cuda.device(target.device):
    _lib.check(_lib.lib().trl_polyak(target.data_ptr(), source.data_ptr(), target.numel(), float(tau),
                                     _lib.stream_ptr(target.device)))


# ------------------------------------------------------------------------------ MLP layers
class _Workspace:
  """Device scratch grown on demand.  A buffer that has been handed out is NEVER released: its address
  may be baked into a captured HIP graph (GraphedStep), whose later replays would otherwise write their
  partial sums into memory the caching allocator has given to someone else.  Superseded buffers are
  kept alive in `retired` (sizes grow geometrically, so the total stays under twice the largest)."""

  def __init__(self, min_bytes=1 << 20):
    self.min_bytes = min_bytes
    self.buf = None
    self.retired = []

  def get(self, device, nbytes):
    if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
      if self.buf is not None:
        self.retired.append(self.buf)
        nbytes = max(nbytes, 2 * self.buf.numel())
      self.buf = torch.empty(max(nbytes, self.min_bytes), dtype=torch.uint8, device=device)
    return self.buf


_lin_ws = {}


def _linear_ws(device, nbytes):
  """Per-device scratch of the layer-by-layer autograd Functions (weight-gradient partials).  Shared by
  every such call on the device: the kernels of one call finish with it before the next call's start
  (same stream); callers that run these Functions on several streams at once need one process each.
  The whole-step objects (FusedPpoLearn, FusedDdpgLearn, FusedDqnLearn) own their workspaces instead."""
  key = device.index if device.index is not None else torch.cuda.current_device()
  ws = _lin_ws.get(key)
  if ws is None:
    ws = _lin_ws[key] = _Workspace()
  return ws.get(device, nbytes)


class _LinearAct(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, weight, bias, act):
    x = _f32c(x, 'x')
    w = _f32c(weight, 'weight')
    b = _f32c(bias, 'bias')
    B, K = x.shape
    N = w.shape[0]
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
      _lib.check(_lib.lib().trl_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), B, K, N, act, y.data_ptr(),
                                           _lib.stream_ptr(x.device)))
    ctx.act = act
    ctx.save_for_backward(x, w, y)
    return y

  @staticmethod
  def backward(ctx, gy):
    x, w, y = ctx.saved_tensors
    gy = gy.contiguous()
    B, K = x.shape
    N = w.shape[0]
    dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
    dw = torch.empty_like(w)
    db = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = _linear_ws(x.device, _lib.lib().trl_linear_bwd_workspace_bytes(B, K, N))
    with torch.cuda.device(x.device):
      _lib.check(_lib.lib().trl_linear_bwd(gy.data_ptr(), y.data_ptr(), x.data_ptr(), w.data_ptr(), B, K, N,
                                           ctx.act, _lib.ptr(dx), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                           _lib.stream_ptr(x.device)))
    return dx, dw, db, None


def linear_act(x, weight, bias, relu=False):
  """act(x @ weight^T + bias) for 2-D float32 device tensors, fan-out <= 256: one HIP launch
  forward, one (+ a small reduce) backward (csrc/trl_mlp.hip)."""
  return _LinearAct.apply(x, weight, bias, 1 if relu else 0)


class _LinearU8(torch.autograd.Function):
  """act(float(x_u8) @ W^T + b) without materialising float(x) (csrc/trl_linear_u8.hip)."""

  @staticmethod
  def forward(ctx, x, weight, bias, act):
    w = _f32c(weight, 'weight')
    b = _f32c(bias, 'bias')
    B, K = x.shape
    N = w.shape[0]
    lib = _lib.lib()
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    ws = _linear_ws(x.device, lib.trl_linear_u8_fwd_workspace_bytes(B, N, K))
    with torch.cuda.device(x.device):
      _lib.check(lib.trl_linear_u8_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), B, K, N, act, y.data_ptr(),
                                       ws.data_ptr(), _lib.stream_ptr(x.device)))
    ctx.act = act
    ctx.save_for_backward(x, y)
    return y

  @staticmethod
  def backward(ctx, gy):
    x, y = ctx.saved_tensors
    g = torch.ops.aten.threshold_backward(gy, y, 0.0) if ctx.act == 1 else gy.contiguous()
    B, K = x.shape
    N = g.shape[1]
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
      _lib.check(_lib.lib().trl_linear_u8_wgrad(g.data_ptr(), x.data_ptr(), B, K, N, dw.data_ptr(),
                                                _lib.stream_ptr(x.device)))
    return None, dw, g.sum(0), None


def linear_u8_supported(x, weight):
  """uint8 device batch [B, K] with B % 32 == 0, K % 32 == 0 and a fan-out that is a multiple of 128."""
  return (x.is_cuda and x.dtype == torch.uint8 and x.dim() == 2 and x.is_contiguous() and x.shape[0] % 32 == 0
          and x.shape[0] > 0 and x.shape[1] % 32 == 0 and weight.shape[0] % 128 == 0
          and weight.dtype == torch.float32 and weight.is_contiguous())


def linear_u8(x, weight, bias, relu=False):
  """act(x.float() @ weight^T + bias) for a uint8 device batch, read as uint8: the first layer of a
  Q-network on frames the replay ring keeps uint8 (the cast is exact: same values as the reference's
  float32 frames)."""
  return _LinearU8.apply(x, weight, bias, 1 if relu else 0)


def _fill_desc(d, weights, biases):
  d.depth = len(weights)
  d.dims[0] = weights[0].shape[1]
  for l, (w, b) in enumerate(zip(weights, biases)):
    d.dims[l + 1] = w.shape[0]
    d.w[l] = w.data_ptr()
    d.b[l] = b.data_ptr()


class _Mlp64Group(torch.autograd.Function):
  """G independent <=64-wide MLPs over batches of the same size: one forward launch, one
  backward launch + one reduce for the whole group (csrc/trl_mlp.hip)."""

  @staticmethod
  def forward(ctx, meta, *tensors):
    import ctypes as C
    direct, depths = meta
    G = len(depths)
    xs = [_f32c(t, 'x') for t in tensors[:G]]
    B = xs[0].shape[0]
    nets, pos = [], G
    for D in depths:
      prm = tensors[pos:pos + 2 * D]
      nets.append(([_f32c(p, 'weight') for p in prm[0::2]], [_f32c(p, 'bias') for p in prm[1::2]], prm))
      pos += 2 * D
    descs = (_lib.MlpDesc * G)()
    acts, actp = [], (C.c_void_p * (3 * G))()
    for g, (ws_, bs_, _) in enumerate(nets):
      if xs[g].shape[0] != B:
        raise ValueError('mlp64_group: every network must see the same batch size')
      _fill_desc(descs[g], ws_, bs_)
      a = [torch.empty((B, w.shape[0]), dtype=torch.float32, device=xs[g].device) for w in ws_]
      acts.append(a)
      for l, t in enumerate(a):
        actp[3 * g + l] = t.data_ptr()
    xp = (C.c_void_p * G)(*[x.data_ptr() for x in xs])
    dev = xs[0].device
    with torch.cuda.device(dev):
      _lib.check(_lib.lib().trl_mlp_fwd_group(descs, G, xp, B, actp, _lib.stream_ptr(dev)))
    ctx.depths = depths
    # direct: the reduce launch writes every parameter's gradient straight into its .grad
    # (OVERWRITING it) and autograd is handed None -- no AccumulateGrad add per parameter
    ctx.grad_dst = [[p.grad for p in prm] for _, _, prm in nets] if direct else None
    saved = list(xs)
    for (ws_, bs_, _), a in zip(nets, acts):
      saved += list(ws_) + list(bs_) + a[:-1]
    ctx.save_for_backward(*saved)
    return tuple(a[-1] for a in acts)

  @staticmethod
  def backward(ctx, *dys):
    import ctypes as C
    depths = ctx.depths
    G = len(depths)
    saved = ctx.saved_tensors
    xs, pos = saved[:G], G
    B, dev = xs[0].shape[0], xs[0].device
    descs = (_lib.MlpDesc * G)()
    actp = (C.c_void_p * (3 * G))()
    nets, nbytes = [], 0
    for g, D in enumerate(depths):
      ws_, bs_, a = saved[pos:pos + D], saved[pos + D:pos + 2 * D], saved[pos + 2 * D:pos + 3 * D - 1]
      pos += 3 * D - 1
      _fill_desc(descs[g], ws_, bs_)
      for l, t in enumerate(a):
        actp[3 * g + l] = t.data_ptr()
      nets.append((ws_, bs_))
      nbytes += _lib.lib().trl_mlp_bwd_workspace_bytes(C.byref(descs[g]), B)
    dyl = []
    for g, dy in enumerate(dys):
      if dy is None:      # an unused head: zero gradient
        dy = torch.zeros((B, nets[g][0][-1].shape[0]), dtype=torch.float32, device=dev)
      dyl.append(dy.contiguous())
    xp = (C.c_void_p * G)(*[x.data_ptr() for x in xs])
    dyp = (C.c_void_p * G)(*[d.data_ptr() for d in dyl])
    ws = _linear_ws(dev, nbytes)
    grads, gradp, dstp = None, None, None
    if ctx.grad_dst is not None:
      dstp = (C.c_void_p * (6 * G))()
      for g, dst in enumerate(ctx.grad_dst):
        for j, t in enumerate(dst):
          dstp[6 * g + j] = t.data_ptr()
    else:
      grads = [torch.empty(sum(w.numel() + b.numel() for w, b in zip(ws_, bs_)), dtype=torch.float32, device=dev)
               for ws_, bs_ in nets]
      gradp = (C.c_void_p * G)(*[t.data_ptr() for t in grads])
    with torch.cuda.device(dev):
      _lib.check(_lib.lib().trl_mlp_bwd_group(descs, G, xp, actp, dyp, B, gradp, dstp, ws.data_ptr(),
                                              _lib.stream_ptr(dev)))
    out = [None] + [None] * G
    for g, (ws_, bs_) in enumerate(nets):
      if grads is None:
        out += [None] * (2 * len(ws_))
        continue
      off = 0
      for w, b in zip(ws_, bs_):
        out.append(grads[g][off:off + w.numel()].view(w.shape))
        off += w.numel()
        out.append(grads[g][off:off + b.numel()])
        off += b.numel()
    return tuple(out)


def mlp64_group(xs, nets, direct_grads=False):
  """[(weights, biases), ...] applied to xs[g] (same batch size): Linear -> ReLU -> ... -> Linear,
  1..3 Linear layers each, every width <= 64, up to 4 networks, as one forward launch and one
  backward launch (+ reduce).  The inputs get no gradient.  Returns the list of outputs.

  direct_grads=True: the backward OVERWRITES each parameter's existing contiguous float32 `.grad`
  in place instead of returning gradients for autograd to accumulate.  Only valid when each
  stack is used once per backward pass and its gradients were meant to start from zero (what
  the controllers' learn() does)."""
  tensors, depths = list(xs), []
  for weights, biases in nets:
    depths.append(len(weights))
    for w, b in zip(weights, biases):
      tensors += [w, b]
  if direct_grads:
    for p in tensors[len(xs):]:
      if p.grad is None or not p.grad.is_contiguous() or p.grad.dtype != torch.float32 or p.grad.shape != p.shape:
        raise ValueError('direct_grads needs an existing contiguous float32 .grad on every parameter')
  return list(_Mlp64Group.apply((bool(direct_grads), tuple(depths)), *tensors))


def mlp64(x, weights, biases, direct_grads=False):
  """One network of mlp64_group."""
  return mlp64_group([x], [(weights, biases)], direct_grads)[0]


# ------------------------------------------------------------------------------ fused PPO learner step
class FusedPpoLearn:
  """forward + PPO loss + backward of an ActorCriticNet whose stacks are at most 64 wide (one hidden
  layer over <= 9 observations: up to 256 wide), as one
  launch (+ a small reduce): csrc/trl_learn.hip, C ABI trl_ppo_learn_fused.

  Built once per controller from the stacks' Linear modules (whose parameters and .grad views live
  in flat buffers, optim.FlatParams): the ctypes argument block is prepared here, so a call costs
  five data_ptr() reads and one foreign call.  Every parameter's .grad is OVERWRITTEN with
  d loss / d parameter (no zero_grad needed); returns the [4] loss tensor
  (actor, critic, entropy, total)."""

  # the reference's own minibatch sizes (32 default, 64 in its spec: examples/ppo.py:12, 92) take the whole-step
  # kernel too (one or two 32-sample strips, two launches per learn()); below that, e.g. the last partial
  # minibatch of an epoch, the layer-by-layer path
  MIN_BATCH = 32

  @staticmethod
  def supports(critic_lin, actor_lin, log_std):
    """log_std: ActorCriticNet's parameter (PPO), or None for A2CNet's softmax actor (A2C)."""
    if len(critic_lin) != len(actor_lin) or len(critic_lin) not in (2, 3):
      return False
    A = actor_lin[-1].out_features
    if critic_lin[-1].out_features != 1 or A > (2 if len(critic_lin) == 3 else 4):
      return False
    if log_std is not None and log_std.numel() != A:
      return False
    if critic_lin[0].in_features != actor_lin[0].in_features:
      return False
    # one hidden layer over a fan-in of at most 9 and at most two actions may be up to 256 wide (the
    # reference's own shape, ppo_controller.py:14 / a2c_controller.py: k_ac_learn_wide)
    wide = 64
    if len(critic_lin) == 2 and critic_lin[0].in_features <= 9 and A <= 2:
      wide = 128 if (A == 2 and critic_lin[0].in_features > 7) else 256     # (register budget of the widest variant)
    for lin in (critic_lin, actor_lin):
      for i, m in enumerate(lin):
        if m.bias is None or m.in_features > wide or (i + 1 < len(lin) and m.out_features > wide):
          return False
        if m.weight.dtype != torch.float32 or not m.weight.is_cuda:
          return False
    return True

  # batches the role-split kernel (and with it the bf16x6 products) takes: 8 x 32 x 128 rows and more
  BX6_MIN_BATCH = 8 * 32 * 128

  def __init__(self, critic_lin, actor_lin, log_std, products='f32'):
    import ctypes as C
    if products not in ('f32', 'bf16x6'):
      raise ValueError("products must be 'f32' or 'bf16x6'")
    self._C = C
    self.critic_lin, self.actor_lin, self.log_std = list(critic_lin), list(actor_lin), log_std
    self.A = actor_lin[-1].out_features
    self.K0 = critic_lin[0].in_features
    self.device = critic_lin[0].weight.device
    self.products = products
    self._key = None
    self._ws = _Workspace()      # this object's own scratch (its address may be captured into a HIP graph)
    self._bind()

  def _ptr_key(self):
    ls = self.log_std
    return (self.critic_lin[0].weight.data_ptr(), self.actor_lin[-1].bias.data_ptr(),
            ls.data_ptr() if ls is not None else 0,
            self.critic_lin[0].weight.grad.data_ptr() if self.critic_lin[0].weight.grad is not None else 0,
            ls.grad.data_ptr() if ls is not None and ls.grad is not None else 0)

  def _bind(self):
    C = self._C
    self.descs = (_lib.MlpDesc * 2)()
    self.gradp = ((C.c_void_p * 6)(), (C.c_void_p * 6)())
    for g, lin in enumerate((self.critic_lin, self.actor_lin)):
      ws_ = [m.weight for m in lin]
      bs_ = [m.bias for m in lin]
      for t in ws_ + bs_ + ([self.log_std] if self.log_std is not None else []):
        if t.grad is None or not t.grad.is_contiguous() or not t.is_contiguous():
          raise ValueError('FusedPpoLearn needs contiguous parameters with an existing contiguous .grad')
      _fill_desc(self.descs[g], ws_, bs_)
      for l, (w, b) in enumerate(zip(ws_, bs_)):
        self.gradp[g][2 * l] = w.grad.data_ptr()
        self.gradp[g][2 * l + 1] = b.grad.data_ptr()
    self._key = self._ptr_key()

  def usable(self, obs):
    return obs.is_cuda and obs.dim() == 2 and obs.dtype == torch.float32 and obs.shape[0] >= self.MIN_BATCH \
        and obs.shape[1] == self.K0

  def bx6_takes(self, B):
    """True when a step over B rows runs its 64 x 64 products on the bf16 MFMA (products='bf16x6' and the shape is the
    role-split kernel's: two hidden layers <= 64 wide over <= 4 observations, <= 2 actions, a large batch); every
    other step of this object is the exact-f32 kernels'."""
    return self.products == 'bf16x6' and len(self.critic_lin) == 3 and self.K0 <= 4 and self.A <= 2 \
        and B >= self.BX6_MIN_BATCH

  def forward(self, obs, action=None, want_value=True, want_mean=True):
    """No-grad forward over obs [B, fan-in] writing only the outputs (trl_ppo_forward_fused):
    returns (value [B] | None, mean [B, A] | None, logp [B, A] | None); logp =
    Normal(mean, exp(log_std)).log_prob(action) per dimension when `action` is given."""
    C = self._C
    obs = _f32c(obs, 'obs')
    B, dev = obs.shape[0], obs.device
    if action is not None:
      action = _f32c(action, 'action')
      if action.numel() != B * self.A:
        raise ValueError('ppo_forward_fused: action must be [B, %d]' % self.A)
    if self._ptr_key() != self._key:
      self._bind()
    value = torch.empty(B, dtype=torch.float32, device=dev) if want_value else None
    mean = torch.empty((B, self.A), dtype=torch.float32, device=dev) if want_mean else None
    logp = torch.empty((B, self.A), dtype=torch.float32, device=dev) if action is not None else None
    lib = _lib.lib()
    fn = lib.trl_ppo_forward_fused_bx6 if self.bx6_takes(B) else lib.trl_ppo_forward_fused
    with torch.cuda.device(dev):
      _lib.check(fn(
          C.byref(self.descs[0]), C.byref(self.descs[1]), _lib.ptr(self.log_std), obs.data_ptr(), _lib.ptr(action),
          B, self.A, _lib.ptr(value), _lib.ptr(mean), _lib.ptr(logp), _lib.stream_ptr(dev)))
    return value, mean, logp

  def __call__(self, obs, action, returns, old_logp, adv, clip, alpha, beta, adam=None):
    """adam: an optim.FlatAdam over the flat buffers holding these parameters -> clip + Adam run
    inside the reduce launch (its step count is advanced here); None -> gradients only."""
    C = self._C
    B = obs.shape[0]
    obs = _f32c(obs, 'obs')
    action = _f32c(action, 'action')
    returns = _f32c(returns, 'returns')
    old_logp = _f32c(old_logp, 'old_log_probs')
    adv = _f32c(adv, 'advantages')
    if action.numel() != B * self.A or old_logp.numel() != B * self.A or returns.numel() != B or adv.numel() != B:
      raise ValueError('ppo_learn_fused: inconsistent shapes')
    if self._ptr_key() != self._key:     # parameters or gradients were re-homed
      self._bind()
    dev = obs.device
    lib = _lib.lib()
    ws = self._ws.get(dev, lib.trl_ppo_learn_workspace_bytes(C.byref(self.descs[0]), C.byref(self.descs[1]), B))
    losses = torch.empty(4, dtype=torch.float32, device=dev)
    aref = None
    if adam is not None:
      aref = C.byref(adam.fused_args())
    fn = lib.trl_ppo_learn_fused_bx6 if self.bx6_takes(B) else lib.trl_ppo_learn_fused
    with torch.cuda.device(dev):
      _lib.check(fn(
          C.byref(self.descs[0]), C.byref(self.descs[1]), self.log_std.data_ptr(), obs.data_ptr(),
          action.data_ptr(), returns.data_ptr(), old_logp.data_ptr(), adv.data_ptr(), B, self.A, float(clip),
          float(alpha), float(beta), losses.data_ptr(), self.gradp[0], self.gradp[1],
          self.log_std.grad.data_ptr(), aref, ws.data_ptr(), _lib.stream_ptr(dev)))
    return losses

  # ---- the minibatch loop in one launch (trl_ppo_learn_epoch, csrc/trl_epoch.hip)
  EPOCH_MAX_MINIBATCH = 4096

  @staticmethod
  def epoch_supports(critic_lin, actor_lin, log_std):
    """Shapes the persistent minibatch-loop kernel takes (wider than the whole-step kernel's on some axes: up to four
    actions at any width): one hidden layer (a multiple of 16, <= 256) or two (multiples of 16, <= 64) over at most 15
    observations, PPO (a log_std), float32 parameters on a HIP device."""
    if log_std is None or len(critic_lin) != len(actor_lin) or len(critic_lin) not in (2, 3):
      return False
    A = actor_lin[-1].out_features
    if critic_lin[-1].out_features != 1 or not 1 <= A <= 4 or log_std.numel() != A:
      return False
    K0 = critic_lin[0].in_features
    if K0 != actor_lin[0].in_features or K0 > 15:
      return False
    for lin in (critic_lin, actor_lin):
      hid = [m.out_features for m in lin[:-1]]
      if any(h % 16 or h < 16 for h in hid):
        return False
      if (len(hid) == 1 and hid[0] > 256) or (len(hid) == 2 and max(hid) > 64):
        return False
      for i, m in enumerate(lin):
        if m.bias is None or m.weight.dtype != torch.float32 or not m.weight.is_cuda:
          return False
        if i > 0 and m.in_features != lin[i - 1].out_features:
          return False
    return True

  def epoch_supported(self):
    if not FusedPpoLearn.epoch_supports(self.critic_lin, self.actor_lin, self.log_std):
      return False
    C = self._C      # the library has the last word (LDS budget of the two workgroups)
    return bool(_lib.lib().trl_ppo_learn_epoch_supported(C.byref(self.descs[0]), C.byref(self.descs[1]), self.A))

  def learn_epoch(self, obs, action, returns, old_logp, adv, minibatch, n_epochs, clip, alpha, beta, adam):
    """K = n_epochs * ceil(n / minibatch) consecutive learn() steps over the consecutive row ranges of the columns
    (examples/ppo.py:66-73), in one launch.  `adam` (optim.FlatAdam over the flat buffers of these parameters) has its
    step count advanced by K.  Returns the [K, 4] loss tensor (actor, critic, entropy, total per step)."""
    C = self._C
    n = obs.shape[0]
    obs = _f32c(obs, 'obs')
    action = _f32c(action, 'action')
    returns = _f32c(returns, 'returns')
    old_logp = _f32c(old_logp, 'old_log_probs')
    adv = _f32c(adv, 'advantages')
    if obs.dim() != 2 or obs.shape[1] != self.K0 or action.numel() != n * self.A or old_logp.numel() != n * self.A \
        or returns.numel() != n or adv.numel() != n:
      raise ValueError('ppo_learn_epoch: inconsistent shapes')
    if self._ptr_key() != self._key:
      self._bind()
    dev = obs.device
    lib = _lib.lib()
    minibatch, n_epochs = int(minibatch), int(n_epochs)
    K = -(-n // minibatch) * n_epochs
    ws = self._ws.get(dev, lib.trl_ppo_learn_epoch_workspace_bytes(n, minibatch, n_epochs))
    losses = torch.empty((K, 4), dtype=torch.float32, device=dev)
    a = adam.fused_args()                 # step = count of the first update
    adam.step_count += K - 1
    with torch.cuda.device(dev):
      _lib.check(lib.trl_ppo_learn_epoch(
          C.byref(self.descs[0]), C.byref(self.descs[1]), self.log_std.data_ptr(), obs.data_ptr(), action.data_ptr(),
          returns.data_ptr(), old_logp.data_ptr(), adv.data_ptr(), n, minibatch, n_epochs, self.A, float(clip),
          float(alpha), float(beta), losses.data_ptr(), C.byref(a), ws.data_ptr(), _lib.stream_ptr(dev)))
    return losses

  def learn_tiles(self, obs, action, returns, old_logp, adv, clip, alpha, beta, adam, update=True):
    """One learn() step over one batch of any size through the tile pipeline spread over the chip (trl_ppo_learn_tiles):
    the whole-step path for shapes the register-chained kernels do not take.  `adam`: the optim.FlatAdam of these
    parameters (gradients land in its flat .grad; update=True also applies clip + Adam and advances its step count).
    Returns the [4] loss tensor."""
    C = self._C
    B = obs.shape[0]
    obs = _f32c(obs, 'obs')
    action = _f32c(action, 'action')
    returns = _f32c(returns, 'returns')
    old_logp = _f32c(old_logp, 'old_log_probs')
    adv = _f32c(adv, 'advantages')
    if obs.dim() != 2 or obs.shape[1] != self.K0 or action.numel() != B * self.A or old_logp.numel() != B * self.A \
        or returns.numel() != B or adv.numel() != B:
      raise ValueError('ppo_learn_tiles: inconsistent shapes')
    if self._ptr_key() != self._key:
      self._bind()
    dev = obs.device
    lib = _lib.lib()
    a = adam.fused_args(update)
    ws = self._ws.get(dev, lib.trl_ppo_learn_tiles_workspace_bytes(a.n, B))
    losses = torch.empty(4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
      _lib.check(lib.trl_ppo_learn_tiles(
          C.byref(self.descs[0]), C.byref(self.descs[1]), self.log_std.data_ptr(), obs.data_ptr(), action.data_ptr(),
          returns.data_ptr(), old_logp.data_ptr(), adv.data_ptr(), B, self.A, float(clip), float(alpha), float(beta),
          losses.data_ptr(), C.byref(a), 1 if update else 0, ws.data_ptr(), _lib.stream_ptr(dev)))
    return losses

  @staticmethod
  def tiles_supports(critic_lin, actor_lin, a2c=False):
    """Shapes the tile pipeline takes for one step over a batch of any size (trl_ppo_learn_tiles / trl_a2c_learn_tiles):
    one hidden layer (a multiple of 16, <= 256) over at most 31 observations with up to six actions, or two hidden
    layers (<= 64) over at most 15 observations with up to four (A2C: at least two actions), float32 parameters on a
    HIP device."""
    if len(critic_lin) != len(actor_lin) or len(critic_lin) not in (2, 3):
      return False
    one = len(critic_lin) == 2
    A = actor_lin[-1].out_features
    if critic_lin[-1].out_features != 1 or not (2 if a2c else 1) <= A <= (6 if one else 4):
      return False
    K0 = critic_lin[0].in_features
    if K0 != actor_lin[0].in_features or K0 > (31 if one else 15):
      return False
    for lin in (critic_lin, actor_lin):
      hid = [m.out_features for m in lin[:-1]]
      if any(h % 16 or h < 16 for h in hid) or (len(hid) == 1 and hid[0] > 256) or (len(hid) == 2 and max(hid) > 64):
        return False
      if any(m.bias is None or m.weight.dtype != torch.float32 or not m.weight.is_cuda for m in lin):
        return False
    return True

  def forward_tiles(self, obs, action=None, want_mean=False, critic_only=False):
    """No-grad forward through the tile pipeline (trl_ac_forward_tiles) for the shapes tiles_supports() takes:
    (value [B], mean [B, A] | None, logp [B, A] | None); logp = Normal(mean, exp(log_std)).log_prob(action) per
    dimension when `action` is given (PPO).  critic_only (or an A2C object: no log_std): the values alone."""
    C = self._C
    obs = _f32c(obs, 'obs')
    B, dev = obs.shape[0], obs.device
    if obs.dim() != 2 or obs.shape[1] != self.K0:
      raise ValueError('ac_forward_tiles: obs must be [B, %d]' % self.K0)
    if self._ptr_key() != self._key:
      self._bind()
    critic_only = critic_only or self.log_std is None or (action is None and not want_mean)
    value = torch.empty(B, dtype=torch.float32, device=dev)
    mean = logp = None
    if not critic_only:
      if action is not None:
        action = _f32c(action, 'action')
        if action.numel() != B * self.A:
          raise ValueError('ac_forward_tiles: action must be [B, %d]' % self.A)
        logp = torch.empty((B, self.A), dtype=torch.float32, device=dev)
      if want_mean or logp is None:
        mean = torch.empty((B, self.A), dtype=torch.float32, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    with torch.cuda.device(dev):
      _lib.check(_lib.lib().trl_ac_forward_tiles(
          C.byref(self.descs[0]), None if critic_only else C.byref(self.descs[1]),
          None if critic_only else self.log_std.data_ptr(), obs.data_ptr(), ptr(action) if not critic_only else None, B,
          self.A, value.data_ptr(), ptr(mean), ptr(logp), _lib.stream_ptr(dev)))
    return value, mean, logp

  def a2c_tiles(self, obs, action, returns, alpha, beta, adam, update=True):
    """A2CController.learn's device work through the tile pipeline (trl_a2c_learn_tiles) for the A2CNet shapes the
    register-chained kernels do not take (three to six actions, a wide layer over 10 - 31 observations): the detached
    advantage returns - V comes from the forward-only kernel where it takes the shape, else from the critic's own
    workgroups of the step (launched ahead of the actor's).  Returns the [4] loss tensor."""
    C = self._C
    B = obs.shape[0]
    obs = _f32c(obs, 'obs')
    action = _f32c(action, 'action')
    returns = _f32c(returns, 'returns')
    if obs.dim() != 2 or obs.shape[1] != self.K0 or action.numel() != B or returns.numel() != B:
      raise ValueError('a2c_learn_tiles: inconsistent shapes')
    if self._ptr_key() != self._key:
      self._bind()
    adv = None      # (the library's own: the critic's workgroups run first and leave returns - V for the actor's)
    if self.K0 <= 9 or (len(self.critic_lin) == 3 and self.K0 <= 12):
      with torch.no_grad():      # the forward-only kernel takes the shape: both roles in one launch behind it
        adv = (returns.view(-1) - self.forward(obs, want_mean=False)[0]).contiguous()
    dev = obs.device
    lib = _lib.lib()
    a = adam.fused_args(update)
    ws = self._ws.get(dev, lib.trl_ppo_learn_tiles_workspace_bytes(a.n, B))
    losses = torch.empty(4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
      _lib.check(lib.trl_a2c_learn_tiles(
          C.byref(self.descs[0]), C.byref(self.descs[1]), obs.data_ptr(), action.data_ptr(), returns.data_ptr(),
          adv.data_ptr() if adv is not None else None, B, self.A, float(alpha), float(beta), losses.data_ptr(), C.byref(a),
          1 if update else 0,
          ws.data_ptr(), _lib.stream_ptr(dev)))
    return losses

  def a2c(self, obs, action, returns, alpha, beta, adam=None):
    """A2CController.learn's device work for A2CNet stacks (trl_a2c_learn_fused): a forward-only
    pass gives the detached advantage returns - V, then one launch for forward + loss + backward
    and one for the reduction (+ Adam).  Returns the [4] loss tensor."""
    C = self._C
    B = obs.shape[0]
    obs = _f32c(obs, 'obs')
    action = _f32c(action, 'action')
    returns = _f32c(returns, 'returns')
    if action.numel() != B or returns.numel() != B:
      raise ValueError('a2c_learn_fused: inconsistent shapes')
    value, _, _ = self.forward(obs, want_mean=False)
    adv = returns.view(-1) - value
    dev = obs.device
    lib = _lib.lib()
    ws = self._ws.get(dev, lib.trl_ppo_learn_workspace_bytes(C.byref(self.descs[0]), C.byref(self.descs[1]), B))
    losses = torch.empty(4, dtype=torch.float32, device=dev)
    aref = C.byref(adam.fused_args()) if adam is not None else None
    fn = lib.trl_a2c_learn_fused_bx6 if self.bx6_takes(B) else lib.trl_a2c_learn_fused
    with torch.cuda.device(dev):
      _lib.check(fn(
          C.byref(self.descs[0]), C.byref(self.descs[1]), obs.data_ptr(), action.data_ptr(), returns.data_ptr(),
          adv.data_ptr(), B, self.A, float(alpha), float(beta), losses.data_ptr(), self.gradp[0], self.gradp[1],
          aref, ws.data_ptr(), _lib.stream_ptr(dev)))
    return losses


# ------------------------------------------------------------------------------ fused off-policy learner steps
def _three_layer(lin, max_width=1024):
  """Linear modules of a two-hidden-layer ReLU stack the phase kernels (csrc/trl_offpolicy.hip) take."""
  if len(lin) != 3:
    return False
  for i, m in enumerate(lin):
    if m.bias is None or m.weight.dtype != torch.float32 or not m.weight.is_cuda:
      return False
    if m.in_features > max_width or m.out_features > max_width:
      return False
    if i < 2 and (m.out_features % 16 or m.out_features < 16):
      return False
  return True


def _flat_ok(flat, lin):
  """Every parameter of `lin` is a view into flat.data (optim.FlatParams), 16 B-aligned weights."""
  lo, hi = flat.data.data_ptr(), flat.data.data_ptr() + 4 * flat.data.numel()
  for m in lin:
    for t in (m.weight, m.bias):
      if not (t.is_contiguous() and lo <= t.data_ptr() and t.data_ptr() + 4 * t.numel() <= hi):
        return False
    if m.weight.data_ptr() % 16:
      return False
  return True


class _FusedOffPolicy:
  MAX_BATCH = 4096

  def _desc(self, descs, i, lin):
    _fill_desc(descs[i], [m.weight for m in lin], [m.bias for m in lin])

  def usable(self, obs):
    B = obs.shape[0]
    return obs.is_cuda and obs.dim() == 2 and obs.dtype == torch.float32 and B >= 16 and B % 16 == 0 \
        and B <= self.MAX_BATCH and obs.shape[1] == self.S


class FusedDdpgLearn(_FusedOffPolicy):
  """DDPGController.learn's device work as 10 phases -- one launch (k_chain), or one per phase -- (C ABI trl_ddpg_learn_fused,
  csrc/trl_offpolicy.hip) for two-hidden-layer actor / critic stacks whose parameters live in
  optim.FlatParams buffers.  Built once per controller; a call is one foreign call."""

  @staticmethod
  def supports(actor_lin, critic_lin, target_actor_lin, target_critic_lin):
    if not all(_three_layer(l) for l in (actor_lin, critic_lin, target_actor_lin, target_critic_lin)):
      return False
    S, A = actor_lin[0].in_features, actor_lin[2].out_features
    return critic_lin[0].in_features == S + A and critic_lin[2].out_features == 1

  def __init__(self, actor_lin, critic_lin, target_actor_lin, target_critic_lin, actor_flat, critic_flat,
               target_actor_flat, target_critic_flat):
    import ctypes as C
    self._C = C
    self.lins = [list(l) for l in (actor_lin, critic_lin, target_actor_lin, target_critic_lin)]
    self.flats = (actor_flat, critic_flat, target_actor_flat, target_critic_flat)
    for lin, flat in zip(self.lins, self.flats):
      if not _flat_ok(flat, lin):
        raise ValueError('FusedDdpgLearn needs parameters that are views into their FlatParams buffer')
    self.S, self.A = actor_lin[0].in_features, actor_lin[2].out_features
    self.device = actor_lin[0].weight.device
    self.descs = (_lib.MlpDesc * 4)()
    for i, lin in enumerate(self.lins):
      self._desc(self.descs, i, lin)
    self._key = self._ptr_key()
    self._ws = _Workspace()

  def _ptr_key(self):
    return tuple(l[0].weight.data_ptr() for l in self.lins) + tuple(f.data.data_ptr() for f in self.flats)

  def __call__(self, obs, action, reward, next_obs, done, gamma, tau, actor_optim, critic_optim, update=True,
               huber_delta=None):
    """update=True: both Adam steps and both Polyak averages happen inside the launches (the
    optimizers' step counts advance here); False: the gradients land in the flat .grad buffers
    only.  Returns the [2] loss tensor (actor, critic)."""
    C = self._C
    B = obs.shape[0]
    obs, action, reward = _f32c(obs, 'obs'), _f32c(action, 'action'), _f32c(reward, 'reward')
    next_obs, done = _f32c(next_obs, 'next_obs'), _f32c(done, 'done')
    if action.numel() != B * self.A or reward.numel() != B or done.numel() != B or next_obs.shape != obs.shape:
      raise ValueError('ddpg_learn_fused: inconsistent shapes')
    if self._ptr_key() != self._key:
      raise _lib.TrlError('FusedDdpgLearn: parameters were re-homed after construction')
    dev = obs.device
    lib = _lib.lib()
    d = self.descs
    ws = self._ws.get(dev, lib.trl_ddpg_learn_workspace_bytes(C.byref(d[0]), C.byref(d[1]), B))
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    a_args, c_args = actor_optim.fused_args(update, True), critic_optim.fused_args(update, True)
    kind, delta = (0, 0.0) if huber_delta is None else (1, float(huber_delta))
    with torch.cuda.device(dev):
      _lib.check(lib.trl_ddpg_learn_fused(
          C.byref(d[0]), C.byref(d[1]), C.byref(d[2]), C.byref(d[3]), obs.data_ptr(), action.data_ptr(),
          reward.data_ptr(), next_obs.data_ptr(), done.data_ptr(), B, float(gamma), kind, delta, losses.data_ptr(),
          C.byref(a_args), C.byref(c_args), self.flats[2].data.data_ptr(), self.flats[3].data.data_ptr(), float(tau),
          ws.data_ptr(), _lib.stream_ptr(dev)))
    return losses


class FusedDqnLearn(_FusedOffPolicy):
  """DQNController.learn's device work up to the Adam step as 6 launches (C ABI trl_dqn_learn_fused)
  for two-hidden-layer Q-networks in optim.FlatParams buffers; the periodic hard sync stays with
  the caller."""

  @staticmethod
  def supports(q_lin, target_lin):
    return all(_three_layer(l) or FusedDqnLearn._u8_shape(l) for l in (q_lin, target_lin))

  @staticmethod
  def _u8_shape(lin):
    """Stacks whose first layer the uint8 kernels take: any fan-in that is a multiple of 32, fan-out a
    multiple of 128 (trl_dqn_learn_fused_u8); the rest as _three_layer."""
    if len(lin) != 3 or lin[0].in_features % 32 or lin[0].in_features > (1 << 20) or lin[0].out_features % 128:
      return False
    class _Narrow:     # the first layer with a fan-in _three_layer accepts; everything else as it is
      pass
    m0 = _Narrow()
    m0.bias, m0.weight, m0.in_features, m0.out_features = lin[0].bias, lin[0].weight, 32, lin[0].out_features
    return _three_layer([m0, lin[1], lin[2]])

  def usable(self, obs):
    B = obs.shape[0]
    if not (obs.is_cuda and obs.dim() == 2 and obs.shape[1] == self.S and obs.is_contiguous()):
      return False
    if obs.dtype == torch.uint8:
      return self.u8_ok and B >= 32 and B % 32 == 0 and B <= self.MAX_BATCH
    return self.f32_ok and obs.dtype == torch.float32 and B >= 16 and B % 16 == 0 and B <= self.MAX_BATCH

  def __init__(self, q_lin, target_lin, q_flat, target_flat):
    import ctypes as C
    self._C = C
    self.lins = [list(q_lin), list(target_lin)]
    self.flats = (q_flat, target_flat)
    for lin, flat in zip(self.lins, self.flats):
      if not _flat_ok(flat, lin):
        raise ValueError('FusedDqnLearn needs parameters that are views into their FlatParams buffer')
    self.S, self.A = q_lin[0].in_features, q_lin[2].out_features
    self.f32_ok = all(_three_layer(l) for l in self.lins)
    self.u8_ok = all(self._u8_shape(l) for l in self.lins)
    self.device = q_lin[0].weight.device
    self.descs = (_lib.MlpDesc * 2)()
    for i, lin in enumerate(self.lins):
      self._desc(self.descs, i, lin)
    self._key = self._ptr_key()
    self._ws = _Workspace()

  def _ptr_key(self):
    return tuple(l[0].weight.data_ptr() for l in self.lins) + tuple(f.data.data_ptr() for f in self.flats)

  def __call__(self, obs, action, reward, next_obs, done, gamma, optim, double_dqn=False, update=True,
               huber_delta=None):
    """Returns the [1] td-loss tensor."""
    C = self._C
    B = obs.shape[0]
    u8 = obs.dtype == torch.uint8
    if u8:
      if next_obs.dtype != torch.uint8 or not next_obs.is_cuda or not next_obs.is_contiguous():
        raise ValueError('dqn_learn_fused: obs and next_obs must both be contiguous uint8 device tensors')
    else:
      obs, next_obs = _f32c(obs, 'obs'), _f32c(next_obs, 'next_obs')
    action, reward, done = _f32c(action, 'action'), _f32c(reward, 'reward'), _f32c(done, 'done')
    if action.numel() != B or reward.numel() != B or done.numel() != B or next_obs.shape != obs.shape:
      raise ValueError('dqn_learn_fused: inconsistent shapes')
    if self._ptr_key() != self._key:
      raise _lib.TrlError('FusedDqnLearn: parameters were re-homed after construction')
    dev = obs.device
    lib = _lib.lib()
    d = self.descs
    ws = self._ws.get(dev, (lib.trl_dqn_learn_u8_workspace_bytes if u8 else lib.trl_dqn_learn_workspace_bytes)(
        C.byref(d[0]), B))
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    args = optim.fused_args(update, True)
    kind, delta = (0, 0.0) if huber_delta is None else (1, float(huber_delta))
    with torch.cuda.device(dev):
      _lib.check((lib.trl_dqn_learn_fused_u8 if u8 else lib.trl_dqn_learn_fused)(
          C.byref(d[0]), C.byref(d[1]), obs.data_ptr(), action.data_ptr(), reward.data_ptr(), next_obs.data_ptr(),
          done.data_ptr(), B, float(gamma), 1 if double_dqn else 0, kind, delta, loss.data_ptr(), C.byref(args),
          ws.data_ptr(), _lib.stream_ptr(dev)))
    return loss

  def ring_usable(self, view):
    """A ring of Transitions with uint8 frames of this network's fan-in (the layout trl_dqn_learn_fused_u8_ring takes)."""
    lay = view['layout']
    B = view['idx'].numel()
    return (self.u8_ok and lay.ncols == 5 and lay.storage_dtypes[0] == torch.uint8 and lay.storage_dtypes[3] == torch.uint8
            and lay.col_bytes[0] == self.S and lay.col_bytes[3] == self.S and lay.col_bytes[1] == lay.col_bytes[2] ==
            lay.col_bytes[4] == 4 and all(d == torch.float32 for d in (lay.storage_dtypes[1], lay.storage_dtypes[2],
                                                                        lay.storage_dtypes[4]))
            and lay.stride % 16 == 0 and lay.offsets[0] % 16 == 0 and lay.offsets[3] % 16 == 0
            and B >= 32 and B % 32 == 0 and B <= self.MAX_BATCH and view['capacity'] < 2 ** 31)

  def from_ring(self, view, gamma, optim, double_dqn=False, update=True, huber_delta=None):
    """The step on rows read in place out of a replay ring (view: TensorTupleDataset.ring_view): no gathered batch,
    the first layer fetches the sampled frames where they lie (trl_dqn_learn_fused_u8_ring).  Returns the [1] td loss."""
    C = self._C
    if self._ptr_key() != self._key:
      raise _lib.TrlError('FusedDqnLearn: parameters were re-homed after construction')
    idx = view['idx']
    B = idx.numel()
    dev = view['ring'].device
    lib = _lib.lib()
    d = self.descs
    ws = self._ws.get(dev, lib.trl_dqn_learn_u8_ring_workspace_bytes(C.byref(d[0]), B))
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    args = optim.fused_args(update, True)
    kind, delta = (0, 0.0) if huber_delta is None else (1, float(huber_delta))
    with torch.cuda.device(dev):
      _lib.check(lib.trl_dqn_learn_fused_u8_ring(
          C.byref(d[0]), C.byref(d[1]), view['ring'].data_ptr(), view['capacity'], view['stride'], C.byref(view['layout'].c),
          view['head'], view['count'], idx.data_ptr(), B, float(gamma), 1 if double_dqn else 0, kind, delta, loss.data_ptr(),
          C.byref(args), ws.data_ptr(), _lib.stream_ptr(dev)))
    view['done']()
    return loss
