"""PPO controller on the GAE scan and the fused clipped-surrogate kernel
(reference: ppo_controller.py)."""
import os

import torch

from ...controllers import Controller
from ... import config
from ... import functional as F
from ..models import ActorCriticNet
from ._common import resolve_device, need_hip, obs_to_tensor, FlatParams, FlatAdam, GradSync, LearnRunner


class PPOController(Controller):
  TILES_MIN_BATCH = 32      # below this (a last partial minibatch) the layer-by-layer path

  def __init__(self, obs_size, action_size,
               gamma=0.99, lmbda=1.0, lr=1e-3,
               alpha=0.5, beta=1.0, clip_ratio=0.2,
               max_grad_norm=1.0, device=None,
               hidden=256, process_group=None, graph=False, fused=None, products=None):
    self.device = resolve_device(device)

    self.ac_net = ActorCriticNet(obs_size, action_size, hidden).to(self.device)
    self._flat = FlatParams(self.ac_net)
    self._sync = GradSync(process_group)
    self._sync.broadcast_params(self._flat)
    # clip_grad_value_ (an elementwise clamp, ppo_controller.py:72-73) and the 1/world_size of
    # the data-parallel mean are folded into the Adam launch, i.e. applied after the reduction
    self.ac_net_optim = FlatAdam(self._flat, lr=lr, grad_scale=self._sync.scale, clip_value=max_grad_norm,
                                 device_state=graph)
    for stack in (self.ac_net.critic, self.ac_net.actor):     # one use per backward, zeroed grads: let the fused backward write .grad itself
      stack.direct_grads = True
    self._epoch = None
    self._tiles = None
    self._fused = None    # whole-step kernel for <= 64-wide stacks (csrc/trl_learn.hip); config.fused_learn = False / fused=False disables
    if self.device.type == 'cuda' and (config.fused_learn if fused is None else fused):
      lins = [[m for m in stack if isinstance(m, torch.nn.Linear)] for stack in (self.ac_net.critic, self.ac_net.actor)]
      if F.FusedPpoLearn.supports(lins[0], lins[1], self.ac_net.log_std):
        self._fused = F.FusedPpoLearn(lins[0], lins[1], self.ac_net.log_std, products=products or config.learn_products)
      # the minibatch loop in one launch (learn_epoch, csrc/trl_epoch.hip) takes some shapes the whole-step kernel does not
      if self._fused is not None and self._fused.epoch_supported():
        self._epoch = self._fused
      elif F.FusedPpoLearn.epoch_supports(lins[0], lins[1], self.ac_net.log_std):
        cand = F.FusedPpoLearn(lins[0], lins[1], self.ac_net.log_std)
        self._epoch = cand if cand.epoch_supported() else None
      # one big batch through the tile pipeline (learn_tiles): the epoch kernel's shapes and, with one hidden layer, up to 31
      # observations and six actions (ActorCriticNet(17, 6, 256))
      self._tiles = self._epoch
      if self._tiles is None and F.FusedPpoLearn.tiles_supports(lins[0], lins[1]):
        self._tiles = self._fused if self._fused is not None else F.FusedPpoLearn(lins[0], lins[1], self.ac_net.log_std)
    self._run = LearnRunner(self._learn_body, [self.ac_net_optim], lambda: [], self.device, graph)

    self.gamma = gamma
    self.lmbda = lmbda
    self.alpha = alpha
    self.beta = beta
    self.clip_ratio = clip_ratio
    self.max_grad_norm = max_grad_norm
    self._act_rng = F.ActRng()

  def act(self, obs):
    with torch.no_grad():
      _, dist = self.ac_net(obs_to_tensor(obs, self.device))
    return dist.sample().cpu().numpy()

  def act_device(self, obs):
    """act() without the `.cpu().numpy()` host sync: the sampled actions as a device tensor."""
    need_hip(self.device, 'PPOController.act_device')
    obs = obs_to_tensor(obs, self.device)
    with torch.no_grad():
      if self._fused is not None and self._fused.usable(obs):
        _, mean, _ = self._fused.forward(obs, want_value=False)         # forward-only kernel, mean alone
      else:
        mean = self.ac_net.actor(obs)
      return F.act_normal(mean, self.ac_net.log_std, self._act_rng)      # one launch (csrc/trl_act.hip)

  def compute_return(self, obs, action, reward, next_obs, done):
    """One env's rollout [T, .] -> (returns [T, 1], log_probs [T, A], values [T, 1])
    (ppo_controller.py:32-51)."""
    T = reward.shape[0]
    returns, _, log_probs, values = self.compute_return_batched(
        obs.unsqueeze(1), action.reshape(T, 1, -1), reward.reshape(T, 1), next_obs[-1:],
        done.reshape(T, 1)[-1])
    return returns.reshape(T, 1), log_probs.reshape(T, -1), values.reshape(T, 1)

  def compute_return_batched(self, obs, action, reward, last_next_obs, last_done, done=None,
                             masked=False, env_major_out=False):
    """N envs at once: obs [T, N, obs_size], action [T, N, A], reward [T, N], last_next_obs
    [N, obs_size], last_done [N] -> (returns, advantages, log_probs [T, N, A], values [T, N]).
    With env_major_out, returns/advantages come back [N, T], the minibatch order of
    examples/ppo.py:61-68.

    A rollout kept in the blocked layout (RolloutBuffer(layout='blocked'): obs [N / E, T, E, obs_size],
    action [N / E, T, E, A], reward [N / E, T, E], E = 8 or 16) is recognised by its reward's three
    dimensions: the value pass then writes its values in that order and the scan streams every env
    group's [T, E] run (functional.gae_blocked); everything comes back blocked, [N / E, T, E(, A)]."""
    need_hip(self.device, 'PPOController.compute_return')
    if reward.dim() == 3:
      return self._compute_return_blocked(obs, action, reward, last_next_obs, last_done, done, masked, env_major_out)
    T, N = reward.shape
    with torch.no_grad():
      flat_obs = obs.reshape(T * N, -1)
      if self._fused is not None and self._fused.usable(flat_obs) and self._fused.usable(last_next_obs):
        # forward-only kernel: values and log-probs straight from registers, no activations written
        values, _, log_probs = self._fused.forward(flat_obs, action.reshape(T * N, -1), want_mean=False)
        bootstrap, _, _ = self._fused.forward(last_next_obs, want_mean=False)
        bootstrap = torch.where(last_done.reshape(N) != 0, torch.zeros_like(bootstrap), bootstrap)
        values = values.view(T, N)
        returns, adv = F.gae(reward, values, bootstrap, self.gamma, self.lmbda, done=done, masked=masked,
                             env_major_out=env_major_out)
        return returns, adv, log_probs.view(T, N, -1), values
      if self._tiles_forward_ok(flat_obs):
        # shapes outside the register-chained kernels: the tile pipeline, forward only (trl_ac_forward_tiles)
        values, _, log_probs = self._tiles.forward_tiles(flat_obs, action.reshape(T * N, -1))
        bootstrap = self._tiles.forward_tiles(last_next_obs, critic_only=True)[0]
        bootstrap = torch.where(last_done.reshape(N) != 0, torch.zeros_like(bootstrap), bootstrap)
        values = values.view(T, N)
        returns, adv = F.gae(reward, values, bootstrap, self.gamma, self.lmbda, done=done, masked=masked,
                             env_major_out=env_major_out)
        return returns, adv, log_probs.view(T, N, -1), values
      values, mean, log_std = self.ac_net.heads(flat_obs)
      values = values.view(T, N)
      bootstrap = self.ac_net.critic(last_next_obs).view(N)
      bootstrap = torch.where(last_done.reshape(N) != 0, torch.zeros_like(bootstrap), bootstrap)
      returns, adv = F.gae(reward, values, bootstrap, self.gamma, self.lmbda, done=done, masked=masked,
                           env_major_out=env_major_out)
      log_probs = F.normal_log_prob(mean, log_std, action.reshape(T * N, -1)).view(T, N, -1)
    return returns, adv, log_probs, values

  def _tiles_forward_ok(self, flat_obs):
    return (self._tiles is not None and flat_obs.is_cuda and flat_obs.dtype == torch.float32
            and flat_obs.shape[0] >= self.TILES_MIN_BATCH)

  def _compute_return_blocked(self, obs, action, reward, last_next_obs, last_done, done, masked, env_major_out):
    G, T, E = reward.shape
    N = G * E
    with torch.no_grad():
      flat_obs = obs.reshape(G * T * E, -1)
      flat_act = action.reshape(G * T * E, -1)
      if self._fused is not None and self._fused.usable(flat_obs) and self._fused.usable(last_next_obs):
        values, _, log_probs = self._fused.forward(flat_obs, flat_act, want_mean=False)
        bootstrap, _, _ = self._fused.forward(last_next_obs, want_mean=False)
      elif self._tiles_forward_ok(flat_obs):
        values, _, log_probs = self._tiles.forward_tiles(flat_obs, flat_act)
        bootstrap = self._tiles.forward_tiles(last_next_obs, critic_only=True)[0]
      else:
        values, mean, log_std = self.ac_net.heads(flat_obs)
        bootstrap = self.ac_net.critic(last_next_obs).view(N)
        log_probs = F.normal_log_prob(mean, log_std, flat_act)
      bootstrap = torch.where(last_done.reshape(N) != 0, torch.zeros_like(bootstrap), bootstrap.reshape(N))
      values = values.view(G, T, E)
      if G > 1 and reward.stride(0) != T * E:       # a [:, :T] slice of a larger-capacity buffer: give value the same strides
        buf = torch.empty((G, reward.stride(0) // E, E), dtype=torch.float32, device=reward.device)
        buf[:, :T].copy_(values)
        values = buf[:, :T]
      returns, adv = F.gae_blocked(reward, values, bootstrap, self.gamma, self.lmbda, done=done, masked=masked,
                                   env_major_out=env_major_out)
    return returns, adv, log_probs.view(G, T, E, -1), values

  def _learn_body(self, obs, action, returns, old_log_probs, advantages):
    """Device work of one learn(): forward, fused loss, backward and (single rank) the update."""
    if self._fused is not None and self._fused.usable(obs):
      solo = not self._sync.enabled
      in_kernel = solo and not self.ac_net_optim.device_state    # eager single rank: update inside the reduce launch
      losses = self._fused(obs, action, returns, old_log_probs, advantages, self.clip_ratio, self.alpha, self.beta,
                           adam=self.ac_net_optim if in_kernel else None)
      if solo and not in_kernel:
        self.ac_net_optim.step()
      return losses[:3]
    if (self._tiles is not None and not self.ac_net_optim.device_state and obs.is_cuda and obs.dim() == 2
        and obs.shape[0] >= self.TILES_MIN_BATCH):
      # shapes outside the register-chained kernels (e.g. three to six actions, a 256-wide layer over 10..31 observations):
      # the tile pipeline spread over the chip instead of the layer-by-layer path
      losses = self._tiles.learn_tiles(obs, action, returns, old_log_probs, advantages, self.clip_ratio, self.alpha,
                                       self.beta, self.ac_net_optim, update=not self._sync.enabled)
      return losses[:3]
    value, mean, log_std = self.ac_net.heads(obs)
    loss, parts = F.ppo_loss(mean, log_std, value, action, old_log_probs, advantages, returns,
                             self.clip_ratio, self.alpha, self.beta)
    self.ac_net_optim.zero_grad()
    loss.backward()
    if not self._sync.enabled:
      self.ac_net_optim.step()
    return parts

  def learn(self, obs, action, reward, next_obs, done,  # pylint: disable=unused-argument
            returns, old_log_probs, advantages):
    need_hip(self.device, 'PPOController.learn')
    parts = self._run(obs, action, returns, old_log_probs, advantages)
    if self._sync.enabled:
      self._sync.all_reduce(self._flat)
      self.ac_net_optim.step()
    self.ac_net_optim.note_step()
    return dict(actor_loss=parts[0], critic_loss=parts[1], entropy_loss=parts[2])

  def learn_epoch(self, dataset, batch_size, n_epochs=1):
    """The reference's learner loop -- `for epoch in range(n_epochs): for minibatch in DataLoader(dataset, batch_size):
    self.learn(*minibatch)` (examples/ppo.py:66-73; sequential, un-shuffled, last partial batch kept) -- as ONE launch
    where the networks allow it (functional.FusedPpoLearn.epoch_supported: a single rank, eager mode, one hidden layer
    up to 256 wide or two up to 64, batch_size <= 4096): weights and Adam moments stay on the CU between the updates
    (csrc/trl_epoch.hip).  Anything else runs the loop of learn() calls.

    dataset: the eight columns of examples/ppo.py:66 -- a TensorTupleDataset, or a sequence (obs, action, reward,
    next_obs, done, returns, old_log_probs, advantages); reward / next_obs / done are not read (may be None).
    Returns dict(actor_loss, critic_loss, entropy_loss) of [K] tensors, one entry per learn() step in order."""
    need_hip(self.device, 'PPOController.learn_epoch')
    cols = dataset[:] if hasattr(dataset, 'truncate') else list(dataset)
    if len(cols) != 8:
      raise ValueError('learn_epoch takes the eight columns of examples/ppo.py:66, got %d' % len(cols))
    obs, action, _reward, _next_obs, _done, returns, old_log_probs, advantages = cols
    n, batch_size, n_epochs = obs.shape[0], int(batch_size), int(n_epochs)
    f = self._epoch
    if (f is not None and not self._sync.enabled and not self.ac_net_optim.device_state
        and self._run.graphed is None and 1 <= batch_size <= f.EPOCH_MAX_MINIBATCH and obs.is_cuda and n >= 1):
      losses = f.learn_epoch(obs, action, returns, old_log_probs, advantages, batch_size, n_epochs, self.clip_ratio,
                             self.alpha, self.beta, self.ac_net_optim)
      return dict(actor_loss=losses[:, 0], critic_loss=losses[:, 1], entropy_loss=losses[:, 2])
    out = dict(actor_loss=[], critic_loss=[], entropy_loss=[])
    for _ in range(n_epochs):
      for s in range(0, n, batch_size):
        e = min(n, s + batch_size)
        info = self.learn(obs[s:e], action[s:e], None, None, None, returns[s:e], old_log_probs[s:e], advantages[s:e])
        for k in out:
          out[k].append(info[k].reshape(()))
    return {k: torch.stack(v) for k, v in out.items()}
