"""DDPG controller on the fused critic-target kernel (reference: ddpg_controller.py)."""
import os

import numpy as np
import torch

from ...controllers import Controller
from ... import config
from ... import functional as F
from ..models import DDPGActorNet, DDPGCriticNet
from ._common import resolve_device, need_hip, obs_to_tensor, FlatParams, FlatAdam, GradSync, LearnRunner


class OUNoise:
  """Ornstein-Uhlenbeck exploration noise, one state of `action_dim` shared by all envs
  (ddpg_controller.py:7-39): dx = theta * (mu - x) + sigma * N(0, 1), drawn from numpy's
  global generator; sigma anneals max_sigma -> min_sigma over decay_period calls."""

  def __init__(self, action_dim, mu=0.0, theta=0.15, max_sigma=0.3, min_sigma=0.3, decay_period=100000):
    self.action_dim, self.mu, self.theta = action_dim, mu, theta
    self.max_sigma, self.min_sigma, self.decay_period = max_sigma, min_sigma, decay_period
    self.sigma = max_sigma
    self.state, self.step = None, 0
    self.reset()

  def reset(self):
    self.state = np.ones(self.action_dim) * self.mu
    self.step = 0

  def evolve_state(self):
    drift = self.theta * (self.mu - self.state)
    self.state = self.state + (drift + self.sigma * np.random.randn(self.action_dim))
    return self.state

  def get_action(self, action):
    noisy = action + self.evolve_state()
    frac = min(1.0, self.step / self.decay_period)
    self.sigma = self.max_sigma - (self.max_sigma - self.min_sigma) * frac
    self.step += 1
    return noisy


def polyak_average_(source, target, tau=1e-3):
  """In-place target <- tau * source + (1 - tau) * target over two modules' parameters
  (ddpg_controller.py:42-57), for callers that hold plain modules.  The controller itself
  averages its flat buffers with one HIP launch (FlatParams.polyak_from)."""
  if not isinstance(source, torch.nn.Module) or not isinstance(target, torch.nn.Module):
    raise AssertionError('polyak_average_ takes two nn.Modules, got %s and %s' % (type(source), type(target)))
  with torch.no_grad():
    for s, t in zip(source.parameters(), target.parameters()):
      t.copy_(tau * s + (1.0 - tau) * t)


class DDPGController(Controller):
  def __init__(self, obs_size, action_size,
               action_low, action_high,
               actor_lr=1e-4, critic_lr=1e-3, gamma=0.99,
               tau=1e-2, n_reset_interval=100, device=None,
               hidden=256, huber_delta=None, process_group=None, graph=False, fused=None):
    self.device = resolve_device(device)
    self.action_low = action_low
    self.action_high = action_high

    # creation order = the reference's, so equal seeds give equal initial weights
    self.actor = DDPGActorNet(obs_size, action_size, hidden).to(self.device)
    self.target_actor = DDPGActorNet(obs_size, action_size, hidden).to(self.device)
    self._actor, self._target_actor = FlatParams(self.actor, fresh_grads=True), FlatParams(self.target_actor)
    self._target_actor.copy_from(self._actor)

    self.critic = DDPGCriticNet(obs_size, action_size, hidden).to(self.device)
    self.target_critic = DDPGCriticNet(obs_size, action_size, hidden).to(self.device)
    self._critic, self._target_critic = FlatParams(self.critic, fresh_grads=True), FlatParams(self.target_critic)
    self._target_critic.copy_from(self._critic)

    self._sync = GradSync(process_group)
    self._sync.broadcast_params(self._actor, self._target_actor, self._critic, self._target_critic)
    self.actor_optim = FlatAdam(self._actor, lr=actor_lr, grad_scale=self._sync.scale, device_state=graph)
    self.critic_optim = FlatAdam(self._critic, lr=critic_lr, grad_scale=self._sync.scale, device_state=graph)
    self._run = LearnRunner(self._learn_body, [self.actor_optim, self.critic_optim],
                            lambda: [self._target_actor.data, self._target_critic.data], self.device,
                            graph and not self._sync.enabled)

    # the whole step as 10 phases in one launch, or 10 launches (csrc/trl_offpolicy.hip) for two-hidden-layer stacks (eager, or captured once and replayed with graph=True: the Adam step counts then live on the device);
    # config.fused_learn = False / fused=False keeps the autograd path
    self._fused = None
    if self.device.type == 'cuda' and (config.fused_learn if fused is None else fused):
      lins = [[m for m in net.net if isinstance(m, torch.nn.Linear)]
              for net in (self.actor, self.critic, self.target_actor, self.target_critic)]
      if F.FusedDdpgLearn.supports(*lins):
        self._fused = F.FusedDdpgLearn(*lins, self._actor, self._critic, self._target_actor, self._target_critic)

    self.gamma = gamma
    self.tau = tau
    self.n_reset_interval = n_reset_interval
    self.huber_delta = huber_delta
    self.noise = OUNoise(action_size)
    self._act_rng = F.ActRng()

  def act(self, obs):
    with torch.no_grad():
      action = self.actor(obs_to_tensor(obs, self.device)).cpu().numpy()
    action = np.clip(self.noise.get_action(action), self.action_low, self.action_high)
    if self.noise.step % self.n_reset_interval == 0:
      self.noise.reset()
    return action

  def act_device(self, obs):
    """act() kept on the device (ddpg_controller.py:87-99): the OU state lives in a device tensor
    and its Gaussian increments come from a counter-based generator inside the one act kernel; returns a float32 device tensor
    [n, A].  Same process (theta, sigma schedule, reset interval, clipping) as act(), different
    random stream."""
    need_hip(self.device, 'DDPGController.act_device')
    obs = obs_to_tensor(obs, self.device)
    noise = self.noise
    with torch.no_grad():
      action = self.actor(obs)
      if getattr(self, '_ou_dev', None) is None:
        # the process carries on from wherever act() has taken it: same state, same place in the sigma / reset schedule
        self._ou_dev = [torch.as_tensor(np.asarray(noise.state, dtype=np.float32), device=self.device).clone()
                        for _ in range(2)]
        self._ou_cur = 0
      if noise.step == 0:                         # fresh or just reset: the state starts at mu (OUNoise.reset)
        self._ou_dev[self._ou_cur].fill_(float(noise.mu))
      src, dst = self._ou_dev[self._ou_cur], self._ou_dev[1 - self._ou_cur]
      # one launch: the OU step on the shared state and the clipped noisy actions (csrc/trl_act.hip)
      action = F.act_ou_noise(action, src, dst, noise.theta, noise.mu, noise.sigma, self.action_low, self.action_high,
                              self._act_rng)
      self._ou_cur = 1 - self._ou_cur
      noise.sigma = noise.max_sigma - (noise.max_sigma - noise.min_sigma) * min(1.0, noise.step / noise.decay_period)
      noise.step += 1
    if noise.step % self.n_reset_interval == 0:
      noise.reset()
    return action

  def _learn_body(self, obs, action, reward, next_obs, done):
    if self._fused is not None and self._fused.usable(obs):
      solo = not self._sync.enabled
      losses = self._fused(obs, action, reward, next_obs, done, self.gamma, self.tau, self.actor_optim,
                           self.critic_optim, update=solo, huber_delta=self.huber_delta)
      if not solo:      # gradients only: all-reduce, then the flat-buffer updates
        self._sync.all_reduce(self._actor)
        self.actor_optim.step()
        self._sync.all_reduce(self._critic)
        self.critic_optim.step()
        self._target_actor.polyak_from(self._actor, self.tau)
        self._target_critic.polyak_from(self._critic, self.tau)
      return losses

    actor_loss = -self.critic(obs, self.actor(obs)).mean()

    current_q = self.critic(obs, action)
    with torch.no_grad():
      next_q = self.target_critic(next_obs, self.target_actor(next_obs))
    critic_loss = F.ddpg_critic_loss(current_q, next_q, reward, done, self.gamma,
                                     huber_delta=self.huber_delta)

    # Actor step first.  The reference back-propagates actor_loss into the critic's .grad too
    # and then throws that away with critic_optim.zero_grad(); restricting the backward to
    # the actor's parameters yields the same actor gradient without the wasted matmuls.
    self.actor_optim.zero_grad()
    torch.autograd.backward(actor_loss, inputs=self._actor.params)
    self._sync.all_reduce(self._actor)
    self.actor_optim.step()

    self.critic_optim.zero_grad()
    critic_loss.backward()
    self._sync.all_reduce(self._critic)
    self.critic_optim.step()

    self._target_actor.polyak_from(self._actor, self.tau)
    self._target_critic.polyak_from(self._critic, self.tau)
    return torch.stack([actor_loss.detach(), critic_loss.detach()])

  def learn(self, obs, action, reward, next_obs, done):
    need_hip(self.device, 'DDPGController.learn')
    self.actor.train()
    self.critic.train()
    losses = self._run(obs, action, reward, next_obs, done)
    self.actor_optim.note_step()
    self.critic_optim.note_step()
    return dict(actor_loss=losses[0], critic_loss=losses[1])
