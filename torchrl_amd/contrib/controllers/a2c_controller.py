"""A2C controller on the GAE scan and fused loss kernels (reference: a2c_controller.py)."""
import os

import torch

from ...controllers import Controller
from ... import config
from ... import functional as F
from ..models import A2CNet
from ._common import resolve_device, need_hip, obs_to_tensor, FlatParams, FlatAdam, GradSync, LearnRunner


class A2CController(Controller):
  def __init__(self, obs_size, action_size,
               gamma=0.99, lmbda=1.0, lr=1e-3,
               alpha=0.5, beta=1.0, device=None,
               hidden=256, process_group=None, graph=False, fused=None, products=None):
    self.device = resolve_device(device)

    self.ac_net = A2CNet(obs_size, action_size, hidden).to(self.device)
    self._flat = FlatParams(self.ac_net)
    self._sync = GradSync(process_group)
    self._sync.broadcast_params(self._flat)
    self.ac_net_optim = FlatAdam(self._flat, lr=lr, grad_scale=self._sync.scale, device_state=graph)
    for stack in (self.ac_net.critic, self.ac_net.actor):     # one use per backward, zeroed grads: let the fused backward write .grad itself
      stack.direct_grads = True
    self._fused = None    # whole-step kernel for <= 64-wide stacks (csrc/trl_learn.hip); config.fused_learn = False / fused=False disables
    self._tiles = None
    if self.device.type == 'cuda' and (config.fused_learn if fused is None else fused):
      lins = [[m for m in stack if isinstance(m, torch.nn.Linear)] for stack in (self.ac_net.critic, self.ac_net.actor)]
      if F.FusedPpoLearn.supports(lins[0], lins[1], None):
        self._fused = F.FusedPpoLearn(lins[0], lins[1], None, products=products or config.learn_products)
      elif F.FusedPpoLearn.tiles_supports(lins[0], lins[1], a2c=True):
        # shapes outside the register-chained kernels (three to six actions, a wide layer over 10..31 observations):
        # the tile pipeline spread over the chip (trl_a2c_learn_tiles, csrc/trl_epoch.hip)
        self._tiles = F.FusedPpoLearn(lins[0], lins[1], None)
    self._run = LearnRunner(self._learn_body, [self.ac_net_optim], lambda: [], self.device, graph)

    self.gamma = gamma
    self.lmbda = lmbda
    self.alpha = alpha
    self.beta = beta
    self._act_rng = F.ActRng()

  def act(self, obs):
    with torch.no_grad():
      _, dist = self.ac_net(obs_to_tensor(obs, self.device))
    return dist.sample().cpu().numpy()

  def act_device(self, obs):
    """act() without the `.cpu().numpy()` host sync: the sampled actions as a device tensor."""
    need_hip(self.device, 'A2CController.act_device')
    with torch.no_grad():
      _, probs = self.ac_net.heads(obs_to_tensor(obs, self.device))
      return F.act_categorical(probs, self._act_rng)                      # one launch (csrc/trl_act.hip)

  def compute_return(self, obs, action, reward, next_obs, done):  # pylint: disable=unused-argument
    """One env's rollout [T, .] -> returns [T, 1] (a2c_controller.py:29-45)."""
    return self.compute_return_batched(obs.unsqueeze(1), reward.reshape(-1, 1),
                                       next_obs[-1:], done.reshape(-1, 1)[-1])[0].reshape(-1, 1)

  def compute_return_batched(self, obs, reward, last_next_obs, last_done, done=None, masked=False,
                             env_major_out=False):
    """N envs at once: obs [T, N, obs_size], reward [T, N], last_next_obs [N, obs_size],
    last_done [N] -> (returns, advantages, values), each [T, N] ([N, T] with env_major_out
    for the first two)."""
    need_hip(self.device, 'A2CController.compute_return')
    blocked = reward.dim() == 3       # [N / E, T, E]: RolloutBuffer(layout='blocked'), see PPOController
    if blocked:
      G, T, E = reward.shape
      N = G * E
    else:
      T, N = reward.shape
    with torch.no_grad():
      flat_obs = obs.reshape(T * N, -1)
      if self._fused is not None and self._fused.usable(flat_obs) and self._fused.usable(last_next_obs):
        values = self._fused.forward(flat_obs, want_mean=False)[0]      # forward-only kernel
        bootstrap = self._fused.forward(last_next_obs, want_mean=False)[0]
      elif (self._tiles is not None and flat_obs.is_cuda and flat_obs.dtype == torch.float32
            and flat_obs.shape[0] >= 32):
        values = self._tiles.forward_tiles(flat_obs)[0]                 # the tile pipeline, forward only, critic alone
        bootstrap = self._tiles.forward_tiles(last_next_obs)[0]
      else:
        values = self.ac_net.critic(flat_obs)
        bootstrap = self.ac_net.critic(last_next_obs).view(N)
      bootstrap = torch.where(last_done.reshape(N) != 0, torch.zeros_like(bootstrap), bootstrap)
      if blocked:
        values = values.view(G, T, E)
        if G > 1 and reward.stride(0) != T * E:
          buf = torch.empty((G, reward.stride(0) // E, E), dtype=torch.float32, device=reward.device)
          buf[:, :T].copy_(values)
          values = buf[:, :T]
        returns, adv = F.gae_blocked(reward, values, bootstrap, self.gamma, self.lmbda, done=done, masked=masked,
                                     env_major_out=env_major_out)
      else:
        values = values.view(T, N)
        returns, adv = F.gae(reward, values, bootstrap, self.gamma, self.lmbda, done=done, masked=masked,
                             env_major_out=env_major_out)
    return returns, adv, values

  def _learn_body(self, obs, action, returns):
    if self._fused is not None and self._fused.usable(obs):
      solo = not self._sync.enabled
      in_kernel = solo and not self.ac_net_optim.device_state    # eager single rank: update inside the reduce launch
      losses = self._fused.a2c(obs, action, returns, self.alpha, self.beta,
                               adam=self.ac_net_optim if in_kernel else None)
      if solo and not in_kernel:
        self.ac_net_optim.step()
      return losses[:3]
    if (self._tiles is not None and not self.ac_net_optim.device_state and obs.is_cuda and obs.dim() == 2
        and obs.dtype == torch.float32 and obs.shape[0] >= 32):
      losses = self._tiles.a2c_tiles(obs, action, returns, self.alpha, self.beta, self.ac_net_optim,
                                     update=not self._sync.enabled)
      return losses[:3]
    value, probs = self.ac_net.heads(obs)
    loss, parts = F.a2c_loss(probs, value, action, returns, self.alpha, self.beta)
    self.ac_net_optim.zero_grad()
    loss.backward()
    if not self._sync.enabled:
      self.ac_net_optim.step()
    return parts

  def learn(self, obs, action, reward, next_obs, done, returns):  # pylint: disable=unused-argument
    need_hip(self.device, 'A2CController.learn')
    parts = self._run(obs, action, returns)
    if self._sync.enabled:
      self._sync.all_reduce(self._flat)
      self.ac_net_optim.step()
    self.ac_net_optim.note_step()
    return dict(actor_loss=parts[0], critic_loss=parts[1], entropy_loss=parts[2])
