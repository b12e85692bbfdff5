"""DQN / Double-DQN controller on the fused TD kernel (reference: dqn_controller.py).

Same constructor, `act` and `learn` as the reference.  `learn` computes the three Q tables
with torch matmuls and hands them to one HIP kernel that forms the (double) target, the loss
and dL/dQ in a single pass (csrc/trl_loss.hip: k_td_loss); Adam and the hard target sync are
one launch each over flat parameter buffers.
"""
import os

import numpy as np
import torch

from ...controllers import Controller
from ...utils.schedule import ExpDecaySchedule
from ... import config
from ... import functional as F
from ..models import QNet, _fusable_stack
from ._common import resolve_device, need_hip, obs_to_tensor, FlatParams, FlatAdam, GradSync, LearnRunner


def epsilon_greedy(action_size: int, choices: np.ndarray, eps: float = 0.1):
  """Batched epsilon-greedy over greedy `choices` (dqn_controller.py:8-24).

  Row i is drawn from the categorical eps/A + (1 - eps) * onehot(choices[i]) with one
  `np.random.choice` call per row, so numpy's global stream is consumed exactly as in the
  reference.  Returns [n, 1]."""
  n = len(choices)
  pmf = np.full((n, action_size), eps / action_size, dtype=np.float32)
  pmf[np.arange(n), choices] += 1.0 - eps
  support = np.arange(action_size)
  picks = np.array([np.random.choice(support, p=row) for row in pmf])
  return picks[:, None]


class DQNController(Controller):
  def __init__(self, obs_size, action_size,
               double_dqn=False, gamma=.99, lr=1e-3,
               eps_max=1.0, eps_min=1e-2, n_eps_anneal=1000,
               n_update_interval=5, device=None,
               huber_delta=None, hidden=128, process_group=None, graph=False, fused=None):
    self.action_size = action_size
    self.device = resolve_device(device)

    self.q_net = QNet(obs_size, action_size, hidden).to(self.device)
    self.target_q_net = QNet(obs_size, action_size, hidden).to(self.device)
    # stacks the whole-network kernels take (<= 64 wide) get their gradients written in place by the
    # fused backward; the reference's 128-wide QNet goes through autograd, where fresh gradient
    # tensors + one multi-tensor copy replace a fill and six accumulate launches per step
    fusable = _fusable_stack(list(self.q_net.net)) is not None
    self._sync = GradSync(process_group)
    # (not under graph replay with a gradient exchange: the replayed backward writes into the capture's own
    # gradient tensors, which collect_grads() outside the graph never sees again -- there the gradients
    # accumulate into the flat views, as in the PPO / A2C controllers)
    self._q = FlatParams(self.q_net, fresh_grads=not fusable and not (graph and self._sync.enabled))
    self._target = FlatParams(self.target_q_net)
    self._target.copy_from(self._q)            # target_q_net.load_state_dict(q_net.state_dict())
    self._sync.broadcast_params(self._q, self._target)
    self.q_net_optim = FlatAdam(self._q, lr=lr, grad_scale=self._sync.scale, device_state=graph)
    if fusable:     # one use per backward, zeroed grads: let the fused backward write .grad itself
      self.q_net.net.direct_grads = True
    self._run = LearnRunner(self._learn_body, [self.q_net_optim], lambda: [], self.device, graph)

    # the step up to Adam as 6 launches (csrc/trl_offpolicy.hip) for two-hidden-layer Q-networks up
    # to 1024 wide / fan-in (any fan-in for uint8 frames), eager or graph-replayed; config.fused_learn = False / fused=False keeps the autograd path
    self._fused = None
    if self.device.type == 'cuda' and (config.fused_learn if fused is None else fused):
      lins = [[m for m in net.net if isinstance(m, torch.nn.Linear)] for net in (self.q_net, self.target_q_net)]
      if F.FusedDqnLearn.supports(*lins):
        self._fused = F.FusedDqnLearn(*lins, self._q, self._target)

    self.double_dqn = double_dqn
    self.gamma = gamma
    self.n_update_interval = n_update_interval
    self.huber_delta = huber_delta             # None = the reference's MSE
    self.eps = ExpDecaySchedule(start=eps_max, end=eps_min, num_steps=n_eps_anneal)
    self._steps = 0
    self._act_rng = F.ActRng()

  def act(self, obs):
    with torch.no_grad():
      greedy = self.q_net(obs_to_tensor(obs, self.device)).argmax(dim=-1).cpu().numpy()
    actions = epsilon_greedy(self.action_size, greedy, self.eps.value)
    return [a[0] for a in actions]

  def act_device(self, obs):
    """act() without the device -> host round trip (dqn_controller.py:57-59): epsilon-greedy drawn
    by one HIP kernel from a counter-based generator (Philox), actions returned as an int64 device tensor [n].  Same action
    distribution as act() (greedy w.p. 1 - eps + eps/A, any other action w.p. eps/A); the draws
    themselves differ because the reference uses numpy's global generator on the host."""
    need_hip(self.device, 'DQNController.act_device')
    obs = obs_to_tensor(obs, self.device)
    with torch.no_grad():
      q = self.q_net(obs)
    return F.act_epsilon_greedy(q, self.eps.value, self._act_rng)      # one launch (csrc/trl_act.hip)

  def _learn_body(self, obs, action, reward, next_obs, done):
    if self._fused is not None and self._fused.usable(obs):
      return self._fused(obs, action, reward, next_obs, done, self.gamma, self.q_net_optim,
                         double_dqn=self.double_dqn, update=not self._sync.enabled,
                         huber_delta=self.huber_delta)[0]
    q = self.q_net(obs)
    with torch.no_grad():
      q_next_target = self.target_q_net(next_obs)
      q_next_online = self.q_net(next_obs) if self.double_dqn else None
    loss = F.td_loss(q, q_next_target, action, reward, done, self.gamma,
                     q_next_online=q_next_online, huber_delta=self.huber_delta)
    self.q_net_optim.zero_grad()
    loss.backward()
    if not self._sync.enabled:
      self.q_net_optim.step()
    return loss.detach()

  def learn_indexed(self, buffer, indices, in_place=False):
    """`self.learn(*[b.to(device) for b in buffer[indices]])` (examples/dqn.py:40-42) from the buffer and the sampled row
    numbers.  in_place=True: where the buffer is a ring of Transitions with uint8 frames and the Q-network is on the fused
    uint8 path, no batch is gathered -- the first layer and its weight gradient read the sampled frames where the ring
    holds them (trl_dqn_learn_fused_u8_ring: the gather as the layer's operand fetch).  Same result, bit for bit.
    It is NOT the default because it measured slower (0.127 against 0.120 ms per step, B = 256, 4 M-record ring): the
    step reads every sampled frame two to three times (obs: forward + weight gradient; next_obs: target and online
    pass), the dense 14 MB batch turns those re-reads into L2 / Infinity-Cache hits, and the 9 us copy that makes it is
    cheaper than re-reading scattered 28 KB records of a 226 GB ring.  indices: int64 tensor of
    logical rows (host: validated like torch indexing; device: not), e.g. buffer.sample_indices(batch_size)."""
    need_hip(self.device, 'DQNController.learn_indexed')
    f = self._fused
    if in_place and f is not None and not self._sync.enabled and self._run.graphed is None and hasattr(buffer, 'ring_view'):
      view = buffer.ring_view(indices)
      if f.ring_usable(view):
        self.q_net.train()
        loss = f.from_ring(view, self.gamma, self.q_net_optim, double_dqn=self.double_dqn, update=True,
                           huber_delta=self.huber_delta)[0]
        return self._after_step(loss)
      view['done']()
    return self.learn(*buffer[indices])

  def learn(self, obs, action, reward, next_obs, done):
    need_hip(self.device, 'DQNController.learn')
    self.q_net.train()
    loss = self._run(obs, action, reward, next_obs, done)
    if self._sync.enabled:
      self._sync.all_reduce(self._q)
      self.q_net_optim.step()
    return self._after_step(loss)

  def _after_step(self, loss):
    """What follows the optimizer step in DQNController.learn (dqn_controller.py:84-88): the step count, the periodic hard
    target sync, the loss dict."""
    self.q_net_optim.note_step()

    self._steps += 1
    if self._steps % self.n_update_interval == 0:
      self._target.copy_from(self._q)

    # 0-d tensor left on the device: the reference's `.cpu()` here is a host sync per step
    return dict(td_loss=loss)
