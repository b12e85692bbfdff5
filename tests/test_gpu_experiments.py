"""End-to-end experiments on the device (BASELINE config 1 plumbing): the assertions of the
reference's examples/test_examples.py on the built-in envs, plus rollout-buffer parity with
the reference-shaped per-env path."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))

from torchrl_amd import functional as F  # noqa: E402
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402
from torchrl_amd.utils.rollout import RolloutBuffer, env_major, iter_minibatches  # noqa: E402
from torchrl_amd.utils.storage import Transition, TransitionTupleDataset  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('graph', [False, True])
def test_dqn_exp(graph):
  from dqn import DQNExperiment
  exp = DQNExperiment(env_id='CartPole-v1', n_frames=600, buffer_size=500, n_train_interval=10, seed=0, graph=graph)
  exp.run()
  assert len(exp.buffer) == min(500, 600) and exp._cur_frames == 600
  assert exp.buffer.device.type == 'cuda'
  loss, step = exp.logger.last['train/td_loss']
  assert torch.isfinite(loss).item() and step == 600
  assert exp.controller._steps == exp.controller.q_net_optim.step_count > 0


def test_ddpg_exp():
  from ddpg import DDPGExperiment
  exp = DDPGExperiment(env_id='Pendulum-v0', n_frames=800, buffer_size=500, n_train_interval=20, seed=0)
  exp.run()
  assert len(exp.buffer) == 500 and exp._cur_frames == 800
  assert all(torch.isfinite(exp.logger.last['train/' + k][0]).item() for k in ('actor_loss', 'critic_loss'))


def test_a2c_exp():
  from a2c import A2CExperiment
  exp = A2CExperiment(env_id='CartPole-v1', n_envs=4, n_frames=600, seed=0)
  exp.run()
  assert exp._cur_frames == 600 and len(exp.rollout) == 0
  assert torch.isfinite(exp.logger.last['train/critic_loss'][0]).item()


def test_ppo_exp():
  from ppo import PPOExperiment
  exp = PPOExperiment(env_id='Pendulum-v0', n_envs=4, n_frames=800, rollout_steps=20, lmbda=0.95, seed=0)
  exp.run()
  assert exp._cur_frames == 800
  assert torch.isfinite(exp.logger.last['train/actor_loss'][0]).item()


def test_ppo_and_a2c_exp_through_whole_step_kernel():
  """Enough envs per rollout that learn() and compute_return take the whole-step / forward-only kernels
  on the reference's default 256-wide networks (csrc/trl_learn.hip, hidden blocks)."""
  from ppo import PPOExperiment
  from a2c import A2CExperiment
  from torchrl_amd import functional as F
  calls = {'learn': 0, 'a2c': 0, 'fwd': 0, 'epoch': 0}
  orig = (F.FusedPpoLearn.__call__, F.FusedPpoLearn.a2c, F.FusedPpoLearn.forward, F.FusedPpoLearn.learn_epoch)

  def count(key, fn):
    def wrapped(self, *a, **k):
      calls[key] += 1
      return fn(self, *a, **k)
    return wrapped
  F.FusedPpoLearn.__call__, F.FusedPpoLearn.a2c, F.FusedPpoLearn.forward, F.FusedPpoLearn.learn_epoch = (
      count('learn', orig[0]), count('a2c', orig[1]), count('fwd', orig[2]), count('epoch', orig[3]))
  try:
    exp = PPOExperiment(env_id='Pendulum-v0', n_envs=256, n_frames=256 * 8 * 2, rollout_steps=8, lmbda=0.95,
                        batch_size=512, n_epochs=2, seed=0)
    exp.run()
    assert exp.controller._fused is not None and exp._cur_frames == 256 * 8 * 2
    # 2 rollouts: each rollout's 2 epochs x 4 minibatches are ONE learn_epoch launch (examples/ppo.py -> PPOController.learn_epoch)
    assert calls['epoch'] == 2 and calls['learn'] == 0 and calls['fwd'] >= 4
    assert exp.controller.ac_net_optim.step_count == 2 * 2 * 4
    assert all(torch.isfinite(exp.logger.last['train/' + k][0]).item() for k in ('actor_loss', 'critic_loss'))
    exp = A2CExperiment(env_id='CartPole-v1', n_envs=256, n_frames=256 * 5 * 2, rollout_steps=5, seed=0)
    exp.run()
    assert exp.controller._fused is not None and calls['a2c'] == 2
    assert torch.isfinite(exp.logger.last['train/critic_loss'][0]).item()
  finally:
    F.FusedPpoLearn.__call__, F.FusedPpoLearn.a2c, F.FusedPpoLearn.forward, F.FusedPpoLearn.learn_epoch = orig


def test_rollout_buffer_matches_per_env_reference_path():
  """[T, N] staging + one batched compute_return == the reference's per-env buffers +
  per-env compute_return + torch.cat, in the reference's env-major minibatch order."""
  dev = 'cuda:0'
  T, N = 23, 5
  rs = np.random.RandomState(0)
  torch.manual_seed(0)
  ctl = PPOController(3, 2, gamma=0.99, lmbda=0.95, device=dev)
  roll = RolloutBuffer(N, device=dev, initial_steps=4)
  per_env = [TransitionTupleDataset(device=dev) for _ in range(N)]
  for t in range(T):
    tl = [Transition(rs.randn(3).astype(np.float32), rs.randn(2).astype(np.float32), float(rs.randn()),
                     rs.randn(3).astype(np.float32), bool(t == T - 1 and e % 2)) for e in range(N)]
    roll.extend(tl)
    for buf, tr in zip(per_env, tl):
      buf.extend([tr])
  assert len(roll) == T
  obs, action, reward, done, last_next = roll.tensors()
  ret, adv, lp, val = ctl.compute_return_batched(obs, action, reward, last_next, done[-1], env_major_out=True)
  cols = [env_major(obs), env_major(action), ret.reshape(-1, 1), env_major(lp), adv.reshape(-1, 1)]
  ref = [[] for _ in range(5)]
  for buf in per_env:                                   # examples/ppo.py:49-59
    batch = buf[:]
    r, l, v = ctl.compute_return(*batch)
    for dst, x in zip(ref, (batch[0], batch[1], r, l, r - v)):
      dst.append(x)
  ref = [torch.cat(c, 0) for c in ref]
  for got, exp in zip(cols, ref):
    assert got.shape == exp.shape
    torch.testing.assert_close(got, exp, rtol=1e-5, atol=1e-5)
  assert torch.equal(cols[0], ref[0]) and torch.equal(cols[1], ref[1])      # data columns: exact
  sizes = [b[0].shape[0] for b in iter_minibatches(cols, 32)]
  assert sizes == [32, 32, 32, 19]
  roll.truncate()
  assert len(roll) == 0


@pytest.mark.parametrize('E,T,N,steps0', [(16, 37, 48, 8), (8, 64, 32, 64), (16, 300, 512, 32)])
def test_blocked_rollout_buffer_equals_tn_buffer(E, T, N, steps0):
  """RolloutBuffer(layout='blocked') ([N / E, T, E, .], one flat run per env group for the scan) holds the same
  rollout as the [T, N, .] buffer and compute_return_batched gives the same numbers on it, bit for bit (PPO:
  whole-step forward kernel rows are independent; A2C the same), incl. growth past the initial capacity and
  planes that are [:, :T] slices of a larger-capacity allocation."""
  from torchrl_amd.contrib.controllers import A2CController
  dev = 'cuda:0'
  rs = np.random.RandomState(E + T)
  torch.manual_seed(1)
  ctl = PPOController(3, 1, gamma=0.99, lmbda=0.95, device=dev)
  a2c = A2CController(3, 2, gamma=0.99, lmbda=1.0, device=dev)
  tn = RolloutBuffer(N, device=dev, initial_steps=steps0)
  bl = RolloutBuffer(N, device=dev, initial_steps=steps0, layout='blocked', block_envs=E)
  for t in range(T):
    obs, act = rs.randn(N, 3).astype(np.float32), rs.randn(N, 1).astype(np.float32)
    rew, nxt = rs.randn(N).astype(np.float32), rs.randn(N, 3).astype(np.float32)
    tl = [Transition(obs[e], act[e], float(rew[e]), nxt[e], bool(t == T - 1 and e % 3 == 0)) for e in range(N)]
    tn.extend(tl)
    bl.extend(tl)
  o1, a1, r1, d1, n1 = tn.tensors()
  o2, a2, r2, d2, n2 = bl.tensors()
  assert o2.shape == (N // E, T, E, 3) and r2.shape == (N // E, T, E)
  assert torch.equal(F.from_blocked(o2), o1) and torch.equal(F.from_blocked(r2), r1) and torch.equal(n1, n2)
  last_done = d1[-1]
  assert torch.equal(F.from_blocked(d2)[-1], last_done)
  ret1, adv1, lp1, v1 = ctl.compute_return_batched(o1, a1, r1, n1, last_done)
  ret2, adv2, lp2, v2 = ctl.compute_return_batched(o2, a2, r2, n2, last_done)
  assert ret2.shape == (N // E, T, E) and lp2.shape == (N // E, T, E, 1)
  for x1, x2 in ((ret1, ret2), (adv1, adv2), (lp1, lp2), (v1, v2)):
    assert torch.equal(F.from_blocked(x2), x1)
  # env-major output from the blocked rollout = the reference's minibatch order (examples/ppo.py:61-68)
  ret3, adv3, _, _ = ctl.compute_return_batched(o2, a2, r2, n2, last_done, env_major_out=True)
  assert torch.equal(ret3, ret1.t()) and torch.equal(adv3, adv1.t())
  b1 = a2c.compute_return_batched(o1, r1, n1, last_done)
  b2 = a2c.compute_return_batched(o2, r2, n2, last_done)
  for x1, x2 in zip(b1, b2):
    assert torch.equal(F.from_blocked(x2), x1)
  with pytest.raises(ValueError):
    RolloutBuffer(20, device=dev, layout='blocked', block_envs=16)


def test_act_device_distribution_and_shapes():
  """f3: on-device exploration keeps the reference's action distribution (not its draws)."""
  from torchrl_amd.contrib.controllers import DQNController, DDPGController, A2CController
  dev = 'cuda:0'
  torch.manual_seed(0)
  c = DQNController(4, 4, eps_max=0.4, eps_min=0.4, device=dev)
  obs = torch.zeros(20000, 4, device=dev)
  with torch.no_grad():
    greedy = int(c.q_net(obs[:1]).argmax())
  a = c.act_device(obs)
  assert a.is_cuda and a.dtype == torch.int64 and a.shape == (20000,)
  freq = torch.bincount(a, minlength=4).float() / 20000
  for i in range(4):
    exp = 0.1 + (0.6 if i == greedy else 0.0)            # eps/A (+ 1 - eps for the greedy action)
    assert abs(float(freq[i]) - exp) < 0.02
  d = DDPGController(3, 2, -2.0, 2.0, device=dev, n_reset_interval=5)
  for _ in range(7):
    act = d.act_device(torch.randn(6, 3, device=dev))
  assert act.is_cuda and act.shape == (6, 2) and float(act.abs().max()) <= 2.0 and d.noise.step == 2
  p = PPOController(3, 2, device=dev)
  assert p.act_device(torch.randn(5, 3, device=dev)).shape == (5, 2)
  assert A2CController(4, 3, device=dev).act_device(torch.randn(5, 4, device=dev)).shape == (5,)


def test_act_kernels_distributions_and_reproducibility():
  """f3 as kernels (csrc/trl_act.hip): the reference draws its exploration from numpy's global generator on the
  host, so parity is distributional.  Checked: exact behaviour where no randomness is involved (eps = 0 is the
  argmax, sigma = 0 OU is the deterministic decay, clipping), the moments / frequencies of each draw, and that a
  draw is a function of (seed, call number) alone."""
  dev = torch.device('cuda:0')
  g = torch.Generator(device=dev).manual_seed(0)
  # ---- epsilon-greedy
  q = torch.randn(50000, 5, device=dev, generator=g)
  greedy = q.argmax(1)
  assert torch.equal(F.act_epsilon_greedy(q, 0.0, F.ActRng(1)), greedy)
  a = F.act_epsilon_greedy(q, 0.3, F.ActRng(1))
  assert torch.equal(a, F.act_epsilon_greedy(q, 0.3, F.ActRng(1)))                 # same seed, same call number
  r = F.ActRng(1)
  r.next()
  assert not torch.equal(a, F.act_epsilon_greedy(q, 0.3, r))                       # the next call draws afresh
  keep = float((a == greedy).float().mean())
  assert abs(keep - (0.7 + 0.3 / 5)) < 0.01
  off = a[a != greedy]
  freq = torch.bincount(off, minlength=5).float() / off.numel()
  assert float((freq - 0.2).abs().max()) < 0.02                                      # uniform over the actions
  # ---- Normal policy
  mean = torch.randn(40000, 2, device=dev, generator=g)
  log_std = torch.tensor([-0.5, 0.3], device=dev)
  x = F.act_normal(mean, log_std, F.ActRng(7))
  z = (x - mean) / log_std.exp()
  assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
  assert abs(float((z[:, 0] * z[:, 1]).mean())) < 0.02                               # the two dimensions are independent
  assert abs(float((z.abs() < 1.0).float().mean()) - 0.6827) < 0.01
  assert torch.equal(x, F.act_normal(mean, log_std, F.ActRng(7)))
  # ---- Categorical policy
  probs = torch.tensor([[0.1, 0.2, 0.7]], device=dev).repeat(60000, 1)
  c = F.act_categorical(probs, F.ActRng(3))
  f = torch.bincount(c, minlength=3).float() / c.numel()
  assert float((f - torch.tensor([0.1, 0.2, 0.7], device=dev)).abs().max()) < 0.01
  onehot = torch.eye(4, device=dev)[torch.randint(0, 4, (1000,), device=dev, generator=g)]
  assert torch.equal(F.act_categorical(onehot, F.ActRng(3)), onehot.argmax(1))       # a certain action is taken
  # ---- OU noise: one shared state; sigma = 0 is the deterministic decay towards mu, and clipping holds
  act = torch.zeros(6, 2, device=dev)
  s0, s1 = torch.tensor([1.0, -2.0], device=dev), torch.empty(2, device=dev)
  out = F.act_ou_noise(act, s0, s1, 0.15, 0.0, 0.0, -1.0, 1.0, F.ActRng(5))
  assert torch.allclose(s1, s0 * 0.85) and torch.equal(out, s1.clamp(-1, 1).expand(6, 2))
  incs = []
  rng = F.ActRng(5)
  cur = torch.zeros(2, device=dev)
  for _ in range(4000):
    nxt = torch.empty(2, device=dev)
    F.act_ou_noise(act, cur, nxt, 0.15, 0.0, 0.3, -9.0, 9.0, rng)
    incs.append((nxt - 0.85 * cur) / 0.3)
    cur = nxt
  incs = torch.stack(incs)
  assert abs(float(incs.mean())) < 0.05 and abs(float(incs.std()) - 1.0) < 0.05
  # the stationary spread of the process: sigma / sqrt(1 - (1 - theta)^2)
  # ---- through the controllers
  from torchrl_amd.contrib.controllers import DDPGController
  d = DDPGController(3, 2, -2.0, 2.0, device=dev, n_reset_interval=5)
  for _ in range(7):
    a = d.act_device(torch.randn(6, 3, device=dev))
  assert a.shape == (6, 2) and float(a.abs().max()) <= 2.0 and d.noise.step == 2
