"""Pin the CPU oracle (oracle/) to the golden vectors generated from the reference
(tests/golden/make_golden.py).  Bit-exact for storage, indices and returns; losses and
updated weights are compared at 1e-6 relative (they run the same torch-CPU ops, so they
are in practice exact as well)."""
import numpy as np
import pytest
import torch

from oracle import replay_oracle, sampler_oracle, gae_oracle, loss_oracle
from torchrl_amd.contrib import models
from tests.helpers import FIELDS, state_from, t, flat, sub, transitions_from

torch.set_num_threads(1)


def test_replay_oracle_matches_reference(golden):
  g = golden('storage.npz')
  for si in range(int(g['n_scenarios'])):
    p = 's%d_' % si
    ds = replay_oracle.FifoColumns(size=int(g[p + 'size']))
    assert len(ds) == 0
    for pi, _k in enumerate(g[p + 'pushes']):
      ds.extend_transitions(transitions_from(g, '%spush%d_' % (p, pi)))
      assert len(ds) == int(g[p + 'lens'][pi])
      if (p + 'after%d_obs' % pi) in g.files:
        for fi, f in enumerate(FIELDS):
          exp = g['%safter%d_%s' % (p, pi, f)]
          got = ds[slice(None)][fi]
          assert got.dtype == np.float32 and got.shape == exp.shape
          np.testing.assert_array_equal(got, exp)
    idx = g[p + 'idx']
    for fi, f in enumerate(FIELDS):
      np.testing.assert_array_equal(ds[idx][fi], g[p + 'gather_' + f])
      np.testing.assert_array_equal(ds[0][fi], g[p + 'item0_' + f])
    ds.truncate()
    assert len(ds) == 0
  ds = replay_oracle.FifoColumns(size=3)
  ds.extend_transitions([])
  assert len(ds) == int(g['empty_len']) == 0
  # dtype promotion of generic columns (storage.py:35 + torch.cat): float64 stays float64, int64 -> float32
  tt = replay_oracle.FifoColumns(size=4)
  tt.extend(np.arange(12, dtype=np.float64).reshape(6, 2), np.arange(6, dtype=np.int64).reshape(6, 1))
  a, b = tt[slice(None)]
  assert a.dtype == g['tt_a'].dtype == np.float64 and b.dtype == g['tt_b'].dtype == np.float32
  np.testing.assert_array_equal(a, g['tt_a'])
  np.testing.assert_array_equal(b, g['tt_b'])


def test_sampler_oracle_matches_torch_randperm(golden):
  g = golden('sampler.npz')
  for ci in range(int(g['n_cases'])):
    st = sampler_oracle.Mt19937Stream(int(g['c%d_seed' % ci]))
    for j, (n, b) in enumerate(g['c%d_calls' % ci]):
      got = sampler_oracle.randperm_prefix(st, int(n), int(b))
      np.testing.assert_array_equal(got, g['c%d_out%d' % (ci, j)])


def test_sampler_oracle_full_permutation():
  torch.manual_seed(11)
  exp = torch.randperm(1000).numpy()
  got = sampler_oracle.randperm(sampler_oracle.Mt19937Stream(11), 1000)
  np.testing.assert_array_equal(got, exp)


def _values_ext(g, p):
  v = np.array(g[p + 'values'])
  boot = np.zeros((1, 1), np.float32) if g[p + 'done'][-1, 0] else np.array(g[p + 'next_value'])
  return v, boot


def test_gae_oracle_matches_reference_bitwise(golden):
  g = golden('returns.npz')
  for kind, n in (('a2c', int(g['n_a2c'])), ('ppo', int(g['n_ppo']))):
    for ci in range(n):
      p = '%s%d_' % (kind, ci)
      v, boot = _values_ext(g, p)
      ret, adv = gae_oracle.gae_numpy(g[p + 'reward'], v, boot.reshape(1), float(g[p + 'gamma']),
                                      float(g[p + 'lmbda']))
      np.testing.assert_array_equal(ret, g[p + 'returns'])
      if kind == 'ppo':
        np.testing.assert_array_equal(adv, g[p + 'adv'])
      # the per-env torch loop used as the CPU baseline
      ve = torch.from_numpy(np.concatenate([v, boot], 0))
      loop = gae_oracle.compute_return_loop(ve, t(g[p + 'reward']), float(g[p + 'gamma']), float(g[p + 'lmbda']))
      np.testing.assert_array_equal(loop.numpy(), g[p + 'returns'])


def test_gae_oracle_block_tn(golden):
  g = golden('returns.npz')
  rew = np.array(g['blk_reward'])[:, :, 0].T            # [T, N]
  val = np.array(g['blk_values'])[:, :, 0].T
  done_last = np.array(g['blk_done'])[:, -1, 0]
  boot = np.where(done_last != 0, 0.0, np.array(g['blk_next_value'])[:, 0, 0]).astype(np.float32)
  ret, _ = gae_oracle.gae_numpy(rew, val, boot, float(g['blk_gamma']), float(g['blk_lmbda']))
  np.testing.assert_array_equal(ret.T, np.array(g['blk_returns'])[:, :, 0])


def test_model_init_matches_reference_under_same_seed(golden):
  """Same torch seed -> same initial weights as the reference's controllers build."""
  g = golden('dqn.npz')
  torch.manual_seed(400)
  q = models.QNet(4, 2)
  for k, v in state_from(g, 'dqn0_q0').items():
    assert torch.equal(q.state_dict()[k], v), k
  g = golden('ddpg.npz')
  torch.manual_seed(500)
  actor = models.DDPGActorNet(3, 1, 256)
  _target_actor = models.DDPGActorNet(3, 1, 256)
  critic = models.DDPGCriticNet(3, 1, 256)
  for k, v in state_from(g, 'ddpg0_actor0').items():
    assert torch.equal(actor.state_dict()[k], v), k
  for k, v in state_from(g, 'ddpg0_critic0').items():
    assert torch.equal(critic.state_dict()[k], v), k
  g = golden('a2c.npz')
  torch.manual_seed(600)
  net = models.A2CNet(4, 2, 256)
  for k, v in state_from(g, 'a2c0_net0').items():
    assert torch.equal(net.state_dict()[k], v), k
  g = golden('ppo.npz')
  torch.manual_seed(701)
  net = models.ActorCriticNet(3, 2, 256)
  for k, v in state_from(g, 'ppo1_net0').items():
    assert torch.equal(net.state_dict()[k], v), k


def _close(a, b, rtol=1e-6, atol=5e-8):
  # 1e-6 relative (north star); the absolute floor is one f32 ulp of the O(1) per-sample terms a
  # loss is the mean of: a surrogate that cancels to ~1e-4 moves by ~2e-9 with the host's
  # vector ISA (torch-CPU reduction order), seen between two build containers.
  np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def _close_w(a, b):
  """Updated weights sit behind those host-dependent gradients (see _close_grad) and Adam's
  g / (sqrt(v) + eps): 1e-5 relative, not part of the north star's 1e-6 bar (losses are)."""
  np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=1e-5, atol=1e-7)


def _close_grad(a, b, rtol=1e-6):
  """Gradients come out of torch-CPU GEMMs whose summation order depends on the host's
  vector ISA (the fixtures were written on another CPU model than some test hosts): elements
  that cancel to near zero carry an absolute error of a few ulps of the LARGEST terms, so the
  absolute floor is relative to the largest gradient."""
  b = np.asarray(b)
  np.testing.assert_allclose(np.asarray(a), b, rtol=rtol, atol=4e-6 * float(np.abs(b).max()))


def test_dqn_oracle_matches_reference(golden):
  g = golden('dqn.npz')
  for ci in range(int(g['n_dqn'])):
    p = 'dqn%d_' % ci
    A = g[p + 'q0.net.4.bias'].shape[0]
    q, tq = models.QNet(4, A), models.QNet(4, A)
    q.load_state_dict(state_from(g, p + 'q0'))
    tq.load_state_dict(state_from(g, p + 't0'))
    opt = torch.optim.Adam(q.parameters(), lr=1e-3)
    for step in range(3):
      batch = [t(g['%sstep%d_%s' % (p, step, f)]) for f in FIELDS]
      info = loss_oracle.dqn_learn(q, tq, opt, *batch, gamma=0.99, double_dqn=bool(g[p + 'double']))
      if (step + 1) % 2 == 0:
        tq.load_state_dict(q.state_dict())
      _close(info['td_loss'], g['%sstep%d_td_loss' % (p, step)])
      _close_grad(sub(flat(q, grad=True)), g['%sstep%d_grad' % (p, step)])
      _close_w(sub(flat(q)), g['%sstep%d_q' % (p, step)])
      _close_w(sub(flat(tq)), g['%sstep%d_t' % (p, step)])


def test_ddpg_oracle_matches_reference(golden):
  g = golden('ddpg.npz')
  for ci in range(int(g['n_ddpg'])):
    p = 'ddpg%d_' % ci
    A = g[p + 'actor0.net.4.bias'].shape[0]
    actor, ta = models.DDPGActorNet(3, A, 256), models.DDPGActorNet(3, A, 256)
    critic, tc = models.DDPGCriticNet(3, A, 256), models.DDPGCriticNet(3, A, 256)
    actor.load_state_dict(state_from(g, p + 'actor0')); ta.load_state_dict(actor.state_dict())
    critic.load_state_dict(state_from(g, p + 'critic0')); tc.load_state_dict(critic.state_dict())
    ao = torch.optim.Adam(actor.parameters(), lr=1e-4)
    co = torch.optim.Adam(critic.parameters(), lr=1e-3)
    for step in range(2):
      batch = [t(g['%sstep%d_%s' % (p, step, f)]) for f in FIELDS]
      info = loss_oracle.ddpg_learn(actor, critic, ta, tc, ao, co, *batch, gamma=0.99, tau=1e-2)
      _close(info['actor_loss'], g['%sstep%d_actor_loss' % (p, step)])
      _close(info['critic_loss'], g['%sstep%d_critic_loss' % (p, step)])
      for nm, m in (('actor', actor), ('critic', critic), ('target_actor', ta), ('target_critic', tc)):
        _close_w(sub(flat(m)), g['%sstep%d_%s' % (p, step, nm)])


def test_a2c_oracle_matches_reference(golden):
  g = golden('a2c.npz')
  for ci in range(int(g['n_a2c'])):
    p = 'a2c%d_' % ci
    A = g[p + 'net0.actor.2.bias'].shape[0]
    net = models.A2CNet(4, A, 256)
    net.load_state_dict(state_from(g, p + 'net0'))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for step in range(2):
      batch = [t(g['%sstep%d_%s' % (p, step, f)]) for f in FIELDS + ('returns',)]
      info = loss_oracle.a2c_learn(net, opt, *batch, alpha=0.5, beta=float(g[p + 'beta']))
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        _close(info[k], g['%sstep%d_%s' % (p, step, k)])
      _close_grad(flat(net, grad=True), g['%sstep%d_grad' % (p, step)])
      _close_w(flat(net), g['%sstep%d_net' % (p, step)])


def test_ppo_oracle_matches_reference(golden):
  g = golden('ppo.npz')
  for ci in range(int(g['n_ppo'])):
    p = 'ppo%d_' % ci
    A = g[p + 'net0.log_std'].shape[1]
    net = models.ActorCriticNet(3, A, 256)
    net.load_state_dict(state_from(g, p + 'net0'))
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)
    for step in range(2):
      names = FIELDS + ('returns', 'old_log_probs', 'advantages')
      batch = [t(g['%sstep%d_%s' % (p, step, f)]) for f in names]
      info = loss_oracle.ppo_learn(net, opt, *batch, clip=0.2, alpha=0.5, beta=float(g[p + 'beta']),
                                   max_grad_norm=float(g[p + 'max_grad_norm']))
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        _close(info[k], g['%sstep%d_%s' % (p, step, k)])
      _close_grad(flat(net, grad=True), g['%sstep%d_grad' % (p, step)])
      _close_w(flat(net), g['%sstep%d_net' % (p, step)])


def _close_net(a, b):
  """compute_return runs the value / policy MLP first (torch-CPU GEMM: bit-identical to the
  fixture only on the CPU model that wrote it); the scan behind it is pinned bit-for-bit on its
  own by the test_gae_* cases above, which feed the fixture's values in."""
  b = np.asarray(b)
  np.testing.assert_allclose(np.asarray(a), b, rtol=1e-5, atol=2e-6 * max(1.0, float(np.abs(b).max())))


def test_compute_return_oracle_matches_reference(golden):
  g = golden('returns.npz')
  for ci in range(int(g['n_ppo'])):
    p = 'ppo%d_' % ci
    A = g[p + 'net.log_std'].shape[1]
    net = models.ActorCriticNet(3, A, 256)
    net.load_state_dict(state_from(g, p + 'net'))
    batch = [t(g[p + f]) for f in FIELDS]
    ret, lp, v = loss_oracle.ppo_compute_return(net, *batch, gamma=float(g[p + 'gamma']), lmbda=float(g[p + 'lmbda']))
    _close_net(ret.numpy(), g[p + 'returns'])
    _close_net(lp.numpy(), g[p + 'log_probs'])
    _close_net(v.numpy(), g[p + 'values'])
  for ci in range(int(g['n_a2c'])):
    p = 'a2c%d_' % ci
    net = models.A2CNet(4, 2, 256)
    net.load_state_dict(state_from(g, p + 'net'))
    batch = [t(g[p + f]) for f in FIELDS]
    ret = loss_oracle.a2c_compute_return(net, *batch, gamma=float(g[p + 'gamma']), lmbda=float(g[p + 'lmbda']))
    _close_net(ret.numpy(), g[p + 'returns'])
