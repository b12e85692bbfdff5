"""HIP reverse-scan GAE vs the reference (golden fixtures) and the numpy oracle.
Bar: returns and advantages within 1e-6 relative (north star); the kernel keeps the
reference's op order, so the tests assert bit-equality, which is stronger."""
import numpy as np
import pytest
import torch

from oracle import gae_oracle
from torchrl_amd import functional as F
from torchrl_amd.contrib.controllers import PPOController, A2CController
from tests.helpers import FIELDS, state_from, t

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(reward, value, boot, gamma, lmbda, **kw):
  r, a = F.gae(torch.from_numpy(reward).to(DEV), torch.from_numpy(value).to(DEV),
               torch.from_numpy(boot).to(DEV), gamma, lmbda, **kw)
  torch.cuda.synchronize()
  return r.cpu().numpy(), a.cpu().numpy()


@pytest.mark.parametrize('T,N', [(1, 1), (5, 16), (20, 16), (127, 3), (128, 17), (129, 64), (300, 1000),
                                 (2048, 48), (64, 8192), (33, 16384 + 5)])
@pytest.mark.parametrize('lmbda', [1.0, 0.95])
def test_kernel_vs_oracle_bitwise(T, N, lmbda):
  rs = np.random.RandomState(T * 31 + N)
  reward = rs.randn(T, N).astype(np.float32)
  value = rs.randn(T, N).astype(np.float32)
  boot = rs.randn(N).astype(np.float32)
  boot[::3] = 0.0
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, lmbda)
  got_r, got_a = _run(reward, value, boot, 0.99, lmbda)
  np.testing.assert_array_equal(got_r, exp_r)
  np.testing.assert_array_equal(got_a, exp_a)
  # env-major output = the concatenation order of examples/ppo.py:61-64
  nt_r, nt_a = _run(reward, value, boot, 0.99, lmbda, env_major_out=True)
  np.testing.assert_array_equal(nt_r, exp_r.T)
  np.testing.assert_array_equal(nt_a, exp_a.T)


def test_full_size_config2():
  """BASELINE config 2: T=2048, N=4096, gamma=.99, lambda=.95."""
  T, N = 2048, 4096
  rs = np.random.RandomState(0)
  reward = rs.randn(T, N).astype(np.float32)
  value = rs.randn(T, N).astype(np.float32)
  boot = rs.randn(N).astype(np.float32)
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95)
  got_r, got_a = _run(reward, value, boot, 0.99, 0.95)
  np.testing.assert_array_equal(got_r, exp_r)
  np.testing.assert_array_equal(got_a, exp_a)
  np.testing.assert_allclose(got_r, exp_r, rtol=1e-6, atol=0)      # the stated bar


def test_masked_mode_vs_oracle():
  T, N = 200, 77
  rs = np.random.RandomState(9)
  reward, value = rs.randn(T, N).astype(np.float32), rs.randn(T, N).astype(np.float32)
  boot = rs.randn(N).astype(np.float32)
  done = (rs.rand(T, N) < 0.05).astype(np.float32)
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95, done=done, masked=True)
  got_r, got_a = _run(reward, value, boot, 0.99, 0.95, done=torch.from_numpy(done).to(DEV), masked=True)
  np.testing.assert_array_equal(got_r, exp_r)
  np.testing.assert_array_equal(got_a, exp_a)
  # with no dones the masked mode degenerates to the reference mode
  z = np.zeros_like(done)
  ref_r, _ = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95)
  got_r, _ = _run(reward, value, boot, 0.99, 0.95, done=torch.from_numpy(z).to(DEV), masked=True)
  np.testing.assert_array_equal(got_r, ref_r)


@pytest.mark.parametrize('T,N', [(5, 16), (128, 16), (129, 64), (300, 1000), (2048, 48), (64, 8192), (263, 16388),
                                 (2048, 4096)])
def test_masked_mode_pipelined_kernel_bitwise(T, N):
  """Masked extension on the pipelined kernel (N % 4 == 0: done rides the LDS ring as a third
  plane) vs the oracle's definition, both output layouts; fractional done values included."""
  rs = np.random.RandomState(T + 7 * N)
  reward, value = rs.randn(T, N).astype(np.float32), rs.randn(T, N).astype(np.float32)
  boot = rs.randn(N).astype(np.float32)
  done = (rs.rand(T, N) < 0.05).astype(np.float32)
  done[rs.rand(T, N) < 0.01] = 0.25
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95, done=done, masked=True)
  d = torch.from_numpy(done).to(DEV)
  got_r, got_a = _run(reward, value, boot, 0.99, 0.95, done=d, masked=True)
  np.testing.assert_array_equal(got_r, exp_r)
  np.testing.assert_array_equal(got_a, exp_a)
  nt_r, nt_a = _run(reward, value, boot, 0.99, 0.95, done=d, masked=True, env_major_out=True)
  np.testing.assert_array_equal(nt_r, exp_r.T)
  np.testing.assert_array_equal(nt_a, exp_a.T)


def test_linearity_property_full_size():
  """Size-independent check at full size: with lambda = 1 and V = 0 the return is the plain
  discounted sum, so returns(r1 + r2) ~= returns(r1) + returns(r2) and a single impulse at
  t0 gives gamma^(t0 - t)."""
  T, N = 2048, 4096
  reward = torch.zeros(T, N, device=DEV)
  reward[T - 1] = 1.0
  value = torch.zeros(T, N, device=DEV)
  boot = torch.zeros(N, device=DEV)
  ret, adv = F.gae(reward, value, boot, 0.99, 1.0)
  g = np.float32(0.99)
  exp = np.empty(T, np.float32)
  acc = np.float32(1.0)
  for i in range(T - 1, -1, -1):
    exp[i] = acc
    acc = np.float32(g * acc)
  assert torch.equal(ret[:, 0].cpu(), torch.from_numpy(exp))
  assert torch.equal(ret, ret[:, :1].expand(T, N)) and torch.equal(ret, adv)


def test_ppo_compute_return_golden(golden):
  """PPOController.compute_return on one env's rollout vs the reference's outputs."""
  g = golden('returns.npz')
  for ci in range(int(g['n_ppo'])):
    p = 'ppo%d_' % ci
    A = g[p + 'net.log_std'].shape[1]
    c = PPOController(3, A, gamma=float(g[p + 'gamma']), lmbda=float(g[p + 'lmbda']), device=DEV)
    c.ac_net.load_state_dict(state_from(g, p + 'net'))
    batch = [t(g[p + f], DEV) for f in FIELDS]
    ret, lp, v = c.compute_return(*batch)
    assert ret.shape == g[p + 'returns'].shape and lp.shape == g[p + 'log_probs'].shape
    # values come from a GPU matmul (different summation order than the CPU's): 1e-6 bar
    np.testing.assert_allclose(v.cpu().numpy(), g[p + 'values'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), g[p + 'log_probs'], rtol=1e-5, atol=1e-5)
    scale = np.abs(g[p + 'returns']).max()
    np.testing.assert_allclose(ret.cpu().numpy(), g[p + 'returns'], rtol=1e-5, atol=1e-5 * scale)
    # the scan itself, fed the reference's own values, is exact
    values = t(g[p + 'values'], DEV)
    boot = torch.zeros(1, device=DEV) if g[p + 'done'][-1, 0] else t(g[p + 'next_value'], DEV).view(1)
    r2, a2 = F.gae(batch[2].reshape(-1, 1), values.reshape(-1, 1), boot, float(g[p + 'gamma']), float(g[p + 'lmbda']))
    np.testing.assert_array_equal(r2.cpu().numpy(), g[p + 'returns'])
    np.testing.assert_array_equal(a2.cpu().numpy(), g[p + 'adv'])


def test_a2c_compute_return_golden(golden):
  g = golden('returns.npz')
  for ci in range(int(g['n_a2c'])):
    p = 'a2c%d_' % ci
    c = A2CController(4, 2, gamma=float(g[p + 'gamma']), lmbda=float(g[p + 'lmbda']), device=DEV)
    c.ac_net.load_state_dict(state_from(g, p + 'net'))
    batch = [t(g[p + f], DEV) for f in FIELDS]
    ret = c.compute_return(*batch)
    scale = max(1.0, np.abs(g[p + 'returns']).max())
    np.testing.assert_allclose(ret.cpu().numpy(), g[p + 'returns'], rtol=1e-5, atol=1e-5 * scale)
    values = t(g[p + 'values'], DEV)
    boot = torch.zeros(1, device=DEV) if g[p + 'done'][-1, 0] else t(g[p + 'next_value'], DEV).view(1)
    r2, _ = F.gae(batch[2].reshape(-1, 1), values.reshape(-1, 1), boot, float(g[p + 'gamma']), float(g[p + 'lmbda']))
    np.testing.assert_array_equal(r2.cpu().numpy(), g[p + 'returns'])


def test_block_of_envs_golden(golden):
  """N per-env reference rollouts as one [T, N] scan, both output layouts."""
  g = golden('returns.npz')
  rew = t(np.ascontiguousarray(np.array(g['blk_reward'])[:, :, 0].T), DEV)
  val = t(np.ascontiguousarray(np.array(g['blk_values'])[:, :, 0].T), DEV)
  done_last = np.array(g['blk_done'])[:, -1, 0]
  boot = t(np.where(done_last != 0, 0.0, np.array(g['blk_next_value'])[:, 0, 0]).astype(np.float32), DEV)
  ret, adv = F.gae(rew, val, boot, float(g['blk_gamma']), float(g['blk_lmbda']), env_major_out=True)
  exp = np.array(g['blk_returns'])[:, :, 0]
  np.testing.assert_array_equal(ret.cpu().numpy(), exp)
  np.testing.assert_array_equal(adv.cpu().numpy(), exp - np.array(g['blk_values'])[:, :, 0])
  # batched controller path: [T, N, .] in, env-major out
  c = PPOController(3, 1, gamma=float(g['blk_gamma']), lmbda=float(g['blk_lmbda']), device=DEV)
  c.ac_net.load_state_dict(state_from(g, 'blk_net'))
  obs = t(np.ascontiguousarray(np.array(g['blk_obs']).transpose(1, 0, 2)), DEV)
  act = t(np.ascontiguousarray(np.array(g['blk_action']).transpose(1, 0, 2)), DEV)
  last_next = t(np.array(g['blk_next_obs'])[:, -1, :], DEV)
  r3, a3, lp3, v3 = c.compute_return_batched(obs, act, rew, last_next, t(done_last, DEV), env_major_out=True)
  np.testing.assert_allclose(r3.cpu().numpy(), exp, rtol=1e-5, atol=1e-5 * np.abs(exp).max())
  np.testing.assert_allclose(lp3.cpu().numpy().transpose(1, 0, 2), g['blk_log_probs'], rtol=1e-5, atol=1e-5)


def test_block_of_envs_golden_through_forward_kernel(golden):
  """The same reference rollouts with compute_return_batched's value / log-prob pass forced through the
  forward-only kernel (k_ac_infer, 256-wide hidden blocks: the reference's ActorCriticNet(3, 1, 256))."""
  g = golden('returns.npz')
  c = PPOController(3, 1, gamma=float(g['blk_gamma']), lmbda=float(g['blk_lmbda']), device=DEV)
  assert c._fused is not None
  c._fused.MIN_BATCH = 1
  c.ac_net.load_state_dict(state_from(g, 'blk_net'))
  rew = t(np.ascontiguousarray(np.array(g['blk_reward'])[:, :, 0].T), DEV)
  obs = t(np.ascontiguousarray(np.array(g['blk_obs']).transpose(1, 0, 2)), DEV)
  act = t(np.ascontiguousarray(np.array(g['blk_action']).transpose(1, 0, 2)), DEV)
  last_next = t(np.array(g['blk_next_obs'])[:, -1, :], DEV)
  done_last = np.array(g['blk_done'])[:, -1, 0]
  assert c._fused.usable(obs.reshape(-1, 3)) and c._fused.usable(last_next)
  r3, a3, lp3, v3 = c.compute_return_batched(obs, act, rew, last_next, t(done_last, DEV), env_major_out=True)
  exp = np.array(g['blk_returns'])[:, :, 0]
  np.testing.assert_allclose(r3.cpu().numpy(), exp, rtol=1e-5, atol=1e-5 * np.abs(exp).max())
  np.testing.assert_allclose(lp3.cpu().numpy().transpose(1, 0, 2), g['blk_log_probs'], rtol=1e-5, atol=1e-5)
  np.testing.assert_allclose(a3.cpu().numpy(), exp - np.array(g['blk_values'])[:, :, 0], rtol=1e-5,
                             atol=1e-5 * np.abs(exp).max())


def _run_blocked(reward, value, boot, gamma, lmbda, E, cap=None, **kw):
  """reward / value [T, N] numpy -> blocked device planes (optionally inside a [G, cap, E] buffer)."""
  T, N = reward.shape
  planes = []
  for x in (reward, value) + ((kw.pop('done_np'),) if 'done_np' in kw else ()):
    b = F.to_blocked(torch.from_numpy(x).to(DEV), E)
    if cap is not None:
      buf = torch.full((N // E, cap, E), float('nan'), device=DEV)
      buf[:, :T] = b
      b = buf[:, :T]
    planes.append(b)
  if len(planes) == 3:
    kw['done'] = planes[2]
  r, a = F.gae_blocked(planes[0], planes[1], torch.from_numpy(boot).to(DEV), gamma, lmbda, **kw)
  torch.cuda.synchronize()
  return r, a


@pytest.mark.parametrize('E', [8, 16])
@pytest.mark.parametrize('T,N', [(1, 16), (5, 16), (63, 32), (64, 48), (65, 16), (129, 64), (300, 1008), (2048, 48),
                                 (64, 8192), (257, 4096 + 16)])
def test_blocked_layout_vs_oracle_bitwise(T, N, E):
  """The blocked rollout layout [N / E][T][E] (flat per-group streams) gives the oracle's numbers bit for bit,
  written blocked or env-major, also when the planes are [:, :T] slices of a larger-capacity buffer."""
  rs = np.random.RandomState(T * 17 + N + E)
  reward, value = rs.randn(T, N).astype(np.float32), rs.randn(T, N).astype(np.float32)
  boot = rs.randn(N).astype(np.float32)
  boot[::5] = 0.0
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95)
  for cap in (None, T + 37):
    r, a = _run_blocked(reward, value, boot, 0.99, 0.95, E, cap=cap)
    assert r.shape == (N // E, T, E)
    np.testing.assert_array_equal(F.from_blocked(r).cpu().numpy(), exp_r)
    np.testing.assert_array_equal(F.from_blocked(a).cpu().numpy(), exp_a)
    r, a = _run_blocked(reward, value, boot, 0.99, 0.95, E, cap=cap, env_major_out=True)
    np.testing.assert_array_equal(r.cpu().numpy(), exp_r.T)
    np.testing.assert_array_equal(a.cpu().numpy(), exp_a.T)


@pytest.mark.parametrize('E', [8, 16])
def test_blocked_full_size_config2(E):
  """BASELINE config 2 (T=2048, N=4096) in the blocked layout: bit-identical to the [T, N] parity mode."""
  T, N = 2048, 4096
  rs = np.random.RandomState(0)
  reward, value = rs.randn(T, N).astype(np.float32), rs.randn(T, N).astype(np.float32)
  boot = rs.randn(N).astype(np.float32)
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95)
  r, a = _run_blocked(reward, value, boot, 0.99, 0.95, E)
  np.testing.assert_array_equal(F.from_blocked(r).cpu().numpy(), exp_r)
  np.testing.assert_array_equal(F.from_blocked(a).cpu().numpy(), exp_a)
  tn_r, tn_a = _run(reward, value, boot, 0.99, 0.95)
  np.testing.assert_array_equal(F.from_blocked(r).cpu().numpy(), tn_r)
  np.testing.assert_array_equal(F.from_blocked(a).cpu().numpy(), tn_a)


@pytest.mark.parametrize('E', [8, 16])
def test_blocked_masked_mode(E):
  T, N = 263, 1024
  rs = np.random.RandomState(5)
  reward, value = rs.randn(T, N).astype(np.float32), rs.randn(T, N).astype(np.float32)
  boot = rs.randn(N).astype(np.float32)
  done = (rs.rand(T, N) < 0.05).astype(np.float32)
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95, done=done, masked=True)
  r, a = _run_blocked(reward, value, boot, 0.99, 0.95, E, done_np=done, masked=True)
  np.testing.assert_array_equal(F.from_blocked(r).cpu().numpy(), exp_r)
  np.testing.assert_array_equal(F.from_blocked(a).cpu().numpy(), exp_a)


def test_blocked_bad_arguments_raise():
  x = torch.zeros(4, 6, 16, device=DEV)
  with pytest.raises(ValueError):
    F.gae_blocked(x, x, torch.zeros(63, device=DEV), 0.99, 0.95)          # bootstrap is not [N]
  with pytest.raises(ValueError):
    F.gae_blocked(x.transpose(0, 1), x.transpose(0, 1), torch.zeros(96, device=DEV), 0.99, 0.95)   # not group-contiguous
  with pytest.raises(ValueError):
    F.gae_blocked(torch.zeros(4, 6, 12, device=DEV), torch.zeros(4, 6, 12, device=DEV), torch.zeros(48, device=DEV),
                  0.99, 0.95)                                               # 12 envs per group


def test_production_library_ignores_tuning_environment(monkeypatch):
  """TRL_GAE_TUNE drives the diagnostic build's ablation switches (no chain, no stores: wrong results by design);
  the shipped library must not react to it."""
  monkeypatch.setenv('TRL_GAE_TUNE', '16,2,15,1')
  T, N = 130, 64
  rs = np.random.RandomState(2)
  reward, value, boot = rs.randn(T, N).astype(np.float32), rs.randn(T, N).astype(np.float32), rs.randn(N).astype(np.float32)
  exp_r, exp_a = gae_oracle.gae_numpy(reward, value, boot, 0.99, 0.95)
  got_r, got_a = _run(reward, value, boot, 0.99, 0.95)
  np.testing.assert_array_equal(got_r, exp_r)
  np.testing.assert_array_equal(got_a, exp_a)
  r, a = _run_blocked(reward, value, boot, 0.99, 0.95, 16)
  np.testing.assert_array_equal(F.from_blocked(r).cpu().numpy(), exp_r)


def test_bad_arguments_raise():
  x = torch.zeros(4, 4, device=DEV)
  with pytest.raises(ValueError):
    F.gae(x, x[:2], x[0], 0.99, 0.95)
  with pytest.raises(Exception):
    F.gae(x.cpu(), x.cpu(), x[0].cpu(), 0.99, 0.95)
  with pytest.raises(ValueError):
    F.gae(x, x, x[0], 0.99, 0.95, masked=True)
