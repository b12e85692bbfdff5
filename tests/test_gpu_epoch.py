"""The reference's PPO minibatch loop as one launch (csrc/trl_epoch.hip, C ABI trl_ppo_learn_epoch, PPOController.learn_epoch)
against (1) the oracle's torch-CPU restatement of the loop `for minibatch in loader: controller.learn(*minibatch)`
(examples/ppo.py:66-73 over ppo_controller.py:53-80; oracle/loss_oracle.py is pinned to the reference by
tests/golden/ppo.npz), (2) the same number of consecutive PPOController.learn() calls on the GPU, and (3) the
reference's own two-step outputs in tests/golden/ppo.npz.

Bars: the first step's losses 1e-6 relative (+ 2e-7 absolute for terms near 0) -- identical inputs; later steps see
weights that differ from the yardstick's by f32 summation order amplified by Adam (a round-off-sized difference of a
near-zero gradient is an lr-sized move: the bar tests/test_gpu_learn.py documents for one step), so their losses are held
to 2e-5 relative and the final weights to `steps x 2.5 lr` everywhere, with fewer than 2 % of them further than 2e-5
off."""
import copy

import numpy as np
import pytest
import torch

from oracle import loss_oracle
from torchrl_amd.contrib.controllers import PPOController

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LR = 3e-4


def _data(obs_size, A, n, seed, net):
  g = torch.Generator().manual_seed(seed)
  obs = torch.randn(n, obs_size, generator=g)
  act = torch.randn(n, A, generator=g)
  ret = torch.randn(n, 1, generator=g)
  adv = torch.randn(n, 1, generator=g)
  with torch.no_grad():
    mean = net.actor(obs)
    old = torch.distributions.Normal(mean, net.log_std.exp().expand_as(mean)).log_prob(act)
    old = old + 0.3 * torch.randn(n, A, generator=g)        # ratios on both sides of the clip range
  return obs, act, ret, old, adv


def _make(obs_size, A, hidden, seed, max_grad_norm=0.05):
  torch.manual_seed(seed)
  ctl = PPOController(obs_size, A, lr=LR, alpha=0.5, beta=1e-2, clip_ratio=0.2, max_grad_norm=max_grad_norm,
                      device=DEV, hidden=hidden)
  with torch.no_grad():
    ctl.ac_net.log_std.add_(torch.linspace(-0.3, 0.2, A, device=DEV).view(1, A))
  return ctl


def _cols(obs, act, ret, old, adv):
  d = lambda t: t.to(DEV)
  return [d(obs), d(act), None, None, None, d(ret), d(old), d(adv)]


# obs size, actions, hidden, minibatch, rows, epochs
CASES = [(3, 1, (64, 64), 64, 64 * 6, 1),          # the BASELINE network at the reference's spec minibatch
         (3, 1, (256,), 64, 64 * 5 + 17, 1),       # the reference's own network (ppo_controller.py:14), short last minibatch
         (3, 1, (256,), 32, 32 * 4, 2),            # the reference's default minibatch, two epochs
         (4, 2, (64, 64), 32, 200, 1),
         (8, 4, (128,), 48, 48 * 4 + 5, 1),
         (15, 1, (32, 16), 100, 350, 1),           # minibatch > one 64-sample tile; widest fan-in
         (3, 2, (256,), 200, 500, 1),
         (2, 2, (16,), 7, 30, 2),
         (1, 3, (48, 64), 64, 128, 1)]


@pytest.mark.parametrize('obs_size,A,hidden,B,n,epochs', CASES)
def test_learn_epoch_matches_the_oracle_loop(obs_size, A, hidden, B, n, epochs):
  ctl = _make(obs_size, A, hidden, obs_size + 10 * A + B)
  assert ctl._epoch is not None
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.Adam(ref.parameters(), lr=LR)
  obs, act, ret, old, adv = _data(obs_size, A, n, 7 + B, ref)
  exp = []
  for _ in range(epochs):
    for s in range(0, n, B):
      e = min(n, s + B)
      exp.append(loss_oracle.ppo_learn(ref, ref_opt, obs[s:e], act[s:e], None, None, None, ret[s:e], old[s:e], adv[s:e],
                                       clip=0.2, alpha=0.5, beta=1e-2, max_grad_norm=0.05))
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])      # the last step's, after clip_grad_value_
  K = len(exp)
  got = ctl.learn_epoch(_cols(obs, act, ret, old, adv), B, epochs)
  torch.cuda.synchronize()
  assert ctl.ac_net_optim.step_count == K
  for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
    assert got[k].shape == (K,)
    for j in range(K):
      e, v = float(exp[j][k]), float(got[k][j].cpu())
      tol = 1e-6 if j == 0 else 2e-5
      assert abs(v - e) <= tol * abs(e) + 2e-7 * (1 if j == 0 else 5), (k, j, v, e)
  exp_w = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
  got_w = ctl._flat.dense().cpu()
  d = (got_w - exp_w).abs()
  assert float(d.max()) <= 2.5 * LR * K, float(d.max())
  assert float((d > 2e-5).float().mean()) < 0.02, float((d > 2e-5).float().mean())
  got_grad = ctl._flat.dense(grad=True).cpu()
  scale = float(exp_grad.abs().max())
  close = ((got_grad - exp_grad).abs() <= 1e-3 * exp_grad.abs() + 1e-4 * scale).float().mean()
  assert float(close) > 0.98, float(close)      # (the last step's gradients, taken at slightly different weights)


@pytest.mark.parametrize('obs_size,A,hidden,B,n,epochs', [(3, 1, (64, 64), 64, 64 * 8, 1), (3, 1, (256,), 64, 64 * 8, 2),
                                                          (4, 2, (64, 64), 32, 32 * 5 + 32, 1)])
def test_learn_epoch_equals_consecutive_learn_calls(obs_size, A, hidden, B, n, epochs):
  """K learn() calls (two launches each: the whole-step kernel + reduce / Adam) against ONE learn_epoch launch: the same
  arithmetic in another summation order.  First step's losses 1e-6, moments and weights as above."""
  a, b = _make(obs_size, A, hidden, 5), _make(obs_size, A, hidden, 5)
  assert torch.equal(a._flat.data, b._flat.data)
  obs, act, ret, old, adv = _data(obs_size, A, n, 3, copy.deepcopy(a.ac_net).cpu())
  cols = _cols(obs, act, ret, old, adv)
  got = a.learn_epoch(cols, B, epochs)
  seq = dict(actor_loss=[], critic_loss=[], entropy_loss=[])
  for _ in range(epochs):
    for s in range(0, n, B):
      info = b.learn(*[c[s:s + B] if c is not None else None for c in cols])
      for k in seq:
        seq[k].append(float(info[k]))
  torch.cuda.synchronize()
  K = len(seq['actor_loss'])
  assert a.ac_net_optim.step_count == b.ac_net_optim.step_count == K
  for k in seq:
    for j in range(K):
      e, v = seq[k][j], float(got[k][j])
      assert abs(v - e) <= (1e-6 if j == 0 else 2e-5) * abs(e) + 2e-7 * (1 if j == 0 else 5), (k, j, v, e)
  d = (a._flat.data - b._flat.data).abs()
  assert float(d.max()) <= 2.5 * LR * K
  assert float((d > 2e-5).float().mean()) < 0.02
  dm = (a.ac_net_optim.exp_avg - b.ac_net_optim.exp_avg).abs()
  assert float(dm.max()) <= 1e-3 * float(b.ac_net_optim.exp_avg.abs().max()) + 1e-6


def test_learn_epoch_golden_reference_two_steps(golden):
  """tests/golden/ppo.npz: two consecutive learn() steps of the REFERENCE's PPOController on ActorCriticNet(3, A, 256).
  Where both steps' batches have the same size they are one learn_epoch call (minibatch = that size): both steps'
  losses, the second step's clipped gradients and the weights after it, at the fixtures' bars."""
  g = golden('ppo.npz')
  from tests.helpers import FIELDS, state_from, t
  ran = 0
  for ci in range(int(g['n_ppo'])):
    p = 'ppo%d_' % ci
    A = g[p + 'net0.log_std'].shape[1]
    names = FIELDS + ('returns', 'old_log_probs', 'advantages')
    steps = [[t(g['%sstep%d_%s' % (p, s, f)], DEV) for f in names] for s in range(2)]
    if steps[0][0].shape[0] != steps[1][0].shape[0] or A > 4:
      continue
    c = PPOController(3, A, lr=3e-4, alpha=0.5, beta=float(g[p + 'beta']), clip_ratio=0.2,
                      max_grad_norm=float(g[p + 'max_grad_norm']), device=DEV)
    c.ac_net.load_state_dict(state_from(g, p + 'net0'))
    assert c._epoch is not None
    B = steps[0][0].shape[0]
    cols = [torch.cat([steps[0][i], steps[1][i]], 0) for i in range(8)]
    info = c.learn_epoch(cols, B, 1)
    for s in range(2):
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        np.testing.assert_allclose(float(info[k][s].cpu()), float(g['%sstep%d_%s' % (p, s, k)]), rtol=1e-6 if s == 0 else 1e-5,
                                   atol=2e-7)
    exp_g, exp_w = np.asarray(g['%sstep1_grad' % p]), np.asarray(g['%sstep1_net' % p])
    np.testing.assert_allclose(c._flat.dense(grad=True).cpu().numpy(), exp_g, rtol=1e-3, atol=1e-4 * float(np.abs(exp_g).max()))
    np.testing.assert_allclose(c._flat.dense().cpu().numpy(), exp_w, rtol=1e-5, atol=1e-5 * float(np.abs(exp_w).max()) + 2.5 * 3e-4)
    ran += 1
  assert ran >= 1


def test_learn_epoch_is_reproducible_and_falls_back():
  ctl = _make(3, 1, (64, 64), 1)
  obs, act, ret, old, adv = _data(3, 1, 300, 2, copy.deepcopy(ctl.ac_net).cpu())
  cols = _cols(obs, act, ret, old, adv)
  outs = []
  for _ in range(2):
    c = _make(3, 1, (64, 64), 1)
    info = c.learn_epoch(cols, 64, 2)
    torch.cuda.synchronize()
    outs.append((torch.stack([info[k] for k in sorted(info)]).clone(), c._flat.data.clone(), c.ac_net_optim.exp_avg_sq.clone()))
  for x, y in zip(outs[0], outs[1]):
    assert torch.equal(x, y)      # fixed summation order: bit for bit
  # a shape outside the kernel's (two hidden layers wider than 64) runs the loop of learn() calls: same return type
  wide = _make(3, 1, (128, 64), 1)
  assert wide._fused is None
  info = wide.learn_epoch(cols, 64, 1)
  assert info['actor_loss'].shape == (5,) and wide.ac_net_optim.step_count == 5
  # a TensorTupleDataset of the eight columns (examples/ppo.py:66) is taken as is
  from torchrl_amd.utils.storage import TensorTupleDataset
  d = lambda x: x.to(DEV)
  ds = TensorTupleDataset(x=[d(obs), d(act), d(ret), d(obs), d(ret), d(ret), d(old), d(adv)])
  c = _make(3, 1, (64, 64), 1)
  info = c.learn_epoch(ds, 64, 2)
  assert torch.equal(torch.stack([info[k] for k in sorted(info)]), outs[0][0])


def test_fast_adam_is_trl_adam_elem_bit_for_bit():
  """The epoch kernel's Adam (divisions spelled out as reciprocal + two residual corrections, four elements at a time)
  must be trl_adam_elem -- the library's one Adam, checked against torch.optim.Adam in tests/test_gpu_losses.py -- bit for
  bit.  Runs tests/adam_compare_child.py on the diagnostic build (the comparison kernel lives there)."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  if not os.path.exists(os.path.join(root, 'torchrl_amd', 'lib', 'libtrl_hip_tune.so')):
    pytest.skip('diagnostic build not present')
  env = dict(os.environ, TRL_LIB='tune', PYTHONPATH=root)
  r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'adam_compare_child.py')], env=env, capture_output=True,
                     text=True, timeout=110)
  assert r.returncode == 0 and 'adam compare child ok' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


# ---------------------------------------------------------------- one big batch through the tile pipeline (trl_ppo_learn_tiles)
TILE_CASES = [(8, 4, (128,), 5000), (12, 3, (256,), 3000), (15, 1, (64, 64), 70001), (3, 4, (48, 32), 999), (10, 2, (256,), 40000),
              (3, 1, (64, 64), 65536), (4, 2, (256,), 131), (2, 3, (16,), 64),
              # the wide form (one hidden layer, two k rounds at layer 0, up to six outputs)
              (17, 6, (256,), 5000),      # ActorCriticNet(17, 6, 256): the reference's default width on a MuJoCo-sized task
              (24, 4, (256,), 3001), (31, 1, (64,), 999), (16, 2, (128,), 40000), (3, 6, (32,), 131), (8, 5, (256,), 70001)]


@pytest.mark.parametrize('obs_size,A,hidden,B', TILE_CASES)
def test_learn_tiles_matches_oracle(obs_size, A, hidden, B):
  """PPOController.learn on shapes the register-chained whole-step kernels do not take (three or four actions, a wide
  layer over 10..15 observations) runs the tile pipeline spread over the chip (trl_ppo_learn_tiles) instead of layer by
  layer; here against the oracle's torch-CPU learn(): tests/test_gpu_learn.py's bars (losses 1e-6, clipped gradients
  1e-4 / 2e-5 of the largest, weights within 2.5 lr and < 2 % further than 1e-5)."""
  ctl = _make(obs_size, A, hidden, obs_size + 10 * A + B)
  assert ctl._tiles is not None and (ctl._epoch is not None or obs_size > 15 or A > 4)
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.Adam(ref.parameters(), lr=LR)
  obs, act, ret, old, adv = _data(obs_size, A, B, 7 + B, ref)
  exp = loss_oracle.ppo_learn(ref, ref_opt, obs, act, None, None, None, ret, old, adv, clip=0.2, alpha=0.5, beta=1e-2,
                              max_grad_norm=0.05)
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
  cols = _cols(obs, act, ret, old, adv)
  if ctl._fused is None:
    got = ctl.learn(*cols)                      # the controller's own dispatch
  else:                                         # a shape the register-chained kernel takes too: call the tile path itself
    l4 = ctl._tiles.learn_tiles(cols[0], cols[1], cols[5], cols[6], cols[7], 0.2, 0.5, 1e-2, ctl.ac_net_optim)
    got = dict(actor_loss=l4[0], critic_loss=l4[1], entropy_loss=l4[2])
    tot = float(l4[3].cpu())
    assert abs(tot - (float(l4[0]) + 0.5 * float(l4[1]) - 1e-2 * float(l4[2]))) <= 1e-6 * abs(tot) + 1e-7
  torch.cuda.synchronize()
  assert ctl.ac_net_optim.step_count == 1
  for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
    e, v = float(exp[k]), float(got[k].cpu())
    assert abs(v - e) <= 1e-6 * abs(e) + 2e-7, (k, v, e)
  got_grad = ctl._flat.dense(grad=True).cpu()
  scale = float(exp_grad.abs().max())
  np.testing.assert_allclose(got_grad.numpy(), exp_grad.numpy(), rtol=1e-4, atol=2e-5 * scale)
  exp_w = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
  d = (ctl._flat.dense().cpu() - exp_w).abs()
  assert float(d.max()) <= 2.5 * LR
  assert float((d > 1e-5).float().mean()) < 0.02


def test_learn_tiles_is_reproducible_and_gradients_only():
  ctl = _make(8, 4, (128,), 3)
  obs, act, ret, old, adv = _data(8, 4, 20000, 4, copy.deepcopy(ctl.ac_net).cpu())
  c = _cols(obs, act, ret, old, adv)
  outs = []
  for _ in range(2):
    l4 = ctl._epoch.learn_tiles(c[0], c[1], c[5], c[6], c[7], 0.2, 0.5, 1e-2, ctl.ac_net_optim, update=False)
    torch.cuda.synchronize()
    outs.append((l4.clone(), ctl._flat.grad.clone()))
  assert ctl.ac_net_optim.step_count == 0                     # gradients only: nothing stepped
  assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # slabs added in a fixed order


# ---------------------------------------------------------------- A2C through the tile pipeline (trl_a2c_learn_tiles)
A2C_TILE_CASES = [(8, 4, (256,), 5000),       # A2CNet(8, 4, 256): the reference's default width, four actions
                  (4, 3, (256,), 3001), (12, 2, (128,), 2500), (15, 4, (128,), 999), (3, 3, (64, 64), 70001), (6, 4, (48, 32), 131),
                  (10, 2, (256,), 40000),
                  # the wide form
                  (17, 6, (256,), 5000), (20, 2, (128,), 3001), (8, 5, (64,), 999), (31, 3, (256,), 40000)]


@pytest.mark.parametrize('obs_size,A,hidden,B', A2C_TILE_CASES)
def test_a2c_learn_tiles_matches_oracle(obs_size, A, hidden, B):
  """A2CController.learn for the A2CNet shapes outside the register-chained kernels (three or four actions, a wide layer
  over 10 - 15 observations) through the tile pipeline, against the oracle's torch-CPU A2CController.learn
  (a2c_controller.py:47-69): losses 2e-6 relative (the slabs' sums are carried in double), gradients, weights after Adam."""
  from torchrl_amd.contrib.controllers import A2CController
  torch.manual_seed(obs_size + 10 * A + B)
  ctl = A2CController(obs_size, A, lr=1e-3, alpha=0.5, beta=1e-2, device=DEV, hidden=hidden)
  assert ctl._fused is None and ctl._tiles is not None
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
  g = torch.Generator().manual_seed(B)
  obs = torch.randn(B, obs_size, generator=g)
  act = torch.randint(0, A, (B, 1), generator=g).float()
  ret = torch.randn(B, 1, generator=g)
  exp = loss_oracle.a2c_learn(ref, ref_opt, obs, act, None, None, None, ret, alpha=0.5, beta=1e-2)
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
  got = ctl.learn(obs.to(DEV), act.to(DEV), None, None, None, ret.to(DEV))
  torch.cuda.synchronize()
  assert ctl.ac_net_optim.step_count == 1
  for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
    e, v = float(exp[k]), float(got[k].cpu())
    assert abs(v - e) <= 2e-6 * abs(e) + 2e-7, (k, v, e)
  got_grad = ctl._flat.dense(grad=True).cpu()
  np.testing.assert_allclose(got_grad.numpy(), exp_grad.numpy(), rtol=1e-4, atol=2e-5 * float(exp_grad.abs().max()))
  exp_w = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
  got_w = ctl._flat.dense().cpu()
  assert float((got_w - exp_w).abs().max()) <= 2.5e-3
  assert float(((got_w - exp_w).abs() > 1e-5).float().mean()) < 0.02


# ---------------------------------------------------------------- ragged tiny batches and refusals, through the C ABI entry itself
@pytest.mark.parametrize('obs_size,A,hidden', [(8, 4, (128,)), (17, 6, (256,)), (20, 5, (48,))])
@pytest.mark.parametrize('B', [1, 31, 65])
def test_learn_tiles_tiny_ragged_batches(obs_size, A, hidden, B):
  """One row, less than a tile, a tile and one row: the controller sends such batches down the layer path
  (TILES_MIN_BATCH), the entry itself takes any B >= 1 -- gradients only, against autograd on the oracle's loss."""
  ctl = _make(obs_size, A, hidden, 3 * B + A)
  ref = copy.deepcopy(ctl.ac_net).cpu()
  obs, act, ret, old, adv = _data(obs_size, A, B, 11 + B, ref)
  ref_opt = torch.optim.SGD(ref.parameters(), lr=0.0)        # (no update: the gradients are what is compared)
  exp = loss_oracle.ppo_learn(ref, ref_opt, obs, act, None, None, None, ret, old, adv, clip=0.2, alpha=0.5, beta=1e-2,
                              max_grad_norm=1e9)
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
  c = _cols(obs, act, ret, old, adv)
  l4 = ctl._tiles.learn_tiles(c[0], c[1], c[5], c[6], c[7], 0.2, 0.5, 1e-2, ctl.ac_net_optim, update=False)
  torch.cuda.synchronize()
  for i, k in enumerate(('actor_loss', 'critic_loss', 'entropy_loss')):
    e, v = float(exp[k]), float(l4[i].cpu())
    assert abs(v - e) <= 1e-6 * abs(e) + 2e-7, (k, v, e)
  got = ctl._flat.dense(grad=True).cpu()
  np.testing.assert_allclose(got.numpy(), exp_grad.numpy(), rtol=1e-4, atol=2e-5 * float(exp_grad.abs().max()))


def test_learn_tiles_refuses_what_it_does_not_take():
  """Two hidden layers stop at 15 observations / four actions, one hidden layer at 31 / six; the entry says so
  (TRL_EINVAL with a message) instead of computing something else."""
  from torchrl_amd import functional as F
  for obs_size, A, hidden in ((17, 2, (64, 64)), (3, 5, (64, 64)), (32, 2, (64,)), (4, 7, (64,))):
    ctl = _make(obs_size, A, hidden, 1)
    assert ctl._tiles is None
    lins = [[m for m in st if isinstance(m, torch.nn.Linear)] for st in (ctl.ac_net.critic, ctl.ac_net.actor)]
    f = ctl._fused if ctl._fused is not None else F.FusedPpoLearn(lins[0], lins[1], ctl.ac_net.log_std)
    obs, act, ret, old, adv = _data(obs_size, A, 64, 2, copy.deepcopy(ctl.ac_net).cpu())
    c = _cols(obs, act, ret, old, adv)
    with pytest.raises(Exception) as ei:
      f.learn_tiles(c[0], c[1], c[5], c[6], c[7], 0.2, 0.5, 1e-2, ctl.ac_net_optim, update=False)
    assert 'trl_ppo_learn_tiles' in str(ei.value)


@pytest.mark.parametrize('obs_size,A,hidden', [(8, 4, (256,)), (17, 6, (256,))])
@pytest.mark.parametrize('B', [1, 33, 130])
def test_a2c_learn_tiles_tiny_ragged_batches(obs_size, A, hidden, B):
  """As above for trl_a2c_learn_tiles, both ways of getting the advantage (8 observations: the forward-only kernel and one
  launch of both kinds of workgroup; 17: the critic's own workgroups, launched first)."""
  from torchrl_amd.contrib.controllers import A2CController
  torch.manual_seed(5 * B + A)
  ctl = A2CController(obs_size, A, lr=1e-3, alpha=0.5, beta=1e-2, device=DEV, hidden=hidden)
  assert ctl._fused is None and ctl._tiles is not None
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.SGD(ref.parameters(), lr=0.0)
  g = torch.Generator().manual_seed(B)
  obs = torch.randn(B, obs_size, generator=g)
  act = torch.randint(0, A, (B, 1), generator=g).float()
  ret = torch.randn(B, 1, generator=g)
  exp = loss_oracle.a2c_learn(ref, ref_opt, obs, act, None, None, None, ret, alpha=0.5, beta=1e-2)
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
  l4 = ctl._tiles.a2c_tiles(obs.to(DEV), act.to(DEV), ret.to(DEV), 0.5, 1e-2, ctl.ac_net_optim, update=False)
  torch.cuda.synchronize()
  for i, k in enumerate(('actor_loss', 'critic_loss', 'entropy_loss')):
    e, v = float(exp[k]), float(l4[i].cpu())
    assert abs(v - e) <= 2e-6 * abs(e) + 2e-7, (k, v, e)
  got = ctl._flat.dense(grad=True).cpu()
  np.testing.assert_allclose(got.numpy(), exp_grad.numpy(), rtol=1e-4, atol=2e-5 * float(exp_grad.abs().max()))


# ---------------------------------------------------------------- forward only through the tile pipeline (trl_ac_forward_tiles)
FWD_CASES = [(8, 4, (256,), 5000), (17, 6, (256,), 70001), (12, 3, (128,), 63), (15, 2, (64, 48), 40000), (3, 4, (32, 32), 1),
             (31, 5, (64,), 999), (24, 1, (256,), 33000)]


@pytest.mark.parametrize('obs_size,A,hidden,B', FWD_CASES)
def test_forward_tiles_matches_the_modules(obs_size, A, hidden, B):
  """values = critic(obs), mean = actor(obs), log_probs = Normal(mean, exp(log_std)).log_prob(action)
  (ppo_controller.py:32-51's forward passes) from the tile pipeline against the torch modules on the CPU."""
  ctl = _make(obs_size, A, hidden, 2 * B + A)
  assert ctl._tiles is not None
  ref = copy.deepcopy(ctl.ac_net).cpu()
  g = torch.Generator().manual_seed(B)
  obs, act = torch.randn(B, obs_size, generator=g), torch.randn(B, A, generator=g)
  with torch.no_grad():
    ev, em = ref.critic(obs).view(-1), ref.actor(obs)
    el = torch.distributions.Normal(em, ref.log_std.exp().expand_as(em)).log_prob(act)
  v, m, l = ctl._tiles.forward_tiles(obs.to(DEV), act.to(DEV), want_mean=True)
  torch.cuda.synchronize()
  torch.testing.assert_close(v.cpu(), ev, rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(m.cpu(), em, rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(l.cpu(), el, rtol=2e-5, atol=2e-5)
  v2 = ctl._tiles.forward_tiles(obs.to(DEV), critic_only=True)[0]
  assert torch.equal(v2, v)                      # the critic alone: the same values


@pytest.mark.parametrize('kind', ['ppo', 'a2c'])
def test_compute_return_on_wide_nets_uses_the_tile_forward(kind):
  """compute_return_batched for a 17-observation, 6-action net of the reference's default width: the forward passes
  come from the tile pipeline and the results are the layer path's (fused=False) up to f32 summation order."""
  from torchrl_amd.contrib.controllers import A2CController
  T, N, obs_size, A = 40, 48, 17, 6
  mk = (lambda fused: PPOController(obs_size, A, lmbda=0.95, device=DEV, fused=fused)) if kind == 'ppo' else \
      (lambda fused: A2CController(obs_size, A, device=DEV, fused=fused))
  torch.manual_seed(3)
  a, b = mk(None), mk(False)
  b.ac_net.load_state_dict(a.ac_net.state_dict())
  assert a._fused is None and a._tiles is not None and b._tiles is None
  g = torch.Generator(device=DEV).manual_seed(4)
  obs = torch.randn(T, N, obs_size, device=DEV, generator=g)
  reward = torch.randn(T, N, device=DEV, generator=g)
  last_obs = torch.randn(N, obs_size, device=DEV, generator=g)
  last_done = (torch.rand(N, device=DEV, generator=g) < 0.3).float()
  if kind == 'ppo':
    action = torch.randn(T, N, A, device=DEV, generator=g)
    ra, rb = a.compute_return_batched(obs, action, reward, last_obs, last_done), b.compute_return_batched(obs, action, reward, last_obs, last_done)
  else:
    ra, rb = a.compute_return_batched(obs, reward, last_obs, last_done), b.compute_return_batched(obs, reward, last_obs, last_done)
  for x, y in zip(ra, rb):
    torch.testing.assert_close(x.reshape(-1), y.reshape(-1), rtol=2e-5, atol=2e-5)
