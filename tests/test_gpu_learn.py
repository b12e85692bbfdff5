"""The whole-step PPO kernel (csrc/trl_learn.hip, C ABI trl_ppo_learn_fused) against the oracle's
torch-CPU restatement of PPOController.learn (oracle/loss_oracle.py: ppo_controller.py:53-80):
losses, every parameter gradient, and the weights after clip_grad_value_ + Adam.  Tolerances:
losses 1e-6 relative (the north star's bar; + a 2e-7 absolute floor for terms near 0), gradients
and weights to fp32 summation-order effects."""
import copy

import numpy as np
import pytest
import torch

from torchrl_amd import config

from oracle import loss_oracle
from torchrl_amd import functional as F
from torchrl_amd.contrib.controllers import PPOController, A2CController

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# obs size, action size, hidden, batch
CASES = [(3, 1, (64, 64), 65536), (3, 1, (64, 64), 4097), (4, 2, (64, 64), 9000), (17, 2, (48, 33), 5000),
         (40, 1, (64, 64), 3000), (64, 2, (64, 64), 2048), (3, 1, (64,), 8192), (5, 4, (64,), 7001),
         (33, 3, (20,), 300), (3, 1, (64, 64), 256),
         # two hidden layers over <= 4 observations at >= 32768 rows: the role-split kernel (k_ac_learn_pair)
         (3, 1, (64, 64), 40001), (4, 2, (64, 64), 33000), (2, 2, (48, 33), 50000), (1, 1, (64, 20), 32768),
         # one hidden layer wider than 64 over <= 4 observations (ppo_controller.py:14 builds 256): hidden blocks
         (3, 1, (256,), 65536), (3, 1, (256,), 4097), (4, 1, (200,), 3000), (4, 2, (128,), 5000), (2, 2, (100,), 900),
         (3, 1, (96,), 1000), (1, 1, (129,), 2048), (4, 2, (256,), 5000), (3, 2, (200,), 999),
         # ... and over 5 .. 9 observations (more k steps, more column groups of the layer-0 gradient)
         (5, 1, (256,), 3000), (6, 2, (128,), 2500), (7, 1, (200,), 4097), (8, 2, (128,), 3000), (7, 2, (256,), 3000), (9, 1, (256,), 2000),
         (9, 2, (100,), 1500), (8, 1, (128,), 40000)]


def _batch(obs_size, A, B, seed):
  g = torch.Generator().manual_seed(seed)
  obs = torch.randn(B, obs_size, generator=g)
  act = torch.randn(B, A, generator=g)
  ret = torch.randn(B, 1, generator=g)
  adv = torch.randn(B, 1, generator=g)
  return obs, act, ret, adv, g


@pytest.mark.parametrize('obs_size,A,hidden,B', CASES)
def test_fused_learn_matches_oracle(obs_size, A, hidden, B):
  torch.manual_seed(obs_size + 10 * A + B)
  ctl = PPOController(obs_size, A, lr=3e-4, alpha=0.5, beta=1e-2, clip_ratio=0.2, max_grad_norm=0.05,
                      device=DEV, hidden=hidden)
  assert ctl._fused is not None
  with torch.no_grad():
    ctl.ac_net.log_std.add_(torch.linspace(-0.3, 0.2, A, device=DEV).view(1, A))
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.Adam(ref.parameters(), lr=3e-4)
  obs, act, ret, adv, g = _batch(obs_size, A, B, 7 + B)
  with torch.no_grad():
    mean = ref.actor(obs)
    old = torch.distributions.Normal(mean, ref.log_std.exp().expand_as(mean)).log_prob(act)
    old = old + 0.3 * torch.randn(B, A, generator=g)        # ratios on both sides of the clip range
  for step in range(1):     # one step deep: Adam turns round-off-sized gradient differences into lr-sized moves
    exp = loss_oracle.ppo_learn(ref, ref_opt, obs, act, None, None, None, ret, old, adv, clip=0.2, alpha=0.5,
                                beta=1e-2, max_grad_norm=0.05)
    exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])      # after clip_grad_value_
    got = ctl.learn(obs.to(DEV), act.to(DEV), None, None, None, ret.to(DEV), old.to(DEV), adv.to(DEV))
    for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
      e, v = float(exp[k]), float(got[k].cpu())
      assert abs(v - e) <= 1e-6 * abs(e) + 2e-7, (k, v, e)
    got_grad = ctl._flat.dense(grad=True).cpu()
    scale = float(exp_grad.abs().max())
    np.testing.assert_allclose(got_grad.numpy(), exp_grad.numpy(), rtol=1e-4, atol=2e-5 * scale)
    exp_w = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    got_w = ctl._flat.dense().cpu()
    # Adam turns a round-off-sized difference of a near-zero gradient into an lr-sized move
    assert float((got_w - exp_w).abs().max()) <= 2.5 * 3e-4
    assert float(((got_w - exp_w).abs() > 1e-5).float().mean()) < 0.02


def test_fused_learn_unclipped_gradients_and_general_path(monkeypatch):
  """Raw gradients (no clamp in the way) of the whole-step kernel vs the layer-by-layer path of the
  same controller class (autograd over the fused MLP kernels + k_ppo_loss)."""
  outs = []
  for mode in ('1', '0'):
    monkeypatch.setattr(config, 'fused_learn', mode == '1')
    torch.manual_seed(11)
    ctl = PPOController(3, 2, lr=1e-3, alpha=0.5, beta=1e-3, max_grad_norm=1e9, device=DEV, hidden=(64, 64))
    assert (ctl._fused is not None) == (mode == '1')
    obs, act, ret, adv, g = _batch(3, 2, 20000, 3)
    obs, act, ret, adv = (t.to(DEV) for t in (obs, act, ret, adv))
    with torch.no_grad():
      _, mean, ls = ctl.ac_net.heads(obs)
      old = F.normal_log_prob(mean, ls, act) + 0.2 * torch.randn(20000, 2, generator=g).to(DEV)
    info = ctl.learn(obs, act, None, None, None, ret, old, adv)
    outs.append((info, ctl._flat.dense(grad=True).clone()))
  for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
    torch.testing.assert_close(outs[0][0][k], outs[1][0][k], rtol=2e-6, atol=2e-7)
  scale = float(outs[1][1].abs().max())
  torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=2e-5 * scale)


def test_fused_learn_is_reproducible_and_rejects_bad_shapes():
  torch.manual_seed(0)
  ctl = PPOController(3, 1, device=DEV, hidden=(64, 64))
  obs, act, ret, adv, _ = _batch(3, 1, 30000, 1)
  obs, act, ret, adv = (t.to(DEV) for t in (obs, act, ret, adv))
  old = torch.zeros(30000, 1, device=DEV)
  res = []
  for _ in range(2):
    losses = ctl._fused(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
    res.append((losses.clone(), ctl._flat.grad.clone()))
  assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])    # fixed-order sums
  with pytest.raises(ValueError):
    ctl._fused(obs, act[:100], ret, old, adv, 0.2, 0.5, 1e-3)
  assert PPOController(3, 1, device=DEV, hidden=(128, 64))._fused is None              # too wide: general path
  assert PPOController(3, 3, device=DEV, hidden=(64, 64))._fused is None               # A > 2 with two hidden layers
  assert PPOController(3, 3, device=DEV, hidden=(256,))._fused is None                 # wide: two actions at most
  assert PPOController(10, 1, device=DEV, hidden=(256,))._fused is None                # wide needs a fan-in <= 9
  assert PPOController(8, 2, device=DEV, hidden=(256,))._fused is None                 # ... <= 7 with two actions above 128 wide
  assert PPOController(3, 1, device=DEV, hidden=(257,))._fused is None


@pytest.mark.parametrize('obs_size,A,hidden,B', [(3, 1, (64, 64), 70001), (4, 2, (64, 64), 4096), (40, 2, (48, 33), 999),
                                                 (5, 4, (64,), 12345), (64, 1, (64, 64), 300),
                                                 (3, 1, (256,), 70001), (4, 2, (128,), 1234), (2, 1, (130,), 333),
                                                 (4, 2, (256,), 4321), (8, 2, (128,), 2000), (6, 2, (130,), 777),
                                                 (9, 1, (256,), 5000)])
def test_fused_forward_matches_torch(obs_size, A, hidden, B):
  """trl_ppo_forward_fused (value / mean / log-prob, outputs only) vs the torch modules."""
  torch.manual_seed(B)
  ctl = PPOController(obs_size, A, device=DEV, hidden=hidden)
  with torch.no_grad():
    ctl.ac_net.log_std.add_(torch.linspace(-0.4, 0.3, A, device=DEV).view(1, A))
  ref = copy.deepcopy(ctl.ac_net).cpu()
  obs, act = torch.randn(B, obs_size), torch.randn(B, A)
  with torch.no_grad():
    ev, em = ref.critic(obs), ref.actor(obs)
    elp = torch.distributions.Normal(em, ref.log_std.exp().expand_as(em)).log_prob(act)
  value, mean, logp = ctl._fused.forward(obs.to(DEV), act.to(DEV))
  torch.testing.assert_close(value.cpu(), ev.view(-1), rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(mean.cpu(), em, rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(logp.cpu(), elp, rtol=1e-5, atol=1e-5)
  v2, m2, l2 = ctl._fused.forward(obs.to(DEV), want_mean=False)
  assert m2 is None and l2 is None and torch.equal(v2, value)


@pytest.mark.parametrize('obs_size,A,hidden,B', [(3, 1, (64, 64), 40001), (3, 1, (64, 64), 5000), (4, 2, (64, 64), 33000),
                                                 (12, 1, (64, 48), 3000), (4, 1, (64,), 2000), (3, 1, (256,), 40001),
                                                 (4, 2, (256,), 3000), (2, 1, (130,), 1000), (4, 2, (128,), 35000),
                                                 (8, 2, (128,), 3000), (7, 2, (256,), 2000), (9, 1, (100,), 1500)])
def test_forward_kernel_and_learner_forward_are_the_same_bits(obs_size, A, hidden, B):
  """The forward-only kernel and the learner kernels evaluate a row to the same bit pattern (PPO's first-epoch ratio is
  exactly 1; A2C's detached advantage uses the learner's own V).  Observable without a debug output: with the forward
  kernel's values as the returns, the learner's critic loss and every critic gradient are exactly zero; with its
  log-probs as the old ones, the ratio is 1 and the surrogate is the plain mean of the advantages, so clipping at a
  ratio range of 1 +- 1e-30 (never active unless a ratio differs from 1) leaves the actor gradients unchanged."""
  torch.manual_seed(B + obs_size)
  ctl = PPOController(obs_size, A, device=DEV, hidden=hidden)
  obs, act, _, adv, _ = _batch(obs_size, A, B, 3)
  obs, act, adv = obs.to(DEV), act.to(DEV), adv.to(DEV)
  value, _, logp = ctl._fused.forward(obs, act)
  ret = value.view(B, 1).clone()
  res = []
  for clip in (0.2, 1e-30):
    ctl._fused(obs, act, ret, logp, adv, clip, 0.5, 1e-3)
    res.append(ctl._flat.grad.clone())
  losses = ctl._fused(obs, act, ret, logp, adv, 0.2, 0.5, 1e-3).cpu()
  assert float(losses[1]) == 0.0                                   # critic loss: (V - V)^2
  off, ncrit = 0, 0
  for name, p_ in ctl.ac_net.named_parameters():
    if name.startswith('critic'):
      assert not res[0][off:off + p_.numel()].any(), name          # critic gradients: exactly zero
      ncrit += 1
    off += p_.numel()
  assert ncrit >= 4
  assert torch.equal(res[0], res[1])                               # no ratio left the range [1 - 1e-30, 1 + 1e-30]


def test_compute_return_fused_equals_general_path(monkeypatch):
  """compute_return_batched through the forward-only kernel == through the general path."""
  outs = []
  g = torch.Generator().manual_seed(9)
  T, N = 64, 40
  obs, act = torch.randn(T, N, 3, generator=g).to(DEV), torch.randn(T, N, 1, generator=g).to(DEV)
  rew, last = torch.randn(T, N, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV)
  last_done = (torch.rand(N, generator=g) < 0.3).float().to(DEV)
  for mode in ('1', '0'):
    monkeypatch.setattr(config, 'fused_learn', mode == '1')
    torch.manual_seed(4)
    ctl = PPOController(3, 1, gamma=0.99, lmbda=0.95, device=DEV, hidden=(64, 64))
    if ctl._fused is not None:
      ctl._fused.MIN_BATCH = 1
    outs.append(ctl.compute_return_batched(obs, act, rew, last, last_done))
  for a, b in zip(outs[0], outs[1]):
    torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('obs_size,A,hidden,B', [(4, 2, 64, 9001), (4, 2, (64, 64), 4096), (4, 2, (64, 64), 40000), (8, 4, 48, 5000),
                                                 (40, 3, 64, 700), (3, 1, (33, 20), 2048), (4, 2, 128, 9001),
                                                 (4, 2, 100, 700), (4, 2, 256, 9001), (4, 2, 180, 1500),
                                                 (8, 2, 128, 3000), (6, 2, 256, 2000), (9, 2, 100, 1000)])
def test_fused_a2c_learn_matches_oracle(obs_size, A, hidden, B):
  """trl_a2c_learn_fused vs the oracle's torch-CPU A2CController.learn (a2c_controller.py:47-69):
  softmax actor + Categorical loss, losses to 1e-6 relative, gradients, weights after Adam."""
  torch.manual_seed(obs_size + 10 * A + B)
  ctl = A2CController(obs_size, A, lr=1e-3, alpha=0.5, beta=1e-2, device=DEV, hidden=hidden)
  assert ctl._fused is not None
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
  g = torch.Generator().manual_seed(B)
  obs = torch.randn(B, obs_size, generator=g)
  act = torch.randint(0, A, (B, 1), generator=g).float()
  ret = torch.randn(B, 1, generator=g)
  exp = loss_oracle.a2c_learn(ref, ref_opt, obs, act, None, None, None, ret, alpha=0.5, beta=1e-2)
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
  got = ctl.learn(obs.to(DEV), act.to(DEV), None, None, None, ret.to(DEV))
  for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
    e, v = float(exp[k]), float(got[k].cpu())
    assert abs(v - e) <= 1e-6 * abs(e) + 2e-7, (k, v, e)
  got_grad = ctl._flat.dense(grad=True).cpu()
  np.testing.assert_allclose(got_grad.numpy(), exp_grad.numpy(), rtol=1e-4, atol=2e-5 * float(exp_grad.abs().max()))
  exp_w = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
  got_w = ctl._flat.dense().cpu()
  assert float((got_w - exp_w).abs().max()) <= 2.5e-3
  assert float(((got_w - exp_w).abs() > 1e-5).float().mean()) < 0.02


def _random_shapes(n, seed):
  rs = np.random.RandomState(seed)
  out = []
  for _ in range(n):
    depth = int(rs.choice([2, 3]))
    hidden = tuple(int(rs.randint(1, 65)) for _ in range(depth - 1))
    A = int(rs.randint(1, 3 if depth == 3 else 5))
    out.append((int(rs.randint(1, 65)), A, hidden, int(rs.randint(256, 3000))))
  return out


@pytest.mark.parametrize('obs_size,A,hidden,B', _random_shapes(14, 2024))
def test_fused_learn_random_shapes(obs_size, A, hidden, B):
  """Seeded random widths / depths / batch sizes (padding, ragged tails, every template variant):
  gradients of the whole-step kernel vs torch-CPU autograd of the same loss, no clamp in the way."""
  torch.manual_seed(B + obs_size)
  ctl = PPOController(obs_size, A, lr=1e-3, alpha=0.5, beta=1e-2, max_grad_norm=1e9, device=DEV, hidden=hidden)
  assert ctl._fused is not None
  ref = copy.deepcopy(ctl.ac_net).cpu()
  obs, act, ret, adv, g = _batch(obs_size, A, B, B)
  with torch.no_grad():
    mean = ref.actor(obs)
    old = torch.distributions.Normal(mean, ref.log_std.exp().expand_as(mean)).log_prob(act)
    old = old + 0.3 * torch.randn(B, A, generator=g)
  value, mean = ref.critic(obs), ref.actor(obs)
  actor, critic, ent = loss_oracle.ppo_losses(value, mean, ref.log_std, act, old, adv, ret, 0.2)
  (actor + 0.5 * critic - 1e-2 * ent).backward()
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
  losses = ctl._fused(obs.to(DEV), act.to(DEV), ret.to(DEV), old.to(DEV), adv.to(DEV), 0.2, 0.5, 1e-2).cpu()
  for e, v in zip((actor, critic, ent), losses[:3]):
    assert abs(float(v) - float(e.detach())) <= 2e-6 * abs(float(e.detach())) + 3e-7, (float(v), float(e.detach()))
  got_grad = ctl._flat.dense(grad=True).cpu()
  np.testing.assert_allclose(got_grad.numpy(), exp_grad.numpy(), rtol=2e-4, atol=3e-5 * float(exp_grad.abs().max()))
  # and the forward-only kernel on the same shapes
  v2, m2, _ = ctl._fused.forward(obs.to(DEV))
  torch.testing.assert_close(v2.cpu(), value.detach().view(-1), rtol=1e-5, atol=3e-6)
  torch.testing.assert_close(m2.cpu(), mean.detach(), rtol=1e-5, atol=3e-6)


# ------------------------------------------------------------------ the reference's own outputs through the whole-step kernel
def _close(got, exp, rtol, atol_frac):
  exp = np.asarray(exp)
  np.testing.assert_allclose(np.asarray(got), exp, rtol=rtol, atol=atol_frac * max(1e-30, float(np.abs(exp).max())))


def test_fused_ppo_learn_golden_reference_shape(golden):
  """tests/golden/ppo.npz holds the reference's PPOController.learn outputs (two steps: losses, clipped
  gradients, weights after Adam) on its own network, ActorCriticNet(3, A, 256) (ppo_controller.py:14).
  Here they go through the whole-step kernel (its batch threshold, a speed choice, lowered for the
  fixtures' 5..64-row batches); the A = 3 fixture is outside the kernel's shapes and stays with
  tests/test_gpu_losses.py."""
  g = golden('ppo.npz')
  from tests.helpers import FIELDS, state_from, t
  ran = 0
  for ci in range(int(g['n_ppo'])):
    p = 'ppo%d_' % ci
    A = g[p + 'net0.log_std'].shape[1]
    c = PPOController(3, A, lr=3e-4, alpha=0.5, beta=float(g[p + 'beta']), clip_ratio=0.2,
                      max_grad_norm=float(g[p + 'max_grad_norm']), device=DEV)
    if A > 2:
      assert c._fused is None
      continue
    assert c._fused is not None
    c._fused.MIN_BATCH = 1
    c.ac_net.load_state_dict(state_from(g, p + 'net0'))
    for step in range(2):
      names = FIELDS + ('returns', 'old_log_probs', 'advantages')
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in names]
      assert c._fused.usable(batch[0])
      info = c.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        np.testing.assert_allclose(float(info[k].cpu()), float(g['%sstep%d_%s' % (p, step, k)]), rtol=1e-6, atol=2e-7)
      _close(c._flat.dense(grad=True).cpu().numpy(), g['%sstep%d_grad' % (p, step)], 1e-4, 1e-5)
      _close(c._flat.dense().cpu().numpy(), g['%sstep%d_net' % (p, step)], 1e-5, 1e-5)
    ran += 1
  assert ran >= 3


def test_fused_a2c_learn_golden_reference_shape(golden):
  """tests/golden/a2c.npz: the reference's A2CController.learn on A2CNet(4, 2, 256) (a2c_controller.py:13),
  through the whole-step kernel's two-output 256-wide variant."""
  g = golden('a2c.npz')
  from tests.helpers import FIELDS, state_from, t
  ran = 0
  for ci in range(int(g['n_a2c'])):
    p = 'a2c%d_' % ci
    A = g[p + 'net0.actor.2.bias'].shape[0]
    c = A2CController(4, A, lr=1e-3, alpha=0.5, beta=float(g[p + 'beta']), device=DEV)
    if A > 2:
      assert c._fused is None
      continue
    assert c._fused is not None
    c._fused.MIN_BATCH = 1
    c.ac_net.load_state_dict(state_from(g, p + 'net0'))
    for step in range(2):
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in FIELDS + ('returns',)]
      assert c._fused.usable(batch[0])
      info = c.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        np.testing.assert_allclose(float(info[k].cpu()), float(g['%sstep%d_%s' % (p, step, k)]), rtol=1e-6, atol=2e-7)
      _close(c._flat.dense(grad=True).cpu().numpy(), g['%sstep%d_grad' % (p, step)], 1e-4, 1e-5)
      _close(c._flat.dense().cpu().numpy(), g['%sstep%d_net' % (p, step)], 1e-5, 1e-5)
    ran += 1
  assert ran >= 1


@pytest.mark.parametrize('A,hidden,B', [(3, 256, 777), (4, 130, 4100)])
def test_wide_forward_more_actions_and_learn_rejects(A, hidden, B):
  """The C ABI's forward takes a wide hidden layer with up to 4 actions (the Python classes only route
  A <= 2 there, where the learn entry follows); the learn entry refuses A > 2 on a wide layer loudly."""
  torch.manual_seed(A + hidden)
  ctl = PPOController(4, A, device=DEV, hidden=(hidden,))
  assert ctl._fused is None
  lins = [[m for m in stack if isinstance(m, torch.nn.Linear)] for stack in (ctl.ac_net.critic, ctl.ac_net.actor)]
  fused = F.FusedPpoLearn(lins[0], lins[1], ctl.ac_net.log_std)
  ref = copy.deepcopy(ctl.ac_net).cpu()
  obs, act = torch.randn(B, 4), torch.randn(B, A)
  with torch.no_grad():
    ev, em = ref.critic(obs), ref.actor(obs)
    elp = torch.distributions.Normal(em, ref.log_std.exp().expand_as(em)).log_prob(act)
  value, mean, logp = fused.forward(obs.to(DEV), act.to(DEV))
  torch.testing.assert_close(value.cpu(), ev.view(-1), rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(mean.cpu(), em, rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(logp.cpu(), elp, rtol=1e-5, atol=1e-5)
  z = torch.zeros(B, A, device=DEV)
  with pytest.raises(Exception, match='action size'):
    fused(obs.to(DEV), act.to(DEV), torch.zeros(B, device=DEV), z, torch.zeros(B, device=DEV), 0.2, 0.5, 1e-3)
