"""The fused off-policy learner steps (csrc/trl_offpolicy.hip, through the C ABI) vs the oracle's
torch-CPU restatement of DDPGController.learn / DQNController.learn (oracle/loss_oracle.py, pinned to
the reference by tests/golden/losses.npz).

Bars: losses 1e-5 relative (north star: 1e-6 for the loss kernels given identical inputs; here the
inputs of the loss are themselves products summed in another order than the CPU's); gradients 1e-4 of
the layer's largest gradient; weights after Adam / Polyak 1e-6 absolute wherever the gradient is not
round-off sized (Adam's first step is -lr * sign(g): an element whose gradient is ~0 may legitimately
step the other way)."""
import copy

import numpy as np
import pytest
import torch

from torchrl_amd import config

from oracle import loss_oracle
from torchrl_amd.contrib.controllers import DDPGController, DQNController

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _batch(rs, B, S, A, discrete=None):
  obs = torch.from_numpy(rs.randn(B, S).astype(np.float32))
  nxt = torch.from_numpy(rs.randn(B, S).astype(np.float32))
  if discrete:
    act = torch.from_numpy(rs.randint(0, discrete, size=(B, 1)).astype(np.float32))
  else:
    act = torch.from_numpy(rs.uniform(-1, 1, size=(B, A)).astype(np.float32))
  rew = torch.from_numpy(rs.uniform(-2, 0, size=(B, 1)).astype(np.float32))
  done = torch.from_numpy((rs.rand(B, 1) < 0.1).astype(np.float32))
  return obs, act, rew, nxt, done


def _perturb(module, rs, scale):
  """Move every parameter off its init (DDPG's last layers start at +-3e-3: gradients through them
  would be tiny and the test would be weak)."""
  with torch.no_grad():
    for p in module.parameters():
      p.add_(torch.from_numpy((rs.randn(*p.shape) * scale).astype(np.float32)).to(p.device))


def _grads(module):
  return [p.grad.detach().cpu().clone() for p in module.parameters()]


def _close_params(got_module, exp_module, exp_grads, what, atol=1e-6):
  for (name, g), e, gr in zip(got_module.named_parameters(), exp_module.parameters(), exp_grads):
    solid = gr.abs() > 1e-4 * gr.abs().max()
    d = (g.detach().cpu() - e.detach()).abs()
    assert float(d[solid].max()) <= atol, (what, name, float(d[solid].max()))
    assert float(d.max()) <= 2.5e-3, (what, name, float(d.max()))       # nowhere more than two steps apart


def _close_grads(got, exp, what):
  for i, (g, e) in enumerate(zip(got, exp)):
    scale = float(e.abs().max()) + 1e-12
    assert float((g - e).abs().max()) <= 1e-4 * scale + 1e-9, (what, i, float((g - e).abs().max()), scale)


DDPG_SHAPES = [(3, 1, 256, 128), (3, 1, 256, 32), (5, 2, (64, 48), 32), (17, 6, (128, 256), 256), (4, 3, 32, 16)]


@pytest.mark.parametrize('S,A,hidden,B', DDPG_SHAPES)
@pytest.mark.parametrize('update', [False, True])
def test_ddpg_fused_step_matches_oracle(S, A, hidden, B, update):
  rs = np.random.RandomState(S * 100 + B)
  torch.manual_seed(S + B)
  ctl = DDPGController(S, A, -1.0, 1.0, actor_lr=1e-3, critic_lr=1e-3, tau=0.05, device=DEV, hidden=hidden)
  assert ctl._fused is not None
  for net in (ctl.actor, ctl.critic, ctl.target_actor, ctl.target_critic):
    _perturb(net, rs, 0.05)
  ref = [copy.deepcopy(n).cpu() for n in (ctl.actor, ctl.critic, ctl.target_actor, ctl.target_critic)]
  opt_a = torch.optim.Adam(ref[0].parameters(), lr=1e-3)
  opt_c = torch.optim.Adam(ref[1].parameters(), lr=1e-3)
  batch = _batch(rs, B, S, A)
  exp = loss_oracle.ddpg_learn(ref[0], ref[1], ref[2], ref[3], opt_a, opt_c, *batch, gamma=0.99, tau=0.05)
  exp_ga, exp_gc = _grads(ref[0]), _grads(ref[1])

  dbatch = [t.to(DEV) for t in batch]
  if update:
    got = ctl.learn(*dbatch)
  else:
    losses = ctl._fused(*dbatch, ctl.gamma, ctl.tau, ctl.actor_optim, ctl.critic_optim, update=False)
    got = dict(actor_loss=losses[0], critic_loss=losses[1])
  torch.cuda.synchronize()
  for k in exp:
    e, v = float(exp[k]), float(got[k])
    assert abs(v - e) <= 1e-5 * abs(e) + 1e-7, (k, v, e)
  _close_grads(_grads(ctl.actor), exp_ga, 'actor grads')
  _close_grads(_grads(ctl.critic), exp_gc, 'critic grads')
  if update:
    assert ctl.actor_optim.step_count == 1 and ctl.critic_optim.step_count == 1
    _close_params(ctl.actor, ref[0], exp_ga, 'actor')
    _close_params(ctl.critic, ref[1], exp_gc, 'critic')
    _close_params(ctl.target_actor, ref[2], exp_ga, 'target actor')
    _close_params(ctl.target_critic, ref[3], exp_gc, 'target critic')
  else:
    assert ctl.actor_optim.step_count == 0


def test_ddpg_fused_vs_autograd_path_three_steps(monkeypatch):
  """Same controller twice, one with the fused step and one on the autograd path: losses of three
  consecutive steps (the later ones see the earlier updates and Polyak averages)."""
  rs = np.random.RandomState(5)
  torch.manual_seed(5)
  fused = DDPGController(3, 1, -2.0, 2.0, device=DEV)
  monkeypatch.setattr(config, 'fused_learn', False)
  plain = DDPGController(3, 1, -2.0, 2.0, device=DEV)
  assert fused._fused is not None and plain._fused is None
  for a, b in ((fused._actor, plain._actor), (fused._critic, plain._critic),
               (fused._target_actor, plain._target_actor), (fused._target_critic, plain._target_critic)):
    b.data.copy_(a.data)
  for step in range(3):
    batch = [t.to(DEV) for t in _batch(rs, 128, 3, 1)]
    lf, lp = fused.learn(*batch), plain.learn(*batch)
    for k in lf:
      assert abs(float(lf[k]) - float(lp[k])) <= 2e-4 * abs(float(lp[k])) + 1e-6, (step, k, float(lf[k]), float(lp[k]))


def test_ddpg_fused_huber_and_reproducible():
  rs = np.random.RandomState(11)
  batch = [t.to(DEV) for t in _batch(rs, 64, 3, 1)]
  outs = []
  for _ in range(2):
    torch.manual_seed(3)
    ctl = DDPGController(3, 1, -2.0, 2.0, device=DEV, hidden=64, huber_delta=0.5)
    info = ctl.learn(*batch)
    torch.cuda.synchronize()
    outs.append((float(info['critic_loss']), ctl._actor.data.clone(), ctl._target_critic.data.clone()))
  assert outs[0][0] == outs[1][0]
  assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
  # Huber critic loss against its definition on the CPU
  torch.manual_seed(3)
  ctl = DDPGController(3, 1, -2.0, 2.0, device=DEV, hidden=64, huber_delta=0.5)
  cpu = [copy.deepcopy(n).cpu() for n in (ctl.critic, ctl.target_critic, ctl.target_actor)]
  o, a, r, n, d = [t.cpu() for t in batch]
  with torch.no_grad():
    y = r + (1.0 - d) * 0.99 * cpu[1](n, cpu[2](n))
    exp = torch.nn.functional.huber_loss(cpu[0](o, a), y, delta=0.5)
  assert abs(outs[0][0] - float(exp)) <= 1e-5 * abs(float(exp)) + 1e-7


@pytest.mark.parametrize('S,A,hidden,B', [(3, 1, 256, 128), (17, 6, (128, 256), 256), (4, 3, 32, 16)])
def test_ddpg_one_launch_chain_is_the_launch_chain_bit_for_bit(S, A, hidden, B):
  """trl_ddpg_learn_fused as ONE launch (device-wide barriers between the phases, coherent accesses for what a phase
  hands to the next) against one launch per phase (trl_offpolicy_set_chain(0)): five consecutive steps, each seeing
  the earlier updates -- losses, all four networks' weights and both optimizers' moments, bit for bit."""
  from torchrl_amd import _lib
  outs = []
  try:
    for chain in (1, 0):
      _lib.lib().trl_offpolicy_set_chain(chain)
      rs = np.random.RandomState(S + B)
      torch.manual_seed(S * 7 + B)
      ctl = DDPGController(S, A, -1.0, 1.0, actor_lr=1e-3, critic_lr=1e-3, tau=0.05, device=DEV, hidden=hidden)
      for net in (ctl.actor, ctl.critic, ctl.target_actor, ctl.target_critic):
        _perturb(net, rs, 0.05)
      losses = []
      for _ in range(5):
        info = ctl.learn(*[t.to(DEV) for t in _batch(rs, B, S, A)])
        losses.append((float(info['actor_loss']), float(info['critic_loss'])))
      torch.cuda.synchronize()
      outs.append((losses, [t.data.clone() for t in (ctl._actor, ctl._critic, ctl._target_actor, ctl._target_critic)]))
  finally:
    _lib.lib().trl_offpolicy_set_chain(1)
  assert outs[0][0] == outs[1][0]
  for a, b in zip(outs[0][1], outs[1][1]):
    assert torch.equal(a, b)


def test_ddpg_one_launch_chain_state_is_per_stream_and_shared_by_learners():
  """The one-launch form keeps its barrier block and table ring per (device, stream): two learners (two workspaces)
  taking turns on one stream share them, and a learner on a side stream gets its own.  Every learner ends with the
  bits the launch form gives it."""
  from torchrl_amd import _lib
  S, A, hidden, B = 3, 1, 64, 128

  def run(chain, side):
    _lib.lib().trl_offpolicy_set_chain(chain)
    rs = np.random.RandomState(11)
    torch.manual_seed(5)
    ctls = [DDPGController(S, A, -1.0, 1.0, actor_lr=1e-3, critic_lr=1e-3, tau=0.05, device=DEV, hidden=hidden)
            for _ in range(2)]
    for c in ctls:
      for net in (c.actor, c.critic, c.target_actor, c.target_critic):
        _perturb(net, rs, 0.05)
    batches = [[t.to(DEV) for t in _batch(rs, B, S, A)] for _ in range(20)]       # 20 > the ring's 16 slots
    torch.cuda.synchronize()
    st = torch.cuda.Stream() if side else torch.cuda.current_stream()
    with torch.cuda.stream(st):
      for i, b in enumerate(batches):
        ctls[i % 2].learn(*b)
    torch.cuda.synchronize()
    return [t.data.clone() for c in ctls for t in (c._actor, c._critic, c._target_actor, c._target_critic)]

  try:
    ref = run(0, False)
    for side in (False, True):
      got = run(1, side)
      for a, b in zip(ref, got):
        assert torch.equal(a, b), side
  finally:
    _lib.lib().trl_offpolicy_set_chain(1)


def test_ddpg_chain_barrier_timeout_is_reported():
  """The one-launch chain's device-wide barrier needs every workgroup resident; when one is not (here: the diagnostic
  build withholds one), the bounded wait must end in an ERROR at the next call on the stream -- not in silently wrong
  weights -- and the steps after it must be right again.  Runs in a child process on lib/libtrl_hip_tune.so
  (tests/chain_fault_child.py has the assertions)."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  if not os.path.exists(os.path.join(root, 'torchrl_amd', 'lib', 'libtrl_hip_tune.so')):
    pytest.skip('diagnostic build not present')
  env = dict(os.environ, TRL_LIB='tune', TRL_CHAIN_WITHHOLD='3', TRL_CHAIN_SPIN='4000', PYTHONPATH=root)
  r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'chain_fault_child.py')], env=env, capture_output=True,
                     text=True, timeout=110)
  assert r.returncode == 0 and 'chain fault child ok' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


DQN_SHAPES = [(4, 2, 128, 256), (4, 2, 128, 32), (8, 6, (64, 32), 48), (33, 18, (256, 128), 128)]


@pytest.mark.parametrize('S,A,hidden,B', DQN_SHAPES)
@pytest.mark.parametrize('double_dqn', [False, True])
def test_dqn_fused_step_matches_oracle(S, A, hidden, B, double_dqn):
  rs = np.random.RandomState(S + 3 * B + int(double_dqn))
  torch.manual_seed(S * B)
  ctl = DQNController(S, A, double_dqn=double_dqn, lr=1e-3, n_update_interval=1000, device=DEV, hidden=hidden)
  assert ctl._fused is not None
  _perturb(ctl.q_net, rs, 0.05)
  _perturb(ctl.target_q_net, rs, 0.05)
  ref_q, ref_t = copy.deepcopy(ctl.q_net).cpu(), copy.deepcopy(ctl.target_q_net).cpu()
  opt = torch.optim.Adam(ref_q.parameters(), lr=1e-3)
  batch = _batch(rs, B, S, A, discrete=A)
  exp = loss_oracle.dqn_learn(ref_q, ref_t, opt, *batch, gamma=0.99, double_dqn=double_dqn)
  exp_g = _grads(ref_q)
  got = ctl.learn(*[t.to(DEV) for t in batch])
  torch.cuda.synchronize()
  e, v = float(exp['td_loss']), float(got['td_loss'])
  assert abs(v - e) <= 1e-5 * abs(e) + 1e-7, (v, e)
  _close_grads(_grads(ctl.q_net), exp_g, 'q grads')
  _close_params(ctl.q_net, ref_q, exp_g, 'q_net')
  for p, q in zip(ctl.target_q_net.parameters(), ref_t.parameters()):       # untouched before the sync interval
    assert torch.equal(p.detach().cpu(), q.detach())


def test_dqn_hard_sync_still_happens():
  rs = np.random.RandomState(2)
  ctl = DQNController(4, 2, n_update_interval=2, device=DEV)
  batch = [t.to(DEV) for t in _batch(rs, 32, 4, 2, discrete=2)]
  ctl.learn(*batch)
  assert not torch.equal(ctl._q.data, ctl._target.data)
  ctl.learn(*batch)
  assert torch.equal(ctl._q.data, ctl._target.data)


def test_fused_rejects_bad_shapes():
  from torchrl_amd import _lib
  ctl = DDPGController(3, 1, -2.0, 2.0, device=DEV)
  rs = np.random.RandomState(0)
  batch = [t.to(DEV) for t in _batch(rs, 24, 3, 1)]     # B % 16 != 0: the autograd path takes it
  assert not ctl._fused.usable(batch[0])
  info = ctl.learn(*batch)
  assert torch.isfinite(info['critic_loss']).item()
  good = [t.to(DEV) for t in _batch(rs, 32, 3, 1)]
  with pytest.raises(ValueError):
    ctl._fused(good[0], good[1][:16], good[2], good[3], good[4], 0.99, 0.01, ctl.actor_optim, ctl.critic_optim)
  with pytest.raises(_lib.TrlError):      # through the C ABI: a batch the kernels do not take
    ctl._fused(*batch, 0.99, 0.01, ctl.actor_optim, ctl.critic_optim, update=False)
