"""The multi-rank branch of every controller on the GPU (BASELINE config 5: learner sharded over ranks, one
gradient all-reduce per optimizer step; no reference counterpart -- the reference is single-process).

With a process group present the fused step stops at the gradients (`adam=None` / `update=False`), the flat
gradient buffer is all-reduced (RCCL when the backend is "nccl"), and a separate `trl_adam_step` applies
1/world_size, PPO's `clip_grad_value_` (ppo_controller.py:72-73, AFTER the reduction) and Adam.  Checked here:
  * a 1-rank RCCL group with config.force_gradsync (the call sequence of an N-rank run) gives bit-identical
    losses, clipped gradients and post-Adam weights to the single-rank in-kernel update, on the reference's own
    golden learn() fixtures for all four controllers, and meets the golden values at the usual tolerances;
  * the 1/world_size scale and the clamp are applied in that order after the reduction;
  * graph replay + gradient exchange + a Q-network the fused chains do not take (autograd path) trains on fresh
    gradients every step (stale gradients from step 2 on);
  * two gloo ranks sharing the card, each learning on half a batch, end up with the full-batch step.
"""
import os
import socket

import numpy as np
import pytest
import torch

from torchrl_amd import config
import torch.distributed as dist
import torch.multiprocessing as mp

from torchrl_amd.contrib.controllers import DQNController, DDPGController, A2CController, PPOController
from tests.helpers import FIELDS, state_from, t, sub

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


@pytest.fixture(scope='module')
def rccl_world1():
  """A 1-rank RCCL process group for this module only (torn down afterwards: the other test modules run
  without one)."""
  assert not dist.is_initialized()
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(_free_port())
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
  yield
  dist.destroy_process_group()


def _pair(make, monkeypatch):
  """(solo, split): the same controller built without and with the gradient exchange in its step."""
  monkeypatch.setattr(config, 'force_gradsync', False)
  solo = make()
  monkeypatch.setattr(config, 'force_gradsync', True)
  split = make()
  assert not solo._sync.enabled and split._sync.enabled and split._sync.world_size == 1
  return solo, split


def _same(a, b, what):
  assert torch.equal(a, b), '%s: max abs diff %g' % (what, float((a - b).abs().max()))


def _close(got, exp, rtol, atol_frac):
  exp = np.asarray(exp)
  np.testing.assert_allclose(np.asarray(got), exp, rtol=rtol, atol=atol_frac * max(1e-30, float(np.abs(exp).max())))


def test_ppo_split_step_equals_in_kernel_update_golden(golden, rccl_world1, monkeypatch):
  g = golden('ppo.npz')
  ran = 0
  for ci in range(int(g['n_ppo'])):
    p = 'ppo%d_' % ci
    A = g[p + 'net0.log_std'].shape[1]
    make = lambda: PPOController(3, A, lr=3e-4, alpha=0.5, beta=float(g[p + 'beta']), clip_ratio=0.2,
                                 max_grad_norm=float(g[p + 'max_grad_norm']), device=DEV)
    solo, split = _pair(make, monkeypatch)
    for c in (solo, split):
      c.ac_net.load_state_dict(state_from(g, p + 'net0'))
      if c._fused is not None:
        c._fused.MIN_BATCH = 1          # the fixtures' 5..64-row batches through the whole-step kernel
    for step in range(2):
      names = FIELDS + ('returns', 'old_log_probs', 'advantages')
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in names]
      i1, i2 = solo.learn(*batch), split.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        _same(i1[k], i2[k], k)
        np.testing.assert_allclose(float(i2[k].cpu()), float(g['%sstep%d_%s' % (p, step, k)]), rtol=1e-6, atol=2e-7)
      _same(solo._flat.grad, split._flat.grad, 'clipped gradient')        # clip_grad_value_ writes back in place
      _same(solo._flat.data, split._flat.data, 'weights after Adam')
      _same(solo.ac_net_optim.exp_avg_sq, split.ac_net_optim.exp_avg_sq, 'second moment')
      assert solo.ac_net_optim.step_count == split.ac_net_optim.step_count == step + 1
      _close(split._flat.dense(grad=True).cpu().numpy(), g['%sstep%d_grad' % (p, step)], 1e-4, 1e-5)
      _close(split._flat.dense().cpu().numpy(), g['%sstep%d_net' % (p, step)], 1e-5, 1e-5)
    ran += 1
  assert ran >= 3


def test_a2c_split_step_equals_in_kernel_update_golden(golden, rccl_world1, monkeypatch):
  g = golden('a2c.npz')
  for ci in range(int(g['n_a2c'])):
    p = 'a2c%d_' % ci
    A = g[p + 'net0.actor.2.bias'].shape[0]
    make = lambda: A2CController(4, A, lr=1e-3, alpha=0.5, beta=float(g[p + 'beta']), device=DEV)
    solo, split = _pair(make, monkeypatch)
    for c in (solo, split):
      c.ac_net.load_state_dict(state_from(g, p + 'net0'))
      if c._fused is not None:
        c._fused.MIN_BATCH = 1
    for step in range(2):
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in FIELDS + ('returns',)]
      i1, i2 = solo.learn(*batch), split.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        _same(i1[k], i2[k], k)
        np.testing.assert_allclose(float(i2[k].cpu()), float(g['%sstep%d_%s' % (p, step, k)]), rtol=1e-6, atol=2e-7)
      _same(solo._flat.grad, split._flat.grad, 'gradient')
      _same(solo._flat.data, split._flat.data, 'weights after Adam')
      _close(split._flat.dense(grad=True).cpu().numpy(), g['%sstep%d_grad' % (p, step)], 1e-4, 1e-5)
      _close(split._flat.dense().cpu().numpy(), g['%sstep%d_net' % (p, step)], 1e-5, 1e-5)


def test_dqn_split_step_equals_in_kernel_update_golden(golden, rccl_world1, monkeypatch):
  g = golden('dqn.npz')
  for ci in range(int(g['n_dqn'])):
    p = 'dqn%d_' % ci
    A = g[p + 'q0.net.4.bias'].shape[0]
    make = lambda: DQNController(4, A, double_dqn=bool(g[p + 'double']), gamma=0.99, lr=1e-3, n_update_interval=2,
                                 device=DEV)
    solo, split = _pair(make, monkeypatch)
    for c in (solo, split):
      c.q_net.load_state_dict(state_from(g, p + 'q0'))
      c.target_q_net.load_state_dict(state_from(g, p + 't0'))
    for step in range(3):
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in FIELDS]
      i1, i2 = solo.learn(*batch), split.learn(*batch)
      _same(i1['td_loss'], i2['td_loss'], 'td_loss')
      np.testing.assert_allclose(float(i2['td_loss'].cpu()), float(g['%sstep%d_td_loss' % (p, step)]), rtol=1e-6, atol=0)
      _same(solo._q.grad, split._q.grad, 'gradient')
      _same(solo._q.data, split._q.data, 'weights after Adam')
      _same(solo._target.data, split._target.data, 'target after the hard sync')
      _close(sub(split._q.dense().cpu().numpy()), g['%sstep%d_q' % (p, step)], 1e-5, 2e-6)
      _close(sub(split._target.dense().cpu().numpy()), g['%sstep%d_t' % (p, step)], 1e-5, 2e-6)


def test_ddpg_split_step_equals_in_kernel_update_golden(golden, rccl_world1, monkeypatch):
  g = golden('ddpg.npz')
  for ci in range(int(g['n_ddpg'])):
    p = 'ddpg%d_' % ci
    A = g[p + 'actor0.net.4.bias'].shape[0]
    make = lambda: DDPGController(3, A, -2.0, 2.0, actor_lr=1e-4, critic_lr=1e-3, gamma=0.99, tau=1e-2, device=DEV)
    solo, split = _pair(make, monkeypatch)
    for c in (solo, split):
      c.actor.load_state_dict(state_from(g, p + 'actor0'))
      c.critic.load_state_dict(state_from(g, p + 'critic0'))
      c._target_actor.copy_from(c._actor)
      c._target_critic.copy_from(c._critic)
    for step in range(2):
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in FIELDS]
      i1, i2 = solo.learn(*batch), split.learn(*batch)
      for k in ('actor_loss', 'critic_loss'):
        _same(i1[k], i2[k], k)
      np.testing.assert_allclose(float(i2['critic_loss'].cpu()), float(g['%sstep%d_critic_loss' % (p, step)]), rtol=1e-6, atol=0)
      for nm in ('_actor', '_critic', '_target_actor', '_target_critic'):
        _same(getattr(solo, nm).data, getattr(split, nm).data, nm)
      for nm, f in (('actor', split._actor), ('critic', split._critic), ('target_actor', split._target_actor),
                    ('target_critic', split._target_critic)):
        _close(sub(f.dense().cpu().numpy()), g['%sstep%d_%s' % (p, step, nm)], 1e-5, 1e-5)


@pytest.mark.parametrize('ctl', ['ppo', 'a2c'])
def test_bf16x6_split_step_equals_in_kernel_update(ctl, rccl_world1, monkeypatch):
  """The second product variant (products='bf16x6', csrc/trl_learn.hip: k_ac_learn_pair_bx) under the N-rank call
  sequence: the gradients-only launch + all-reduce + trl_adam_step give the bits of its in-kernel update (what
  `bench.py --gpus N` runs on every rank; the exact variant is covered by the golden tests above)."""
  B = 8 * 32 * 128 + 640
  if ctl == 'ppo':
    make = lambda: PPOController(3, 1, lr=3e-4, beta=1e-2, max_grad_norm=0.05, device=DEV, hidden=(64, 64), products='bf16x6')
  else:
    make = lambda: A2CController(4, 2, lr=1e-3, beta=1e-2, device=DEV, hidden=(64, 64), products='bf16x6')
  torch.manual_seed(11)
  solo, split = _pair(make, monkeypatch)
  split._flat.data.copy_(solo._flat.data)
  assert solo._fused is not None and solo._fused.bx6_takes(B) and split._fused.bx6_takes(B)
  gen = torch.Generator(device=DEV).manual_seed(12)
  for step in range(2):
    ret = torch.randn(B, 1, device=DEV, generator=gen)
    if ctl == 'ppo':
      obs, act = torch.randn(B, 3, device=DEV, generator=gen), torch.randn(B, 1, device=DEV, generator=gen)
      old, adv = torch.randn(B, 1, device=DEV, generator=gen) - 1.0, torch.randn(B, 1, device=DEV, generator=gen)
      batch = (obs, act, None, None, None, ret, old, adv)
    else:
      obs = torch.randn(B, 4, device=DEV, generator=gen)
      act = torch.randint(0, 2, (B, 1), device=DEV, generator=gen).float()
      batch = (obs, act, None, None, None, ret)
    i1, i2 = solo.learn(*batch), split.learn(*batch)
    for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
      _same(i1[k], i2[k], k)
    _same(solo._flat.grad, split._flat.grad, 'gradient')
    _same(solo._flat.data, split._flat.data, 'weights after Adam')
    _same(solo.ac_net_optim.exp_avg_sq, split.ac_net_optim.exp_avg_sq, 'second moment')


def test_ppo_scale_then_clamp_after_the_reduction(rccl_world1, monkeypatch):
  """grad_scale (1 / world_size) and clip_grad_value_ are applied to the REDUCED gradient, scale first: the update
  of a 2-rank run whose ranks hold the same half-batch gradient g is Adam on clamp(0.5 * (g + g))."""
  import copy
  monkeypatch.setattr(config, 'force_gradsync', True)
  torch.manual_seed(2)
  c = PPOController(3, 2, lr=1e-3, beta=1e-3, max_grad_norm=0.01, device=DEV, hidden=(64, 64))
  assert c._sync.enabled and c._fused is not None
  c.ac_net_optim.grad_scale = 0.25            # as if four ranks had contributed
  ref = copy.deepcopy(c.ac_net).cpu()
  B = 512
  gen = torch.Generator(device=DEV).manual_seed(9)
  obs, act = torch.randn(B, 3, device=DEV, generator=gen), torch.randn(B, 2, device=DEV, generator=gen)
  ret, adv = torch.randn(B, 1, device=DEV, generator=gen), torch.randn(B, 1, device=DEV, generator=gen)
  old = torch.randn(B, 2, device=DEV, generator=gen) - 1.0
  # raw gradient of this batch, from the fused step itself (gradients only)
  c._fused(obs, act, ret, old, adv, c.clip_ratio, c.alpha, c.beta)
  raw = c._flat.dense(grad=True).cpu()
  assert float(raw.abs().max()) * 0.25 > 2 * 0.01     # the clamp is active on the scaled gradient
  c.learn(obs, act, None, None, None, ret, old, adv)
  exp_g = (raw * 0.25).clamp(-0.01, 0.01)
  assert torch.equal(c._flat.dense(grad=True).cpu(), exp_g)
  opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
  off = 0
  for prm in ref.parameters():
    prm.grad = exp_g[off:off + prm.numel()].view(prm.shape).clone()
    off += prm.numel()
  opt.step()
  exp_w = torch.cat([prm.detach().reshape(-1) for prm in ref.parameters()])
  torch.testing.assert_close(c._flat.dense().cpu(), exp_w, rtol=1e-6, atol=1e-7)


def test_dqn_graph_replay_with_gradient_exchange_on_the_autograd_path(rccl_world1, monkeypatch):
  """graph=True + gradient exchange + a Q-network outside the fused chains (float32 fan-in > 1024: BLAS + autograd):
  the graph body stops after backward, the all-reduce and Adam run outside it.  Every replay must hand them THIS
  step's gradients (round 1 advice: with detached per-parameter gradients the replays wrote into the capture's own
  tensors and the update re-used the first step's)."""
  monkeypatch.setattr(config, 'force_gradsync', True)
  torch.manual_seed(4)
  eager = DQNController(1040, 4, lr=1e-3, n_update_interval=3, device=DEV, graph=False)
  graph = DQNController(1040, 4, lr=1e-3, n_update_interval=3, device=DEV, graph=True)
  assert eager._fused is None or not eager._fused.usable(torch.zeros(48, 1040, device=DEV))
  assert graph._sync.enabled and not graph._q.fresh_grads
  graph._q.data.copy_(eager._q.data)
  graph._target.data.copy_(eager._target.data)
  gen = torch.Generator(device=DEV).manual_seed(6)
  B = 48                                       # not a multiple of 16 either
  for step in range(5):
    obs, nxt = torch.randn(B, 1040, device=DEV, generator=gen), torch.randn(B, 1040, device=DEV, generator=gen)
    act = torch.randint(0, 4, (B, 1), device=DEV, generator=gen).float()
    rew, done = torch.randn(B, 1, device=DEV, generator=gen), (torch.rand(B, 1, device=DEV, generator=gen) < 0.1).float()
    graph._q.data.copy_(eager._q.data)         # one step deep, as in test_graphed_learn_tracks_eager_over_many_steps
    graph._target.data.copy_(eager._target.data)
    graph.q_net_optim.load_state_dict(eager.q_net_optim.state_dict())
    before = eager._q.data.clone()
    l1 = eager.learn(obs, act, rew, nxt, done)['td_loss']
    l2 = graph.learn(obs, act, rew, nxt, done)['td_loss']
    torch.testing.assert_close(l2, l1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(graph._q.grad, eager._q.grad, rtol=1e-3, atol=1e-5 * float(eager._q.grad.abs().max()))
    moved = (eager._q.data - before).abs()
    diff = (graph._q.data - eager._q.data).abs()
    assert float(moved.max()) > 1e-4
    assert float((diff > 5e-5).float().mean()) < 0.02, step


# ------------------------------------------------------------------ two real ranks (gloo, sharing the card)
def _two_rank_worker(rank, world, port, out):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  config.force_gradsync = False
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    dev = torch.device(DEV)
    torch.manual_seed(50 + rank)                          # ranks start apart; the constructor broadcasts rank 0's weights
    c = PPOController(3, 1, lr=1e-3, beta=1e-3, max_grad_norm=0.5, device=dev, hidden=(64, 64))
    assert c._sync.enabled and c._sync.world_size == world and c._fused is not None
    gen = torch.Generator().manual_seed(77)
    B = 2048
    obs, act = torch.randn(B, 3, generator=gen), torch.randn(B, 1, generator=gen)
    ret, adv, old = torch.randn(B, 1, generator=gen), torch.randn(B, 1, generator=gen), torch.randn(B, 1, generator=gen) - 1
    lo, hi = rank * B // world, (rank + 1) * B // world     # this rank's shard of the minibatch
    w0 = c._flat.data.clone().cpu()
    info = c.learn(*[x[lo:hi].to(dev) if x is not None else None for x in (obs, act, None, None, None, ret, old, adv)])
    torch.cuda.synchronize()
    out[rank] = (w0, c._flat.grad.cpu(), c._flat.data.cpu(), {k: float(v) for k, v in info.items()})
  finally:
    dist.destroy_process_group()


def test_two_gloo_ranks_on_half_batches_equal_the_full_batch_step():
  world, port = 2, _free_port()
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_two_rank_worker, args=(world, port, out), nprocs=world, join=True)
  (w0a, ga, wa, ia), (w0b, gb, wb, ib) = out[0], out[1]
  assert torch.equal(w0a, w0b)                              # broadcast at construction
  assert torch.equal(ga, gb) and torch.equal(wa, wb)        # same reduced gradient, same update on both ranks
  # single process, whole minibatch, same starting weights
  config.force_gradsync = False
  c = PPOController(3, 1, lr=1e-3, beta=1e-3, max_grad_norm=0.5, device=DEV, hidden=(64, 64))
  c._flat.data.copy_(w0a.to(DEV))
  gen = torch.Generator().manual_seed(77)
  B = 2048
  obs, act = torch.randn(B, 3, generator=gen), torch.randn(B, 1, generator=gen)
  ret, adv, old = torch.randn(B, 1, generator=gen), torch.randn(B, 1, generator=gen), torch.randn(B, 1, generator=gen) - 1
  info = c.learn(*[x.to(DEV) if x is not None else None for x in (obs, act, None, None, None, ret, old, adv)])
  # the ranks' loss values are their shards' means: their average is the minibatch's
  for k in info:
    assert abs(0.5 * (ia[k] + ib[k]) - float(info[k])) <= 1e-5 * abs(float(info[k])) + 1e-6
  # mean of the shard gradients == full-batch gradient up to fp32 reassociation; the ranks hold the scaled, clamped one
  full = c._flat.grad.cpu()
  torch.testing.assert_close(ga, full, rtol=1e-4, atol=1e-5 * float(full.abs().max()))
  moved = (c._flat.data.cpu() - w0a).abs()
  diff = (wa - c._flat.data.cpu()).abs()
  assert float(moved.max()) > 1e-4 and float((diff > 5e-5).float().mean()) < 0.02
