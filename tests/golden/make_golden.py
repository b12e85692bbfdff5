#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from /root/reference: torchrl.utils.storage,
torchrl.contrib.models, torchrl.contrib.controllers.  `gym` is not installed in
the image and the reference only uses it as a type annotation
(torchrl/controllers/controller.py:17), so an empty in-memory module object with
the three class names stands in for it.  Nothing here is
reference source text: the outputs are numbers (inputs, weights, expected
outputs) written as .npz files that numpy loads with allow_pickle=False.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

_gym = types.ModuleType('gym')
for _n in ('Space', 'Env', 'Wrapper'):
  setattr(_gym, _n, type(_n, (), {}))
sys.modules['gym'] = _gym
sys.path.insert(0, '/root/reference')

from torchrl.utils.storage import (  # noqa: E402
    TensorTupleDataset, TransitionTupleDataset, Transition)
from torchrl.contrib.controllers import (  # noqa: E402
    DQNController, DDPGController, A2CController, PPOController)

SUB = 16  # stride of the sample kept from big post-step weight tensors


def npy(t):
  return t.detach().cpu().numpy().copy()


def save(name, **arrs):
  path = os.path.join(HERE, name)
  np.savez_compressed(path, **arrs)
  print('%-28s %8.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(arrs)))


def flat_params(module, grad=False):
  vs = [(p.grad if grad else p.data).reshape(-1) for p in module.parameters()]
  return npy(torch.cat(vs))


def state_arrays(prefix, module):
  return {'%s.%s' % (prefix, k): npy(v) for k, v in module.state_dict().items()}


# --------------------------------------------------------------------------- storage
def make_transitions(rs, k, obs_shape, act_dim, obs_dtype):
  out = []
  for _ in range(k):
    if obs_dtype == np.uint8:
      obs = rs.randint(0, 256, size=obs_shape).astype(np.uint8)
      nobs = rs.randint(0, 256, size=obs_shape).astype(np.uint8)
    else:
      obs = rs.randn(*obs_shape).astype(obs_dtype)
      nobs = rs.randn(*obs_shape).astype(obs_dtype)
    if act_dim == 0:   # discrete scalar action, as DQNController.act hands back
      act = np.int64(rs.randint(0, 6))
    else:
      act = rs.uniform(-2, 2, size=(act_dim,)).astype(np.float32)
    rew = float(rs.randn())
    done = bool(rs.rand() < 0.2)
    out.append(Transition(obs, act, rew, nobs, done))
  return out


def gen_storage():
  scenarios = [
      # size, pushes, obs_shape, act_dim, obs dtype
      (5, [1] * 8, (4,), 0, np.float32),
      (500, [10] * 60, (4,), 0, np.float32),
      (-1, [1, 10, 3], (3,), 1, np.float32),
      (5, [12], (4,), 0, np.float64),
      (7, [3, 10, 2, 7, 1], (3,), 2, np.float32),
      (6, [4, 4, 1], (2, 3, 2), 0, np.uint8),
      (64, [64, 1, 63, 64], (4,), 0, np.float32),
  ]
  arrs = {'n_scenarios': np.int64(len(scenarios))}
  rs = np.random.RandomState(1234)
  for si, (size, pushes, oshape, adim, odt) in enumerate(scenarios):
    ds = TransitionTupleDataset(size=size)
    p = 's%d_' % si
    arrs[p + 'size'] = np.int64(size)
    arrs[p + 'pushes'] = np.asarray(pushes, np.int64)
    lens = []
    for pi, k in enumerate(pushes):
      tl = make_transitions(rs, k, oshape, adim, odt)
      for fi, f in enumerate(Transition._fields):
        arrs['%spush%d_%s' % (p, pi, f)] = np.stack([np.asarray(t[fi]) for t in tl])
      ds.extend(tl)
      lens.append(len(ds))
      # logical snapshot after every push (small scenarios only)
      if size != 500 or pi in (0, 49, 50, 59):
        for fi, f in enumerate(Transition._fields):
          arrs['%safter%d_%s' % (p, pi, f)] = npy(ds[:][fi])
    arrs[p + 'lens'] = np.asarray(lens, np.int64)
    n = len(ds)
    idx = torch.from_numpy(rs.randint(0, n, size=min(2 * n, 40)).astype(np.int64))
    arrs[p + 'idx'] = npy(idx)
    for fi, f in enumerate(Transition._fields):
      arrs['%sgather_%s' % (p, f)] = npy(ds[idx][fi])
    for fi, f in enumerate(Transition._fields):
      arrs['%sitem0_%s' % (p, f)] = npy(ds[0][fi])
    ds.truncate()
    assert len(ds) == 0
  # empty extend is a no-op (storage.py:62-63)
  ds = TransitionTupleDataset(size=3)
  ds.extend([])
  arrs['empty_len'] = np.int64(len(ds))
  # generic TensorTupleDataset with dtype promotion (float32 zeros cat float64/int64)
  t = TensorTupleDataset(size=4)
  a = torch.arange(12, dtype=torch.float64).reshape(6, 2)
  b = torch.arange(6, dtype=torch.int64).reshape(6, 1)
  t.extend(a, b)
  arrs['tt_a'] = npy(t[:][0])
  arrs['tt_b'] = npy(t[:][1])
  save('storage.npz', **arrs)


# --------------------------------------------------------------------------- sampler
def gen_sampler():
  arrs = {}
  cases = [(0, [(33, 32), (33, 32), (40, 7)]),
           (1, [(5000, 256), (5000, 256), (5010, 256)]),
           (3, [(1000000, 128), (1000000, 128)]),
           (1234, [(1000000, 4096), (999, 999), (2, 2), (1, 1), (70000, 64)])]
  arrs['n_cases'] = np.int64(len(cases))
  for ci, (seed, calls) in enumerate(cases):
    torch.manual_seed(seed)
    arrs['c%d_seed' % ci] = np.int64(seed)
    arrs['c%d_calls' % ci] = np.asarray(calls, np.int64)
    for j, (n, b) in enumerate(calls):
      arrs['c%d_out%d' % (ci, j)] = npy(torch.randperm(n)[:b])
    arrs['c%d_next_u32' % ci] = npy(torch.randint(0, 2 ** 31, (4,)))
  save('sampler.npz', **arrs)


# --------------------------------------------------------------------------- returns / GAE
def rollout(rs, T, obs_size, act_dim, discrete_actions, done_last):
  obs = torch.from_numpy(rs.randn(T, obs_size).astype(np.float32))
  nobs = torch.from_numpy(rs.randn(T, obs_size).astype(np.float32))
  if discrete_actions:
    act = torch.from_numpy(rs.randint(0, discrete_actions, size=(T, 1)).astype(np.float32))
  else:
    act = torch.from_numpy(rs.uniform(-2, 2, size=(T, act_dim)).astype(np.float32))
  rew = torch.from_numpy(rs.randn(T, 1).astype(np.float32))
  done = torch.from_numpy((rs.rand(T, 1) < 0.05).astype(np.float32))
  done[-1] = float(done_last)
  return [obs, act, rew, nobs, done]


def gen_returns():
  rs = np.random.RandomState(77)
  arrs = {}
  # A2C: lambda = 1.0 default and a non-default
  cases = []
  ci = 0
  for (T, lm, dl) in [(5, 1.0, 0), (5, 1.0, 1), (20, 0.9, 0), (1, 1.0, 0)]:
    torch.manual_seed(100 + ci)
    c = A2CController(4, 2, gamma=0.99, lmbda=lm)
    batch = rollout(rs, T, 4, 0, 2, dl)
    ret = c.compute_return(*batch)
    with torch.no_grad():
      v, _ = c.ac_net(batch[0])
      nv, _ = c.ac_net(batch[3][-1:])
    p = 'a2c%d_' % ci
    arrs.update(state_arrays(p + 'net', c.ac_net))
    for f, b in zip(Transition._fields, batch):
      arrs[p + f] = npy(b)
    arrs[p + 'gamma'] = np.float64(0.99)
    arrs[p + 'lmbda'] = np.float64(lm)
    arrs[p + 'returns'] = npy(ret)
    arrs[p + 'values'] = npy(v)
    arrs[p + 'next_value'] = npy(nv)
    ci += 1
  arrs['n_a2c'] = np.int64(ci)
  ci = 0
  for (T, A, lm, dl) in [(5, 1, 0.95, 0), (20, 1, 0.95, 1), (20, 2, 0.95, 0),
                         (2048, 1, 0.95, 0), (2048, 1, 1.0, 1), (1, 1, 0.95, 0)]:
    torch.manual_seed(200 + ci)
    c = PPOController(3, A, gamma=0.99, lmbda=lm)
    batch = rollout(rs, T, 3, A, 0, dl)
    ret, logp, vals = c.compute_return(*batch)
    with torch.no_grad():
      nv, _ = c.ac_net(batch[3][-1:])
    p = 'ppo%d_' % ci
    arrs.update(state_arrays(p + 'net', c.ac_net))
    for f, b in zip(Transition._fields, batch):
      arrs[p + f] = npy(b)
    arrs[p + 'gamma'] = np.float64(0.99)
    arrs[p + 'lmbda'] = np.float64(lm)
    arrs[p + 'returns'] = npy(ret)
    arrs[p + 'log_probs'] = npy(logp)
    arrs[p + 'values'] = npy(vals)
    arrs[p + 'next_value'] = npy(nv)
    arrs[p + 'adv'] = npy(ret - vals)          # examples/ppo.py:54
    ci += 1
  arrs['n_ppo'] = np.int64(ci)

  # [T, N] multi-env block: N per-env buffers pushed through the reference one by
  # one, exactly as examples/ppo.py:49-59 does, then laid out env-major.
  torch.manual_seed(300)
  T, N = 37, 6
  c = PPOController(3, 1, gamma=0.99, lmbda=0.95)
  arrs.update(state_arrays('blk_net', c.ac_net))
  cols = [[] for _ in range(5)]
  rets, lps, vls, nvs = [], [], [], []
  for e in range(N):
    batch = rollout(rs, T, 3, 1, 0, e % 2)
    r, lp, v = c.compute_return(*batch)
    with torch.no_grad():
      nv, _ = c.ac_net(batch[3][-1:])
    for i, b in enumerate(batch):
      cols[i].append(npy(b))
    rets.append(npy(r)); lps.append(npy(lp)); vls.append(npy(v)); nvs.append(npy(nv))
  for f, cl in zip(Transition._fields, cols):
    arrs['blk_' + f] = np.stack(cl)          # [N, T, .]
  arrs['blk_returns'] = np.stack(rets)
  arrs['blk_log_probs'] = np.stack(lps)
  arrs['blk_values'] = np.stack(vls)
  arrs['blk_next_value'] = np.stack(nvs)
  arrs['blk_gamma'] = np.float64(0.99)
  arrs['blk_lmbda'] = np.float64(0.95)
  save('returns.npz', **arrs)


# --------------------------------------------------------------------------- losses + updates
def sub(a):
  return a if a.size <= 4000 else a[::SUB]


def gen_dqn():
  rs = np.random.RandomState(5)
  arrs = {}
  ci = 0
  for (double, A, B) in [(False, 2, 32), (True, 2, 32), (True, 6, 256), (False, 3, 7)]:
    torch.manual_seed(400 + ci)
    c = DQNController(4, A, double_dqn=double, gamma=0.99, lr=1e-3, n_update_interval=2)
    # de-synchronise the target net so that the double / plain targets differ
    with torch.no_grad():
      for p_ in c.target_q_net.parameters():
        p_.add_(0.05 * torch.randn_like(p_))
    p = 'dqn%d_' % ci
    arrs[p + 'double'] = np.int64(double)
    arrs.update(state_arrays(p + 'q0', c.q_net))
    arrs.update(state_arrays(p + 't0', c.target_q_net))
    for step in range(3):
      obs = torch.from_numpy(rs.randn(B, 4).astype(np.float32))
      nobs = torch.from_numpy(rs.randn(B, 4).astype(np.float32))
      act = torch.from_numpy(rs.randint(0, A, size=(B, 1)).astype(np.float32))
      rew = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
      done = torch.from_numpy((rs.rand(B, 1) < 0.3).astype(np.float32))
      batch = [obs, act, rew, nobs, done]
      for f, b in zip(Transition._fields, batch):
        arrs['%sstep%d_%s' % (p, step, f)] = npy(b)
      info = c.learn(*batch)
      arrs['%sstep%d_td_loss' % (p, step)] = npy(info['td_loss'])
      arrs['%sstep%d_grad' % (p, step)] = sub(flat_params(c.q_net, grad=True))
      arrs['%sstep%d_q' % (p, step)] = sub(flat_params(c.q_net))
      arrs['%sstep%d_t' % (p, step)] = sub(flat_params(c.target_q_net))
    ci += 1
  arrs['n_dqn'] = np.int64(ci)
  save('dqn.npz', **arrs)


def gen_ddpg():
  rs = np.random.RandomState(6)
  arrs = {}
  ci = 0
  for (A, B) in [(1, 16), (2, 128)]:
    torch.manual_seed(500 + ci)
    c = DDPGController(3, A, -2.0, 2.0, actor_lr=1e-4, critic_lr=1e-3, gamma=0.99, tau=1e-2)
    p = 'ddpg%d_' % ci
    arrs.update(state_arrays(p + 'actor0', c.actor))
    arrs.update(state_arrays(p + 'critic0', c.critic))
    for step in range(2):
      obs = torch.from_numpy(rs.uniform(-1, 1, (B, 3)).astype(np.float32))
      nobs = torch.from_numpy(rs.uniform(-1, 1, (B, 3)).astype(np.float32))
      act = torch.from_numpy(rs.uniform(-2, 2, (B, A)).astype(np.float32))
      rew = torch.from_numpy(rs.uniform(-16, 0, (B, 1)).astype(np.float32))
      done = torch.from_numpy((rs.rand(B, 1) < 0.3).astype(np.float32))
      batch = [obs, act, rew, nobs, done]
      for f, b in zip(Transition._fields, batch):
        arrs['%sstep%d_%s' % (p, step, f)] = npy(b)
      info = c.learn(*batch)
      arrs['%sstep%d_actor_loss' % (p, step)] = npy(info['actor_loss'])
      arrs['%sstep%d_critic_loss' % (p, step)] = npy(info['critic_loss'])
      for nm, m in (('actor', c.actor), ('critic', c.critic),
                    ('target_actor', c.target_actor), ('target_critic', c.target_critic)):
        arrs['%sstep%d_%s' % (p, step, nm)] = sub(flat_params(m))
      arrs['%sstep%d_actor_grad' % (p, step)] = sub(flat_params(c.actor, grad=True))
      arrs['%sstep%d_critic_grad' % (p, step)] = sub(flat_params(c.critic, grad=True))
    ci += 1
  arrs['n_ddpg'] = np.int64(ci)
  save('ddpg.npz', **arrs)


def gen_a2c():
  rs = np.random.RandomState(7)
  arrs = {}
  ci = 0
  for (A, B, beta) in [(2, 80, 1.0), (4, 33, 1e-3)]:
    torch.manual_seed(600 + ci)
    c = A2CController(4, A, gamma=0.99, lmbda=1.0, lr=1e-3, alpha=0.5, beta=beta)
    p = 'a2c%d_' % ci
    arrs[p + 'beta'] = np.float64(beta)
    arrs.update(state_arrays(p + 'net0', c.ac_net))
    for step in range(2):
      obs = torch.from_numpy(rs.randn(B, 4).astype(np.float32))
      nobs = torch.from_numpy(rs.randn(B, 4).astype(np.float32))
      act = torch.from_numpy(rs.randint(0, A, size=(B, 1)).astype(np.float32))
      rew = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
      done = torch.from_numpy((rs.rand(B, 1) < 0.1).astype(np.float32))
      ret = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
      batch = [obs, act, rew, nobs, done, ret]
      for f, b in zip(Transition._fields + ('returns',), batch):
        arrs['%sstep%d_%s' % (p, step, f)] = npy(b)
      info = c.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        arrs['%sstep%d_%s' % (p, step, k)] = npy(info[k])
      arrs['%sstep%d_grad' % (p, step)] = flat_params(c.ac_net, grad=True)
      arrs['%sstep%d_net' % (p, step)] = flat_params(c.ac_net)
    ci += 1
  arrs['n_a2c'] = np.int64(ci)
  save('a2c.npz', **arrs)


def gen_ppo():
  rs = np.random.RandomState(8)
  arrs = {}
  ci = 0
  for (A, B, mgn, beta) in [(1, 64, 1.0, 1.0), (2, 64, 0.01, 1e-3), (1, 5, 1.0, 1e-3), (3, 300, 0.5, 1e-3)]:
    torch.manual_seed(700 + ci)
    c = PPOController(3, A, gamma=0.99, lmbda=0.95, lr=3e-4, alpha=0.5, beta=beta,
                      clip_ratio=0.2, max_grad_norm=mgn)
    p = 'ppo%d_' % ci
    arrs[p + 'beta'] = np.float64(beta)
    arrs[p + 'max_grad_norm'] = np.float64(mgn)
    arrs.update(state_arrays(p + 'net0', c.ac_net))
    for step in range(2):
      obs = torch.from_numpy(rs.randn(B, 3).astype(np.float32))
      nobs = torch.from_numpy(rs.randn(B, 3).astype(np.float32))
      with torch.no_grad():
        _, dist = c.ac_net(obs)
        act = dist.sample()
        # old log-probs of a slightly different policy so that some ratios clip
        old_lp = dist.log_prob(act) + torch.from_numpy(
            rs.uniform(-0.5, 0.5, (B, A)).astype(np.float32))
      rew = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
      done = torch.from_numpy((rs.rand(B, 1) < 0.1).astype(np.float32))
      ret = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
      adv = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
      if step == 1:
        adv[0] = 0.0        # tie case surr1 == surr2 == 0
      batch = [obs, act, rew, nobs, done, ret, old_lp, adv]
      names = Transition._fields + ('returns', 'old_log_probs', 'advantages')
      for f, b in zip(names, batch):
        arrs['%sstep%d_%s' % (p, step, f)] = npy(b)
      info = c.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        arrs['%sstep%d_%s' % (p, step, k)] = npy(info[k])
      arrs['%sstep%d_grad' % (p, step)] = flat_params(c.ac_net, grad=True)   # post clip_grad_value_
      arrs['%sstep%d_net' % (p, step)] = flat_params(c.ac_net)
    ci += 1
  arrs['n_ppo'] = np.int64(ci)
  save('ppo.npz', **arrs)


if __name__ == '__main__':
  torch.set_num_threads(1)
  gen_storage()
  gen_sampler()
  gen_returns()
  gen_dqn()
  gen_ddpg()
  gen_a2c()
  gen_ppo()
