"""Fused loss kernels, their backward seeds and the fused update vs the reference.

Bars (north star): losses within 1e-6 relative of the reference.  Gradients and updated
weights are compared at 1e-5 relative with a small absolute floor: they sum signed terms,
and the GPU matmuls in front of the kernels already differ from the CPU's in summation
order by a few ulp.  A tiny absolute floor is also applied to losses that the fixtures put
near zero by cancellation (the entropy term)."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle
from torchrl_amd import functional as F
from torchrl_amd.contrib import models
from torchrl_amd.contrib.controllers import DQNController, DDPGController, A2CController, PPOController
from tests.helpers import FIELDS, state_from, t, sub

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LOSS_RTOL = 1e-6


def _loss_close(got, exp, atol=2e-7):
  np.testing.assert_allclose(float(got), float(exp), rtol=LOSS_RTOL, atol=atol)


def _vec_close(got, exp, rtol=1e-5, atol_frac=1e-6):
  exp = np.asarray(exp)
  atol = atol_frac * max(1e-30, float(np.abs(exp).max()))
  np.testing.assert_allclose(np.asarray(got), exp, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ kernels vs torch-CPU oracle
@pytest.mark.parametrize('B,A,double', [(1, 2, False), (32, 2, True), (256, 6, True), (257, 18, False),
                                        (5000, 4, True), (70000, 3, False)])
def test_td_loss_kernel(B, A, double):
  rs = np.random.RandomState(B + A)
  q = torch.from_numpy(rs.randn(B, A).astype(np.float32)).requires_grad_()
  qnt = torch.from_numpy(rs.randn(B, A).astype(np.float32))
  qno = torch.from_numpy(rs.randn(B, A).astype(np.float32))
  qno[::7, 1] = qno[::7, 0] = 3.0      # ties: first maximum wins, as torch's CPU max does
  act = torch.from_numpy(rs.randint(0, A, (B, 1)).astype(np.float32))
  rew = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
  done = torch.from_numpy((rs.rand(B, 1) < 0.3).astype(np.float32))
  # reference arithmetic (dqn_controller.py:64-78) on the CPU
  cur = q.gather(1, act.long())
  if double:
    nq = qnt.gather(1, qno.max(1, keepdim=True)[1])
  else:
    nq = qnt.max(1)[0].unsqueeze(1)
  exp = ((rew + 0.99 * nq * (1.0 - done)) - cur).pow(2).mean()
  exp.backward()
  qd = q.detach().to(DEV).requires_grad_()
  got = F.td_loss(qd, qnt.to(DEV), act.to(DEV), rew.to(DEV), done.to(DEV), 0.99,
                  q_next_online=qno.to(DEV) if double else None)
  got.backward()
  _loss_close(got, exp)
  _vec_close(qd.grad.cpu().numpy(), q.grad.numpy())


def test_td_huber_variant():
  rs = np.random.RandomState(1)
  B, A = 300, 4
  q = torch.from_numpy(rs.randn(B, A).astype(np.float32) * 3).requires_grad_()
  qnt = torch.from_numpy(rs.randn(B, A).astype(np.float32) * 3)
  act = torch.from_numpy(rs.randint(0, A, (B, 1)).astype(np.float32))
  rew = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
  done = torch.from_numpy((rs.rand(B, 1) < 0.3).astype(np.float32))
  y = rew + 0.99 * qnt.max(1)[0].unsqueeze(1) * (1.0 - done)
  exp = torch.nn.functional.huber_loss(q.gather(1, act.long()), y, delta=1.0)
  exp.backward()
  qd = q.detach().to(DEV).requires_grad_()
  got = F.td_loss(qd, qnt.to(DEV), act.to(DEV), rew.to(DEV), done.to(DEV), 0.99, huber_delta=1.0)
  got.backward()
  _loss_close(got, exp)
  _vec_close(qd.grad.cpu().numpy(), q.grad.numpy())


@pytest.mark.parametrize('B', [1, 16, 128, 4096, 100000])
def test_ddpg_critic_kernel(B):
  rs = np.random.RandomState(B)
  cur = torch.from_numpy(rs.randn(B, 1).astype(np.float32)).requires_grad_()
  nxt = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
  rew = torch.from_numpy(rs.uniform(-16, 0, (B, 1)).astype(np.float32))
  done = torch.from_numpy((rs.rand(B, 1) < 0.3).astype(np.float32))
  exp = torch.nn.functional.mse_loss(cur, rew + (1.0 - done) * 0.99 * nxt)
  exp.backward()
  cd = cur.detach().to(DEV).requires_grad_()
  got = F.ddpg_critic_loss(cd, nxt.to(DEV), rew.to(DEV), done.to(DEV), 0.99)
  got.backward()
  _loss_close(got, exp)
  _vec_close(cd.grad.cpu().numpy(), cur.grad.numpy())


@pytest.mark.parametrize('B,A', [(1, 1), (5, 1), (64, 1), (64, 2), (300, 3), (4096, 6), (70000, 1), (66000, 17)])
def test_ppo_loss_kernel(B, A):
  rs = np.random.RandomState(B * 7 + A)
  mean = torch.from_numpy(rs.randn(B, A).astype(np.float32)).requires_grad_()
  log_std = torch.from_numpy(rs.uniform(-1, 0.5, (1, A)).astype(np.float32)).requires_grad_()
  value = torch.from_numpy(rs.randn(B, 1).astype(np.float32)).requires_grad_()
  action = (mean.detach() + torch.from_numpy(rs.randn(B, A).astype(np.float32)) * log_std.detach().exp())
  with torch.no_grad():
    old = torch.distributions.Normal(mean, log_std.exp().expand_as(mean)).log_prob(action)
    old = old + torch.from_numpy(rs.uniform(-0.5, 0.5, (B, A)).astype(np.float32))
  adv = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
  adv[0] = 0.0
  ret = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
  alpha, beta, clip = 0.5, 1e-3, 0.2
  a, c, e = loss_oracle.ppo_losses(value, mean, log_std, action, old, adv, ret, clip)
  (a + alpha * c - beta * e).backward()
  md, ld, vd = (x.detach().to(DEV).requires_grad_() for x in (mean, log_std, value))
  loss, parts = F.ppo_loss(md, ld, vd, action.to(DEV), old.to(DEV), adv.to(DEV), ret.to(DEV), clip, alpha, beta)
  loss.backward()
  _loss_close(parts[0], a)
  _loss_close(parts[1], c)
  _loss_close(parts[2], e)
  _loss_close(loss, a + alpha * c - beta * e)
  _vec_close(md.grad.cpu().numpy(), mean.grad.numpy())
  _vec_close(vd.grad.cpu().numpy(), value.grad.numpy())
  _vec_close(ld.grad.cpu().numpy(), log_std.grad.numpy(), rtol=1e-4, atol_frac=1e-4)
  lp = F.normal_log_prob(md.detach(), ld.detach(), action.to(DEV))
  with torch.no_grad():
    exp_lp = torch.distributions.Normal(mean, log_std.exp().expand_as(mean)).log_prob(action)
  np.testing.assert_allclose(lp.cpu().numpy(), exp_lp.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('B,A', [(1, 2), (33, 4), (80, 2), (5000, 6), (70000, 18)])
def test_a2c_loss_kernel(B, A):
  rs = np.random.RandomState(B * 3 + A)
  logits = torch.from_numpy(rs.randn(B, A).astype(np.float32) * 2)
  probs = torch.softmax(logits, 1).requires_grad_()
  value = torch.from_numpy(rs.randn(B, 1).astype(np.float32)).requires_grad_()
  act = torch.from_numpy(rs.randint(0, A, (B, 1)).astype(np.float32))
  ret = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
  alpha, beta = 0.5, 1.0
  a, c, e = loss_oracle.a2c_losses(value, probs, act, ret)
  (a + alpha * c - beta * e).backward()
  pd, vd = probs.detach().to(DEV).requires_grad_(), value.detach().to(DEV).requires_grad_()
  loss, parts = F.a2c_loss(pd, vd, act.to(DEV), ret.to(DEV), alpha, beta)
  loss.backward()
  _loss_close(parts[0], a)
  _loss_close(parts[1], c)
  _loss_close(parts[2], e)
  _vec_close(pd.grad.cpu().numpy(), probs.grad.numpy())
  _vec_close(vd.grad.cpu().numpy(), value.grad.numpy())


@pytest.mark.parametrize('n', [1, 1000, 67073, 1 << 20])
def test_adam_and_polyak_kernels(n):
  rs = np.random.RandomState(n % 1000)
  p0 = torch.from_numpy(rs.randn(n).astype(np.float32))
  ref = p0.clone().requires_grad_()
  opt = torch.optim.Adam([ref], lr=1e-3)
  p = p0.clone().to(DEV)
  g_buf = torch.zeros(n, device=DEV)
  m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
  for step in range(1, 4):
    g = torch.from_numpy(rs.randn(n).astype(np.float32) * (3.0 if step == 2 else 0.01))
    ref.grad = g.clone()
    torch.nn.utils.clip_grad_value_([ref], 0.5)
    opt.step()
    g_buf.copy_(g * 2.0)                       # scale 0.5 models the 1/world_size of 2 ranks
    F.adam_step_(p, g_buf, m, v, step, 1e-3, grad_scale=0.5, clip_value=0.5)
    np.testing.assert_allclose(g_buf.cpu().numpy(), ref.grad.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)
  src = torch.from_numpy(rs.randn(n).astype(np.float32))
  exp = 1e-2 * src + (1.0 - 1e-2) * p.cpu()
  F.polyak_(p, src.to(DEV), 1e-2)
  assert torch.equal(p.cpu(), exp)


# ------------------------------------------------------------------ controllers vs golden learn()
@pytest.mark.parametrize('graph', [False, True])
def test_dqn_controller_learn_golden(golden, graph):
  g = golden('dqn.npz')
  for ci in range(int(g['n_dqn'])):
    p = 'dqn%d_' % ci
    A = g[p + 'q0.net.4.bias'].shape[0]
    c = DQNController(4, A, double_dqn=bool(g[p + 'double']), gamma=0.99, lr=1e-3, n_update_interval=2, device=DEV, graph=graph)
    c.q_net.load_state_dict(state_from(g, p + 'q0'))
    c.target_q_net.load_state_dict(state_from(g, p + 't0'))
    for step in range(3):
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in FIELDS]
      info = c.learn(*batch)
      _loss_close(info['td_loss'].cpu(), g['%sstep%d_td_loss' % (p, step)], atol=0)
      _vec_close(sub(c._q.dense(grad=True).cpu().numpy()), g['%sstep%d_grad' % (p, step)])
      _vec_close(sub(c._q.dense().cpu().numpy()), g['%sstep%d_q' % (p, step)], atol_frac=2e-6)
      _vec_close(sub(c._target.dense().cpu().numpy()), g['%sstep%d_t' % (p, step)], atol_frac=2e-6)


@pytest.mark.parametrize('graph', [False, True])
def test_ddpg_controller_learn_golden(golden, graph):
  g = golden('ddpg.npz')
  for ci in range(int(g['n_ddpg'])):
    p = 'ddpg%d_' % ci
    A = g[p + 'actor0.net.4.bias'].shape[0]
    c = DDPGController(3, A, -2.0, 2.0, actor_lr=1e-4, critic_lr=1e-3, gamma=0.99, tau=1e-2, device=DEV, graph=graph)
    c.actor.load_state_dict(state_from(g, p + 'actor0'))
    c.critic.load_state_dict(state_from(g, p + 'critic0'))
    c._target_actor.copy_from(c._actor)
    c._target_critic.copy_from(c._critic)
    for step in range(2):
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in FIELDS]
      info = c.learn(*batch)
      _loss_close(info['actor_loss'].cpu(), g['%sstep%d_actor_loss' % (p, step)], atol=1e-7)
      _loss_close(info['critic_loss'].cpu(), g['%sstep%d_critic_loss' % (p, step)], atol=0)
      _vec_close(sub(c._actor.dense(grad=True).cpu().numpy()), g['%sstep%d_actor_grad' % (p, step)], rtol=1e-4, atol_frac=1e-5)
      _vec_close(sub(c._critic.dense(grad=True).cpu().numpy()), g['%sstep%d_critic_grad' % (p, step)], rtol=1e-4, atol_frac=1e-5)
      for nm, f in (('actor', c._actor), ('critic', c._critic), ('target_actor', c._target_actor),
                    ('target_critic', c._target_critic)):
        _vec_close(sub(f.dense().cpu().numpy()), g['%sstep%d_%s' % (p, step, nm)], atol_frac=1e-5)


@pytest.mark.parametrize('graph', [False, True])
def test_a2c_controller_learn_golden(golden, graph):
  g = golden('a2c.npz')
  for ci in range(int(g['n_a2c'])):
    p = 'a2c%d_' % ci
    A = g[p + 'net0.actor.2.bias'].shape[0]
    c = A2CController(4, A, lr=1e-3, alpha=0.5, beta=float(g[p + 'beta']), device=DEV, graph=graph)
    c.ac_net.load_state_dict(state_from(g, p + 'net0'))
    for step in range(2):
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in FIELDS + ('returns',)]
      info = c.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        _loss_close(info[k].cpu(), g['%sstep%d_%s' % (p, step, k)])
      _vec_close(c._flat.dense(grad=True).cpu().numpy(), g['%sstep%d_grad' % (p, step)], rtol=1e-4, atol_frac=1e-5)
      _vec_close(c._flat.dense().cpu().numpy(), g['%sstep%d_net' % (p, step)], atol_frac=1e-5)


@pytest.mark.parametrize('graph', [False, True])
def test_ppo_controller_learn_golden(golden, graph):
  g = golden('ppo.npz')
  for ci in range(int(g['n_ppo'])):
    p = 'ppo%d_' % ci
    A = g[p + 'net0.log_std'].shape[1]
    c = PPOController(3, A, lr=3e-4, alpha=0.5, beta=float(g[p + 'beta']), clip_ratio=0.2,
                      max_grad_norm=float(g[p + 'max_grad_norm']), device=DEV, graph=graph)
    c.ac_net.load_state_dict(state_from(g, p + 'net0'))
    for step in range(2):
      names = FIELDS + ('returns', 'old_log_probs', 'advantages')
      batch = [t(g['%sstep%d_%s' % (p, step, f)], DEV) for f in names]
      info = c.learn(*batch)
      for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
        _loss_close(info[k].cpu(), g['%sstep%d_%s' % (p, step, k)])
      _vec_close(c._flat.dense(grad=True).cpu().numpy(), g['%sstep%d_grad' % (p, step)], rtol=1e-4, atol_frac=1e-5)
      _vec_close(c._flat.dense().cpu().numpy(), g['%sstep%d_net' % (p, step)], atol_frac=1e-5)


def test_split_k_linear_matches_nn_linear():
  """models.SkinnyLinear: same forward values, gradients equal up to fp32 summation order."""
  torch.manual_seed(0)
  for (B, i, o) in [(65536, 3, 64), (4096, 64, 64), (8192, 64, 1), (3000, 5, 7)]:
    ref = torch.nn.Linear(i, o).to(DEV)
    lin = models.SkinnyLinear(i, o).to(DEV)
    lin.load_state_dict(ref.state_dict())
    x = torch.randn(B, i, device=DEV)
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    gy = torch.randn(B, o, device=DEV)
    y1, y2 = ref(x1), lin(x2)
    torch.testing.assert_close(y2, y1, rtol=1e-6, atol=1e-6)
    y1.backward(gy)
    y2.backward(gy)
    torch.testing.assert_close(x2.grad, x1.grad, rtol=1e-5, atol=1e-5)
    scale = float(ref.weight.grad.abs().max())
    torch.testing.assert_close(lin.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(lin.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-5 * float(ref.bias.grad.abs().max()))


def test_graphed_learn_tracks_eager_over_many_steps():
  """Replayed-graph learn() == eager learn(), step by step, incl. Adam's step-dependent bias
  correction (device-resident in graph mode) and a second batch shape.  Before every step the
  graph controller is put into the eager one's state, so the comparison is one step deep: Adam
  turns round-off-sized gradient differences into +-lr moves, and whole trajectories of two
  runs whose BLAS reductions are not bitwise reproducible drift apart chaotically."""
  torch.manual_seed(3)
  eager = PPOController(3, 2, lr=1e-3, beta=1e-3, device=DEV, hidden=(64, 64), graph=False)
  graph = PPOController(3, 2, lr=1e-3, beta=1e-3, device=DEV, hidden=(64, 64), graph=True)
  g = torch.Generator(device=DEV).manual_seed(5)
  for step in range(12):
    B = 4096 if step % 4 != 3 else 1000
    obs = torch.randn(B, 3, device=DEV, generator=g)
    act = torch.randn(B, 2, device=DEV, generator=g)
    ret = torch.randn(B, 1, device=DEV, generator=g)
    adv = torch.randn(B, 1, device=DEV, generator=g)
    graph._flat.data.copy_(eager._flat.data)
    graph.ac_net_optim.load_state_dict(eager.ac_net_optim.state_dict())
    with torch.no_grad():
      _, mean, ls = eager.ac_net.heads(obs)
      old = F.normal_log_prob(mean, ls, act) + 0.1
    before = eager._flat.data.clone()
    i1 = eager.learn(obs, act, None, None, None, ret, old, adv)
    i2 = graph.learn(obs, act, None, None, None, ret, old, adv)
    for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
      torch.testing.assert_close(i2[k], i1[k], rtol=1e-5, atol=1e-6)
    assert graph.ac_net_optim.step_count == eager.ac_net_optim.step_count == step + 1
    moved = (eager._flat.data - before).abs()
    diff = (graph._flat.data - eager._flat.data).abs()
    assert float(moved.max()) > 1e-4                         # the step did something
    # ... and the replay did the same thing: a frozen step count would put ~all weights off by
    # >= 10 % of an lr-sized move; round-off on nearly dead units touches a percent or two
    assert float((diff > 5e-5).float().mean()) < 0.05, step


def test_learn_without_hip_device_raises():
  c = PPOController(3, 1, device='cpu')
  x = torch.zeros(4, 3)
  with pytest.raises(Exception, match='HIP'):
    c.learn(x, x[:, :1], x[:, :1], x, x[:, :1], x[:, :1], x[:, :1], x[:, :1])
