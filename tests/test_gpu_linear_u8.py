"""First layer on uint8 frames (csrc/trl_linear_u8.hip) against float64 references of the same
products, and a QNet / DQNController.learn fed uint8 frames against the same fed float32 frames
(what the reference stores, storage.py:65-67): the cast is exact, so only the summation order differs."""
import copy

import numpy as np
import pytest
import torch

from torchrl_amd import functional as F
from torchrl_amd.contrib import models
from torchrl_amd.contrib.controllers import DQNController

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _case(B, K, N, seed):
  g = torch.Generator().manual_seed(seed)
  x = torch.randint(0, 256, (B, K), dtype=torch.uint8, generator=g)
  w = (torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5
  b = torch.rand(N, generator=g) * 2 - 1
  return x, w, b


@pytest.mark.parametrize('B,K,N', [(256, 28224, 128), (32, 64, 128), (96, 320, 256), (200, 96, 128), (1, 32, 128)])
@pytest.mark.parametrize('relu', [False, True])
def test_forward_matches_float64(B, K, N, relu):
  x, w, b = _case(B, K, N, 1)
  ref = x.double() @ w.double().t() + b.double()
  if relu:
    ref = ref.clamp_min(0)
  xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
  with torch.no_grad():
    got = F._LinearU8.apply(xd, wd, bd, 1 if relu else 0).cpu().double()
  scale = float(ref.abs().max())
  np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-6, atol=2e-6 * scale)


@pytest.mark.parametrize('B,K,N', [(256, 28224, 128), (32, 64, 128), (96, 324, 256), (64, 4, 128)])
def test_wgrad_matches_float64(B, K, N):
  x, _, _ = _case(B, K, N, 2)
  g = torch.randn(B, N, generator=torch.Generator().manual_seed(3)) / B
  ref = g.double().t() @ x.double()
  dw = torch.full((N, K), float('nan'), device=DEV)
  from torchrl_amd import _lib
  gd, xd = g.to(DEV), x.to(DEV)
  _lib.check(_lib.lib().trl_linear_u8_wgrad(gd.data_ptr(), xd.data_ptr(), B, K, N, dw.data_ptr(), _lib.stream_ptr(torch.device(DEV))))
  got = dw.cpu().double()
  assert torch.isfinite(got).all()          # every element written
  scale = float(ref.abs().max())
  np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-6, atol=2e-6 * scale)


def test_rejects_unsupported_shapes():
  from torchrl_amd import _lib
  x, w, b = _case(32, 48, 128, 4)     # K % 32 != 0
  assert not F.linear_u8_supported(x.to(DEV), w.to(DEV))
  assert _lib.lib().trl_linear_u8_fwd_workspace_bytes(32, 128, 48) == 0
  xd, wd, bd, y = x.to(DEV), w.to(DEV), b.to(DEV), torch.empty(32, 128, device=DEV)
  rc = _lib.lib().trl_linear_u8_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 32, 48, 128, 0, y.data_ptr(),
                                    y.data_ptr(), None)
  assert rc == -1 and b'K % 32' in _lib.lib().trl_last_error()


def test_qnet_on_uint8_frames_equals_qnet_on_float_frames():
  torch.manual_seed(0)
  net = models.QNet(28224, 6).to(DEV)
  x = torch.randint(0, 256, (64, 28224), dtype=torch.uint8, device=DEV)
  q8 = net(x)
  q8.sum().backward()
  g8 = [p.grad.clone() for p in net.parameters()]
  net.zero_grad()
  qf = net(x.float())
  qf.sum().backward()
  gf = [p.grad.clone() for p in net.parameters()]
  np.testing.assert_allclose(q8.detach().cpu().numpy(), qf.detach().cpu().numpy(), rtol=1e-5, atol=1e-5 * float(qf.abs().max()))
  for a, b_ in zip(g8, gf):
    np.testing.assert_allclose(a.cpu().numpy(), b_.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(b_.abs().max()))
  # a shape the HIP layer does not take is cast and runs on the plain modules
  small = models.QNet(20, 3).to(DEV)
  xs = torch.randint(0, 256, (8, 20), dtype=torch.uint8, device=DEV)
  assert torch.equal(small(xs), small(xs.float()))


@pytest.mark.parametrize('graph', [False, True])
def test_dqn_learn_on_uint8_frames_matches_float_frames(graph):
  K, A, B = 28224, 6, 64
  torch.manual_seed(1)
  c8 = DQNController(K, A, double_dqn=True, gamma=0.99, lr=1e-3, n_update_interval=2, device=DEV, graph=graph)
  cf = DQNController(K, A, double_dqn=True, gamma=0.99, lr=1e-3, n_update_interval=2, device=DEV, graph=graph)
  cf._q.data.copy_(c8._q.data)
  cf._target.data.copy_(c8._target.data)
  g = torch.Generator(device=DEV).manual_seed(5)
  for step in range(3):
    obs = torch.randint(0, 256, (B, K), dtype=torch.uint8, device=DEV, generator=g)
    nxt = torch.randint(0, 256, (B, K), dtype=torch.uint8, device=DEV, generator=g)
    act = torch.randint(0, A, (B, 1), device=DEV, generator=g).float()
    rew = torch.randn(B, 1, device=DEV, generator=g)
    done = (torch.rand(B, 1, device=DEV, generator=g) < 0.1).float()
    l8 = c8.learn(obs, act, rew, nxt, done)['td_loss']
    lf = cf.learn(obs.float(), act, rew, nxt.float(), done)['td_loss']
    np.testing.assert_allclose(float(l8), float(lf), rtol=2e-5)
  np.testing.assert_allclose(c8._q.data.cpu().numpy(), cf._q.data.cpu().numpy(), rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize('K,A,hidden,B,double_dqn', [(64, 3, 128, 32, True), (320, 6, 128, 96, False),
                                                     (28224, 6, 128, 64, True), (96, 2, (256, 64), 64, True)])
def test_dqn_uint8_chain_matches_oracle(K, A, hidden, B, double_dqn):
  """trl_dqn_learn_fused_u8 (8 launches, frames read as uint8) against the oracle's torch-CPU
  restatement of DQNController.learn (dqn_controller.py:61-83) on the same frames as float32."""
  from oracle import loss_oracle
  rs = np.random.RandomState(K + B)
  torch.manual_seed(K + 7 * B)
  ctl = DQNController(K, A, double_dqn=double_dqn, lr=1e-3, n_update_interval=1000, device=DEV, hidden=hidden)
  assert ctl._fused is not None and ctl._fused.u8_ok
  with torch.no_grad():       # frames are 0..255: scale the first layer so activations stay O(1)
    for net in (ctl.q_net, ctl.target_q_net):
      net.net[0].weight.mul_(1.0 / 64)
  ctl._target.data.add_(torch.from_numpy(rs.randn(ctl._target.data.numel()).astype(np.float32)).to(DEV) * 1e-3)
  ref_q, ref_t = copy.deepcopy(ctl.q_net).cpu(), copy.deepcopy(ctl.target_q_net).cpu()
  opt = torch.optim.Adam(ref_q.parameters(), lr=1e-3)
  for step in range(2):
    obs = torch.from_numpy(rs.randint(0, 256, (B, K)).astype(np.uint8))
    nxt = torch.from_numpy(rs.randint(0, 256, (B, K)).astype(np.uint8))
    act = torch.from_numpy(rs.randint(0, A, (B, 1)).astype(np.float32))
    rew = torch.from_numpy(rs.randn(B, 1).astype(np.float32))
    done = torch.from_numpy((rs.rand(B, 1) < 0.1).astype(np.float32))
    # the same loss in float64 from the same float32 weights: the yardstick for both float32 runs
    q64, t64 = copy.deepcopy(ref_q).double(), copy.deepcopy(ref_t).double()
    with torch.no_grad():
      cur = q64(obs.double()).gather(1, act.long())
      nq = t64(nxt.double()).gather(1, q64(nxt.double()).max(1, keepdim=True)[1]) if double_dqn else \
          t64(nxt.double()).max(1)[0].unsqueeze(1)
      exact = float(((rew.double() + 0.99 * nq * (1.0 - done.double()) - cur) ** 2).mean())
    exp = loss_oracle.dqn_learn(ref_q, ref_t, opt, obs.float(), act, rew, nxt.float(), done, gamma=0.99,
                                double_dqn=double_dqn)
    assert ctl._fused.usable(obs.to(DEV))
    got = ctl.learn(obs.to(DEV), act.to(DEV), rew.to(DEV), nxt.to(DEV), done.to(DEV))
    e, v = float(exp['td_loss']), float(got['td_loss'])
    assert abs(v - e) <= 2e-5 * abs(e) + 1e-7, (step, v, e)
    # 1e-6 between two float32 runs is not defined for a 28,224-term dot product (either run is ~1e-6
    # from exact arithmetic by itself): the HIP path must be as close to float64 as the oracle's run is
    assert abs(v - exact) <= max(2.0 * abs(e - exact), 2e-6 * abs(exact)), (step, v, e, exact)
    for (name, p), q in zip(ctl.q_net.named_parameters(), ref_q.parameters()):
      gq = q.grad.numpy()
      np.testing.assert_allclose(p.grad.cpu().numpy(), gq, rtol=1e-3, atol=2e-5 * float(np.abs(gq).max()) + 1e-9,
                                 err_msg='%s grad, step %d' % (name, step))
  for (name, p), q in zip(ctl.q_net.named_parameters(), ref_q.parameters()):
    # Adam's first steps move every weight by ~lr whatever the gradient's size: compare on that scale
    np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), rtol=0, atol=2e-4, err_msg=name)


@pytest.mark.parametrize('K,A,B,cap,pushes', [(28224, 6, 64, 300, (200, 180)),      # Atari-shaped records, ring wrapped
                                              (1024, 4, 32, 100, (60,)),            # filling ring (frames >= 1 KiB: 16 B-aligned columns)
                                              (2048, 3, 96, 128, (128, 50))])
def test_dqn_learn_indexed_reads_the_ring_in_place(K, A, B, cap, pushes):
  """DQNController.learn_indexed(buffer, idx) -- the first layer and its weight gradient fetch the sampled uint8 frames
  where the replay ring holds them (trl_dqn_learn_fused_u8_ring) -- against `learn(*buffer[idx])` (gather, then the dense
  uint8 step) on the same indices: three consecutive steps, losses, weights and moments bit for bit; host indices drawn
  as examples/dqn.py:40 draws them (torch's generator ends in the same state) and device indices."""
  from torchrl_amd.utils.storage import TensorTupleDataset
  dts = [torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32]
  g = torch.Generator(device=DEV).manual_seed(K + B)

  def fill(ds):
    gg = torch.Generator(device=DEV).manual_seed(cap)
    for k in pushes:
      ds.extend(torch.randint(0, 256, (k, K), device=DEV, dtype=torch.uint8, generator=gg),
                torch.randint(0, A, (k, 1), device=DEV, generator=gg).float(),
                torch.randn(k, 1, device=DEV, generator=gg),
                torch.randint(0, 256, (k, K), device=DEV, dtype=torch.uint8, generator=gg),
                (torch.rand(k, 1, device=DEV, generator=gg) < 0.1).float())
  ds = TensorTupleDataset(size=cap, device=DEV, storage_dtypes=dts, gather_dtypes=dts)
  fill(ds)
  n = len(ds)
  assert n == min(cap, sum(pushes))
  ctls = []
  for _ in range(2):
    torch.manual_seed(3)
    c = DQNController(K, A, double_dqn=True, lr=1e-3, n_update_interval=2, device=DEV, hidden=128)
    with torch.no_grad():
      for net in (c.q_net, c.target_q_net):
        net.net[0].weight.mul_(1.0 / 64)
    c._target.data.add_(1e-3)
    ctls.append(c)
  a, b = ctls
  assert torch.equal(a._q.data, b._q.data)
  for step in range(3):
    if step < 2:
      torch.manual_seed(100 + step)
      idx = ds.sample_indices(B)
      state = torch.get_rng_state()
      torch.manual_seed(100 + step)
      assert torch.equal(idx, torch.randperm(n)[:B]) and torch.equal(state, torch.get_rng_state())
    else:
      idx = torch.randint(0, n, (B,), device=DEV, generator=g)      # device-resident indices
    view = ds.ring_view(idx)
    assert a._fused.ring_usable(view)
    view['done']()
    la = a.learn_indexed(ds, idx, in_place=True)['td_loss']
    lb = b.learn(*ds[idx])['td_loss']
    torch.cuda.synchronize()
    assert float(la) == float(lb), (step, float(la), float(lb))
  assert torch.equal(a._q.data, b._q.data) and torch.equal(a._target.data, b._target.data)
  assert torch.equal(a.q_net_optim.exp_avg, b.q_net_optim.exp_avg) and torch.equal(a.q_net_optim.exp_avg_sq, b.q_net_optim.exp_avg_sq)
  assert torch.equal(a._q.grad, b._q.grad)
  # a buffer the in-place step does not take (float32 frames) goes through learn(*buffer[idx])
  dsf = TensorTupleDataset(size=64, device=DEV)
  dsf.extend(torch.randn(64, 8, device=DEV), torch.randint(0, 2, (64, 1), device=DEV).float(), torch.randn(64, 1, device=DEV),
             torch.randn(64, 8, device=DEV), torch.zeros(64, 1, device=DEV))
  cf = DQNController(8, 2, device=DEV)
  assert torch.isfinite(cf.learn_indexed(dsf, torch.arange(32), in_place=True)['td_loss']).item()


@pytest.mark.parametrize('B,double_dqn', [(256, True), (64, False), (1024, True)])
def test_dqn_uint8_middle_chain_equals_the_launches(B, double_dqn):
  """With trl_offpolicy_set_chain(2) the uint8 DQN step issues layer 1, layer 2, the TD loss and the input gradients as ONE
  launch (k_chain with the loss as a one-workgroup member; opt-in: measured slower than the four launches it replaces);
  trl_offpolicy_set_chain(0) issues them as four.  Three consecutive steps: up to 256 rows
  the loss is the same bits (a row per thread, k_td_loss's own block sum) and so are weights and moments; beyond that the
  loss is another fixed-order sum (1e-6) of the same rows' terms, everything else the same bits."""
  from torchrl_amd import _lib
  K, A = 1024, 6
  g = torch.Generator(device=DEV).manual_seed(B)
  batches = [(torch.randint(0, 256, (B, K), device=DEV, dtype=torch.uint8, generator=g),
              torch.randint(0, A, (B, 1), device=DEV, generator=g).float(), torch.randn(B, 1, device=DEV, generator=g),
              torch.randint(0, 256, (B, K), device=DEV, dtype=torch.uint8, generator=g),
              (torch.rand(B, 1, device=DEV, generator=g) < 0.1).float()) for _ in range(3)]
  res = []
  try:
    for chain in (2, 0):
      _lib.lib().trl_offpolicy_set_chain(chain)
      torch.manual_seed(3)
      c = DQNController(K, A, double_dqn=double_dqn, lr=1e-3, n_update_interval=2, device=DEV, hidden=128)
      with torch.no_grad():
        for net in (c.q_net, c.target_q_net):
          net.net[0].weight.mul_(1.0 / 64)
      c._target.data.add_(1e-3)
      losses = [c.learn(*b)['td_loss'].clone() for b in batches]
      torch.cuda.synchronize()
      res.append((torch.stack(losses), c._q.data.clone(), c.q_net_optim.exp_avg.clone(), c.q_net_optim.exp_avg_sq.clone()))
  finally:
    _lib.lib().trl_offpolicy_set_chain(1)
  assert torch.isfinite(res[0][0]).all()
  if B <= 256:
    assert torch.equal(res[0][0], res[1][0])
  else:
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-6, atol=1e-7)
  for x, y in zip(res[0][1:], res[1][1:]):
    assert torch.equal(x, y)
