"""The whole-step kernel's second variant, products='bf16x6' (csrc/trl_learn.hip: k_ac_learn_pair_bx, C ABI
trl_ppo_learn_fused_bx6 / trl_a2c_learn_fused_bx6): the three 64 x 64 products of each network on the bf16 MFMA with
both operands split exactly into three bf16 parts.  It is NOT the parity mode (products='f32' stays that, and the
default) -- these tests are its guard rails:

  * the product alone against float64 on cancelling inputs (trl_bx6_product_probe): error bounded by
    1.2e-7 sum |a_k| |b_k| + the f32 rounding of a 64-term sum, and no worse than an f32 matmul's;
  * the step against the oracle's torch-CPU PPOController.learn (ppo_controller.py:53-80) at the exact variant's own
    tolerances (losses 1e-6 relative: the north star's bar);
  * the step against the exact-f32 kernel on the same weights.

The reference's golden fixtures (tests/golden/ppo.npz, a2c.npz) hold its own network, ONE hidden layer of 256: there is no
hidden x hidden product in it, products='bf16x6' changes nothing for those shapes and they run the exact kernels
(tests/test_gpu_learn.py)."""
import copy

import numpy as np
import pytest
import torch

from oracle import loss_oracle
from torchrl_amd import _lib
from torchrl_amd.contrib.controllers import PPOController, A2CController

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _probe(w, x):
  y = torch.empty(64, x.shape[1], device=DEV)
  _lib.check(_lib.lib().trl_bx6_product_probe(w.data_ptr(), x.data_ptr(), y.data_ptr(), x.shape[1], _lib.stream_ptr(torch.device(DEV))))
  torch.cuda.synchronize()
  return y.cpu().double()


@pytest.mark.parametrize('kind', ['gauss', 'cancelling', 'wide_range', 'tiny'])
def test_bx6_product_error_bound_against_float64(kind):
  g = torch.Generator().manual_seed(5)
  n = 4096 + 17
  w = torch.randn(64, 64, generator=g)
  x = torch.randn(64, n, generator=g)
  if kind == 'cancelling':
    # pairs of terms that cancel to ~1e-7 of their size: the dot product is the sum of what the low parts carry
    w[:, 1::2] = -w[:, 0::2] * (1 + 3e-7 * torch.randn(64, 32, generator=g))
    x[1::2, :] = x[0::2, :] * (1 + 3e-7 * torch.randn(32, n, generator=g))
  elif kind == 'wide_range':
    w = w * torch.exp(8 * torch.randn(64, 64, generator=g))
    x = x * torch.exp(8 * torch.randn(64, n, generator=g))
  elif kind == 'tiny':
    w, x = w * 1e-18, x * 1e-15        # products near the bottom of the normal range: low parts go subnormal / to zero
  exact = w.double() @ x.double()
  mag = w.double().abs() @ x.double().abs()                     # sum |a_k| |b_k|
  got = _probe(w.to(DEV).contiguous(), x.to(DEV).contiguous())
  err = (got - exact).abs()
  f32 = ((w @ x).double() - exact).abs()                         # torch's f32 matmul on the host, for scale
  if kind == 'tiny':
    # the matrix pipe flushes partial products below the normal range (2^-126) to zero: per term at most three of them
    # (the exact product of such operands is itself below 1e-31: an absolute bound takes over from the relative one)
    assert float(err.max()) <= 1.2e-7 * float(mag.max()) + 64 * 3 * 2.0 ** -126
    return
  # dropped partial products: <= 2^-23 per term; accumulation in f32: <= 64 roundings of at most the magnitude sum
  assert float((err / mag).max()) <= 2.0 ** -23 + 64 * 2.0 ** -24, float((err / mag).max())
  # in practice far inside that: the typical error is a few 1e-8 of the magnitude sum, as for an f32 product (one f32
  # rounding of a sum that a single term dominates is already 2^-24 = 6e-8 of it: the wide_range case)
  assert float((err / mag).mean()) <= 1.0e-7, float((err / mag).mean())
  assert float(err.max()) <= 3.0 * float(f32.max()) + 1e-30, (float(err.max()), float(f32.max()))


def _batch(obs_size, A, B, seed):
  g = torch.Generator().manual_seed(seed)
  return (torch.randn(B, obs_size, generator=g), torch.randn(B, A, generator=g), torch.randn(B, 1, generator=g),
          torch.randn(B, 1, generator=g), g)


CASES = [(3, 1, (64, 64), 40001), (4, 2, (64, 64), 33000), (2, 2, (48, 33), 50000), (1, 1, (64, 20), 32768),
         (3, 1, (64, 64), 300000)]


@pytest.mark.parametrize('obs_size,A,hidden,B', CASES)
def test_bx6_learn_matches_oracle(obs_size, A, hidden, B):
  torch.manual_seed(obs_size + 10 * A + B)
  ctl = PPOController(obs_size, A, lr=3e-4, alpha=0.5, beta=1e-2, clip_ratio=0.2, max_grad_norm=0.05, device=DEV,
                      hidden=hidden, products='bf16x6')
  assert ctl._fused is not None and ctl._fused.bx6_takes(B)
  with torch.no_grad():
    ctl.ac_net.log_std.add_(torch.linspace(-0.3, 0.2, A, device=DEV).view(1, A))
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.Adam(ref.parameters(), lr=3e-4)
  obs, act, ret, adv, g = _batch(obs_size, A, B, 7 + B)
  with torch.no_grad():
    mean = ref.actor(obs)
    old = torch.distributions.Normal(mean, ref.log_std.exp().expand_as(mean)).log_prob(act)
    old = old + 0.3 * torch.randn(B, A, generator=g)        # ratios on both sides of the clip range
  exp = loss_oracle.ppo_learn(ref, ref_opt, obs, act, None, None, None, ret, old, adv, clip=0.2, alpha=0.5, beta=1e-2,
                              max_grad_norm=0.05)
  exp_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])      # after clip_grad_value_
  got = ctl.learn(obs.to(DEV), act.to(DEV), None, None, None, ret.to(DEV), old.to(DEV), adv.to(DEV))
  for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
    e, v = float(exp[k]), float(got[k].cpu())
    assert abs(v - e) <= 1e-6 * abs(e) + 2e-7, (k, v, e)      # the exact variant's bar (tests/test_gpu_learn.py)
  got_grad = ctl._flat.dense(grad=True).cpu()
  scale = float(exp_grad.abs().max())
  np.testing.assert_allclose(got_grad.numpy(), exp_grad.numpy(), rtol=1e-4, atol=2e-5 * scale)
  exp_w = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
  got_w = ctl._flat.dense().cpu()
  assert float((got_w - exp_w).abs().max()) <= 2.5 * 3e-4
  assert float(((got_w - exp_w).abs() > 1e-5).float().mean()) < 0.02


@pytest.mark.parametrize('obs_size,A,B', [(3, 1, 65536), (4, 2, 40000)])
def test_bx6_against_the_exact_kernel(obs_size, A, B):
  """Same weights, same batch, raw gradients (no clamp in the way): the two variants differ by the products' rounding
  only.  Also: the variant is reproducible run to run, and the default stays exact."""
  outs = []
  for products in ('f32', 'bf16x6'):
    torch.manual_seed(21)
    ctl = PPOController(obs_size, A, lr=1e-3, alpha=0.5, beta=1e-3, max_grad_norm=1e9, device=DEV, hidden=(64, 64),
                        products=products)
    assert ctl._fused.bx6_takes(B) == (products == 'bf16x6')
    obs, act, ret, adv, g = _batch(obs_size, A, B, 3)
    obs, act, ret, adv = (t.to(DEV) for t in (obs, act, ret, adv))
    _, _, logp = ctl._fused.forward(obs, act)
    old = logp + 0.2 * torch.randn(B, A, generator=g).to(DEV)
    res = []
    for _ in range(2):
      losses = ctl._fused(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
      res.append((losses.clone(), ctl._flat.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])      # fixed-order sums in both
    outs.append(res[0])
  torch.testing.assert_close(outs[0][0][:3], outs[1][0][:3], rtol=1e-6, atol=2e-7)
  scale = float(outs[0][1].abs().max())
  # (a gradient is a mean over the batch of terms of either sign: absolute differences are judged against the largest)
  torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-5 * scale)
  assert not torch.equal(outs[0][1], outs[1][1])      # (it IS another arithmetic: the variant is not silently the exact one)
  assert PPOController(3, 1, device=DEV, hidden=(64, 64))._fused.products == 'f32'


def test_bx6_a2c_and_shapes_outside_the_variant():
  """A2C through the same kernel pair; shapes the variant does not cover keep the exact kernels (bx6_takes is False)
  and the C entry refuses them."""
  outs = []
  B = 40000
  for products in ('f32', 'bf16x6'):
    torch.manual_seed(4)
    ctl = A2CController(4, 2, lr=1e-3, alpha=0.5, beta=1e-2, device=DEV, hidden=(64, 64), products=products)
    g = torch.Generator().manual_seed(8)
    obs = torch.randn(B, 4, generator=g).to(DEV)
    act = torch.randint(0, 2, (B, 1), generator=g).float().to(DEV)
    ret = torch.randn(B, 1, generator=g).to(DEV)
    info = ctl.learn(obs, act, None, None, None, ret)
    outs.append(({k: float(v) for k, v in info.items()}, ctl._flat.dense(grad=True).clone()))
  for k in ('actor_loss', 'critic_loss', 'entropy_loss'):
    assert abs(outs[0][0][k] - outs[1][0][k]) <= 1e-6 * abs(outs[0][0][k]) + 2e-7, k
  scale = float(outs[0][1].abs().max())
  torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-5 * scale)
  wide = PPOController(3, 1, device=DEV, hidden=(256,), products='bf16x6')
  assert wide._fused is not None and not wide._fused.bx6_takes(1 << 20)      # one hidden layer: no square product
  assert not PPOController(8, 1, device=DEV, hidden=(64, 64), products='bf16x6')._fused.bx6_takes(1 << 20)
  small = PPOController(3, 1, device=DEV, hidden=(64, 64), products='bf16x6')
  assert not small._fused.bx6_takes(4096)                                    # small batches: the exact kernels
  with pytest.raises(ValueError):
    PPOController(3, 1, device=DEV, hidden=(64, 64), products='fp8')


@pytest.mark.parametrize('obs_size,A,B', [(3, 1, 40001), (4, 2, 33000)])
def test_bx6_forward_kernel_matches_torch_and_the_learner_forward(obs_size, A, B):
  """trl_ppo_forward_fused_bx6 (compute_return's value / log-prob pass with layer 1 on the bf16 MFMA): against the torch
  modules at the exact forward's tolerances, and -- same staging, split and accumulation order -- the same BITS as the
  forward inside trl_ppo_learn_fused_bx6: with its values as the returns the learner's critic loss and critic gradients are
  exactly zero, with its log-probs as the old ones no ratio leaves [1 - 1e-30, 1 + 1e-30]."""
  torch.manual_seed(B + obs_size)
  ctl = PPOController(obs_size, A, device=DEV, hidden=(64, 64), products='bf16x6')
  with torch.no_grad():
    ctl.ac_net.log_std.add_(torch.linspace(-0.4, 0.3, A, device=DEV).view(1, A))
  ref = copy.deepcopy(ctl.ac_net).cpu()
  obs, act, _, adv, _ = _batch(obs_size, A, B, 3)
  with torch.no_grad():
    ev, em = ref.critic(obs), ref.actor(obs)
    elp = torch.distributions.Normal(em, ref.log_std.exp().expand_as(em)).log_prob(act)
  obs, act, adv = obs.to(DEV), act.to(DEV), adv.to(DEV)
  assert ctl._fused.bx6_takes(B)
  value, mean, logp = ctl._fused.forward(obs, act)
  torch.testing.assert_close(value.cpu(), ev.view(-1), rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(mean.cpu(), em, rtol=1e-5, atol=2e-6)
  torch.testing.assert_close(logp.cpu(), elp, rtol=1e-5, atol=1e-5)
  ret = value.view(B, 1).clone()
  res = []
  for clip in (0.2, 1e-30):
    ctl._fused(obs, act, ret, logp, adv, clip, 0.5, 1e-3)
    res.append(ctl._flat.grad.clone())
  losses = ctl._fused(obs, act, ret, logp, adv, 0.2, 0.5, 1e-3).cpu()
  assert float(losses[1]) == 0.0
  off = 0
  for name, p_ in ctl.ac_net.named_parameters():
    if name.startswith('critic'):
      assert not res[0][off:off + p_.numel()].any(), name
    off += p_.numel()
  assert torch.equal(res[0], res[1])
