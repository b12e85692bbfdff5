"""Fused Linear(+ReLU) layer kernels (csrc/trl_mlp.hip) vs torch's own fp32 Linear/ReLU on the
same device: forward values and all three gradients, over the fan-in / fan-out shapes of the
reference's networks and ragged batch sizes.  Tolerance: fp32 summation-order effects only."""
import numpy as np
import pytest
import torch

from torchrl_amd import config

from torchrl_amd import functional as F
from torchrl_amd.contrib import models

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

SHAPES = [(1, 3, 1), (130, 3, 64), (4096, 64, 64), (5000, 64, 1), (1000, 4, 128), (777, 128, 128), (2048, 128, 6),
          (3000, 3, 256), (1500, 256, 256), (300, 256, 2), (65536, 64, 64), (257, 100, 37), (1024, 1000, 128)]


@pytest.mark.parametrize('B,K,N', SHAPES)
@pytest.mark.parametrize('relu', [False, True])
def test_linear_act_matches_torch(B, K, N, relu):
  g = torch.Generator(device=DEV).manual_seed(B + 7 * K + 13 * N)
  x = torch.randn(B, K, device=DEV, generator=g)
  w = torch.randn(N, K, device=DEV, generator=g) / np.sqrt(K)
  b = torch.randn(N, device=DEV, generator=g)
  gy = torch.randn(B, N, device=DEV, generator=g)
  x1, w1, b1 = (t.clone().requires_grad_() for t in (x, w, b))
  x2, w2, b2 = (t.clone().requires_grad_() for t in (x, w, b))
  y1 = torch.nn.functional.linear(x1, w1, b1)
  if relu:
    y1 = torch.relu(y1)
  y2 = F.linear_act(x2, w2, b2, relu=relu)
  torch.testing.assert_close(y2, y1, rtol=1e-5, atol=1e-5)
  y1.backward(gy)
  y2.backward(gy)
  torch.testing.assert_close(x2.grad, x1.grad, rtol=1e-4, atol=1e-4)
  ws = float(w1.grad.abs().max())
  torch.testing.assert_close(w2.grad, w1.grad, rtol=1e-4, atol=2e-5 * max(ws, 1.0))
  torch.testing.assert_close(b2.grad, b1.grad, rtol=1e-4, atol=2e-5 * max(float(b1.grad.abs().max()), 1.0))


def test_no_input_grad_and_reproducibility():
  g = torch.Generator(device=DEV).manual_seed(1)
  x = torch.randn(10000, 3, device=DEV, generator=g)           # first layer: dx not needed
  w = torch.randn(64, 3, device=DEV, generator=g).requires_grad_()
  b = torch.randn(64, device=DEV, generator=g).requires_grad_()
  gy = torch.randn(10000, 64, device=DEV, generator=g)
  grads = []
  for _ in range(2):
    w.grad = b.grad = None
    F.linear_act(x, w, b, relu=True).backward(gy)
    grads.append((w.grad.clone(), b.grad.clone()))
  assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])   # fixed-order sums


def test_mlp_container_fused_equals_plain(monkeypatch):
  """models._Mlp: fused path and plain torch modules agree for a whole network, forward and
  parameter gradients; state_dict keys are those of the reference's Sequential."""
  torch.manual_seed(0)
  net = models.QNet(4, 2, hidden=(64, 32)).to(DEV)
  assert list(net.state_dict()) == ['net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias', 'net.4.weight',
                                    'net.4.bias']
  x = torch.randn(9000, 4, device=DEV)
  gy = torch.randn(9000, 2, device=DEV)
  out = {}
  for mode in ('1', '0'):
    monkeypatch.setattr(config, 'fused_mlp', mode == '1')
    net.zero_grad()
    y = net(x)
    y.backward(gy)
    out[mode] = (y.detach().clone(), [p.grad.clone() for p in net.parameters()])
  torch.testing.assert_close(out['1'][0], out['0'][0], rtol=1e-5, atol=1e-5)
  for a, b in zip(out['1'][1], out['0'][1]):
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dims,B', [((3, 64, 64, 1), 65536), ((3, 64, 64, 2), 9000), ((4, 64, 6), 8200),
                                    ((17, 33, 5), 10000), ((64, 64, 64, 64), 8192), ((3, 2), 12345),
                                    ((8, 1, 1, 1), 9999)])
def test_whole_network_kernels_match_torch(dims, B):
  """mlp64: forward output and every parameter gradient vs torch's Linear/ReLU stack."""
  g = torch.Generator(device=DEV).manual_seed(sum(dims) + B)
  layers = []
  for i in range(len(dims) - 1):
    layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
    if i + 2 < len(dims):
      layers.append(torch.nn.ReLU())
  ref = torch.nn.Sequential(*layers).to(DEV)
  lin = [m for m in ref if isinstance(m, torch.nn.Linear)]
  x = torch.randn(B, dims[0], device=DEV, generator=g)
  gy = torch.randn(B, dims[-1], device=DEV, generator=g)
  y1 = ref(x)
  y1.backward(gy)
  exp = [p.grad.clone() for p in ref.parameters()]
  ref.zero_grad(set_to_none=True)
  y2 = F.mlp64(x, [m.weight for m in lin], [m.bias for m in lin])
  torch.testing.assert_close(y2, y1, rtol=1e-5, atol=1e-5)
  y2.backward(gy)
  for p, e in zip(ref.parameters(), exp):
    torch.testing.assert_close(p.grad, e, rtol=1e-4, atol=2e-5 * max(1.0, float(e.abs().max())))


def test_mlp_container_uses_whole_network_path_and_matches(monkeypatch):
  torch.manual_seed(1)
  net = models.ActorCriticNet(3, 2, (64, 64)).to(DEV)
  x = torch.randn(20000, 3, device=DEV)
  out = {}
  for mode in ('1', '0'):
    monkeypatch.setattr(config, 'fused_mlp', mode == '1')
    net.zero_grad()
    v, mean, _ = net.heads(x)
    (v.sum() + (mean * mean).sum()).backward()
    out[mode] = (v.detach().clone(), mean.detach().clone(),
                 [p.grad.clone() for p in net.parameters() if p.grad is not None])
  torch.testing.assert_close(out['1'][0], out['0'][0], rtol=1e-5, atol=1e-5)
  torch.testing.assert_close(out['1'][1], out['0'][1], rtol=1e-5, atol=1e-5)
  for a, b in zip(out['1'][2], out['0'][2]):
    torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * max(1.0, float(b.abs().max())))
  # DDPG's tanh tail stays on torch, after the fused stack
  actor = models.DDPGActorNet(3, 1, 64).to(DEV)
  monkeypatch.setattr(config, 'fused_mlp', True)
  a1 = actor(x)
  monkeypatch.setattr(config, 'fused_mlp', False)
  torch.testing.assert_close(a1, actor(x), rtol=1e-5, atol=1e-6)


def test_direct_grads_equal_autograd_accumulation():
  """direct_grads=True writes the same numbers into .grad that autograd would accumulate into
  zeroed gradients (bit for bit: same kernels, same fixed-order sums)."""
  from torchrl_amd.optim import FlatParams
  torch.manual_seed(2)
  net = models.ActorCriticNet(3, 1, (64, 64)).to(DEV)
  flat = FlatParams(net)
  x = torch.randn(16384, 3, device=DEV)
  res = []
  for direct in (False, True):
    net.critic.direct_grads = net.actor.direct_grads = direct
    flat.zero_grad()
    v, mean, _ = net.heads(x)
    ((v * v).mean() + mean.sum()).backward()
    res.append(flat.grad.clone())
  assert torch.equal(res[0], res[1]) and float(res[0].abs().max()) > 0
  with pytest.raises(ValueError):
    w = torch.zeros(2, 3, device=DEV, requires_grad=True)
    F.mlp64(x, [w], [torch.zeros(2, device=DEV, requires_grad=True)], direct_grads=True)   # no .grad yet


def test_bad_arguments():
  x = torch.zeros(8, 4, device=DEV)
  with pytest.raises(Exception):
    F.linear_act(x, torch.zeros(300, 4, device=DEV), torch.zeros(300, device=DEV))       # fan-out > 256
  with pytest.raises(Exception):
    F.linear_act(x.cpu(), torch.zeros(2, 4), torch.zeros(2))
