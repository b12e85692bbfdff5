"""The role-split kernel k_ac_learn_pair against the four-wave k_ac_learn on one input (tuning build, TRL_LIB=tune:
TRL_LEARN_NW=4 selects the four-wave kernel): losses and gradients, per parameter tensor."""
import os
import sys
os.environ['TRL_LIB'] = 'tune'
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import PPOController

DEV = 'cuda:0'
torch.manual_seed(0)
B = int(os.environ.get('B', 32768))
ctl = PPOController(3, 1, device=DEV, hidden=(64, 64))
g = torch.Generator().manual_seed(1)
obs = torch.randn(B, 3, generator=g).to(DEV); act = torch.randn(B, 1, generator=g).to(DEV)
ret = torch.randn(B, 1, generator=g).to(DEV); adv = torch.randn(B, 1, generator=g).to(DEV)
old = torch.zeros(B, 1, device=DEV)
res = {}
for nw in ('4', '9'):
  os.environ['TRL_LEARN_NW'] = nw
  losses = ctl._fused(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
  torch.cuda.synchronize()
  res[nw] = (losses.clone().cpu(), ctl._flat.grad.clone().cpu())
print('losses 4:', res['4'][0].tolist())
print('losses 9:', res['9'][0].tolist())
print('pairs vs 4 waves: bitwise equal grads', bool(torch.equal(res['4'][1], res['9'][1])), 'losses', bool(torch.equal(res['4'][0], res['9'][0])))
d = (res['4'][1] - res['9'][1]).abs()
print('grad max diff', float(d.max()), 'scale', float(res['4'][1].abs().max()))
off = 0
for name, p in ctl.ac_net.named_parameters():
  n = p.numel()
  print(name, tuple(p.shape), 'maxdiff %.3e' % float(d[off:off + n].max()), 'ref max %.3e' % float(res['4'][1][off:off + n].abs().max()))
  off += n
