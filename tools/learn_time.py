"""Device time of one fused PPO learn() launch pair at a batch (tuning build; env knobs
HIDDEN / OBS / ACT, TRL_LEARN_NW / TRL_LEARN_PRIO; argv: batch, then K=V,K=V labels / env settings per timed run)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TRL_LIB', 'tune')
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
A = int(os.environ.get('ACT', 1))
OBS = int(os.environ.get('OBS', 3))
dev = torch.device('cuda', 0)
torch.manual_seed(0)
HIDDEN = tuple(int(x) for x in os.environ.get('HIDDEN', '64,64').split(','))
ctl = PPOController(OBS, A, device=dev, hidden=HIDDEN if len(HIDDEN) > 1 else HIDDEN[0])
obs, act = torch.randn(B, OBS, device=dev), torch.randn(B, A, device=dev)
ret, adv, old = torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev), torch.randn(B, A, device=dev) - 1
f = ctl._fused
for cfg in sys.argv[2:]:
  for kv in cfg.split(','):
    k, v = kv.split('=')
    os.environ[k] = v
  for _ in range(20):
    f(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  n = 100
  e0.record()
  for _ in range(n):
    f(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
  e1.record()
  torch.cuda.synchronize()
  print('%-40s %8.2f us per learn()' % (cfg, e0.elapsed_time(e1) * 1000.0 / n), flush=True)
