"""Host-side cost of one PPOController.learn() at the reference's minibatch (B = 64): cProfile over N calls."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
dev = torch.device('cuda', 0)
torch.manual_seed(0)
ctl = PPOController(3, 1, device=dev, hidden=(64, 64))
obs, act = torch.randn(B, 3, device=dev), torch.randn(B, 1, device=dev)
ret, adv, old = torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev) - 1
for _ in range(200):
  ctl.learn(obs, act, None, None, None, ret, old, adv)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
  ctl.learn(obs, act, None, None, None, ret, old, adv)
torch.cuda.synchronize()
print('wall %.2f us per learn()' % ((time.perf_counter() - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N // 4):
  ctl.learn(obs, act, None, None, None, ret, old, adv)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
