"""Times of the whole-step PPO kernel (trl_ppo_learn_fused) and of learn() around it on one GPU.
  python tools/learn_probe.py [B] [hidden,hidden] [A]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
hidden = tuple(int(h) for h in (sys.argv[2] if len(sys.argv) > 2 else '64,64').split(','))
A = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device('cuda', 0)
torch.manual_seed(0)
for graph in (False, True):
  ctl = PPOController(3, A, lr=3e-4, beta=1e-3, device=dev, hidden=hidden, graph=graph)
  obs, act = torch.randn(B, 3, device=dev), torch.randn(B, A, device=dev)
  ret, adv, old = torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev), torch.randn(B, A, device=dev) - 1
  for _ in range(5):
    ctl.learn(obs, act, None, None, None, ret, old, adv)
  torch.cuda.synchronize()
  n = 300
  t0 = time.perf_counter()
  for _ in range(n):
    ctl.learn(obs, act, None, None, None, ret, old, adv)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print('graph=%s: learn() %.1f us/call wall (host issue %.1f us/call)' % (graph, (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6))
  if not graph and ctl._fused is not None:
    f = ctl._fused
    g = torch.cuda.CUDAGraph()
    f(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
      for _ in range(20):
        f(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    print('k_ac_learn + k_learn_reduce: %.1f us per pair (graph of 20)' % (a.elapsed_time(b) * 1e3 / 20))
