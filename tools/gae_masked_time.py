"""Masked-GAE kernel time at T=2048, N=4096 (graph of 20 launches, warm inputs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import functional as F
dev = torch.device('cuda', 0)
T, N = 2048, 4096
r, v, b = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.randn(N, device=dev)
d = (torch.rand(T, N, device=dev) < 0.01).float()
out = (torch.empty(T, N, device=dev), torch.empty(T, N, device=dev))
for masked in (False, True):
  fn = lambda: F.gae(r, v, b, 0.99, 0.95, done=d if masked else None, masked=masked, out=out)
  for _ in range(3): fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(20): fn()
  g.replay(); torch.cuda.synchronize()
  a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record(); g.replay(); e.record(); torch.cuda.synchronize()
  us = a.elapsed_time(e) * 1e3 / 20
  nbytes = (20 if masked else 16) * T * N
  print('masked=%s: %.1f us per launch, %.0f GB/s algorithmic (%d B per step)' % (masked, us, nbytes / us / 1e3, 20 if masked else 16))
