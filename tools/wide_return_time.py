#!/usr/bin/env python3
"""compute_return_batched over a 1 M-row rollout for nets outside the register-chained forward kernel: the tile pipeline's
forward-only form (trl_ac_forward_tiles) against the module path (fused=False) (diagnostic)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import PPOController, A2CController  # noqa: E402

dev = 'cuda:0'
T, N = 2048, 512
for kind, obs_size, A in (('ppo', 8, 4), ('ppo', 17, 6), ('a2c', 17, 6)):
  obs = torch.randn(T, N, obs_size, device=dev)
  reward = torch.randn(T, N, device=dev)
  last_obs, last_done = torch.randn(N, obs_size, device=dev), torch.zeros(N, device=dev)
  action = torch.randn(T, N, A, device=dev)
  line = '%s obs %2d actions %d hidden 256, T x N = %d x %d:' % (kind, obs_size, A, T, N)
  for fused in (None, False):
    c = (PPOController if kind == 'ppo' else A2CController)(obs_size, A, device=dev, fused=fused)
    fn = (lambda: c.compute_return_batched(obs, action, reward, last_obs, last_done)) if kind == 'ppo' else \
        (lambda: c.compute_return_batched(obs, reward, last_obs, last_done))
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
      fn()
    torch.cuda.synchronize()
    line += '  %s %.3f ms' % ('tiles' if fused is None else 'modules', (time.perf_counter() - t0) / 10 * 1e3)
  print(line, flush=True)
