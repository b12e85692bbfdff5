#!/usr/bin/env python3
"""Host cost of the bit-exact sampler (trl_randperm_prefix) beside torch.randperm, on this host's CPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torchrl_amd.utils.sampler import randperm_prefix

for n in (1_000_000, 3_000_000, 10_000_000):
  for _ in range(8):      # past the point where the jump polynomials of a full ring's two skip lengths exist
    randperm_prefix(n, 128)
  k = 40
  t = time.perf_counter()
  for _ in range(k):
    randperm_prefix(n, 128)
  a = (time.perf_counter() - t) / k * 1e3
  t = time.perf_counter()
  for _ in range(3):
    torch.randperm(n)[:128]
  b = (time.perf_counter() - t) / 3 * 1e3
  print('n=%d  randperm_prefix %.3f ms   torch.randperm %.2f ms' % (n, a, b), flush=True)
