#!/usr/bin/env python3
"""Where an off-policy step's time goes around the sampler (DDPG, 1 M-row Pendulum-shaped ring)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import DDPGController  # noqa: E402
from torchrl_amd.utils.sampler import randperm_prefix  # noqa: E402
from torchrl_amd.utils.storage import TensorTupleDataset  # noqa: E402

DEV = torch.device('cuda:0')


def timeit(fn, n=400, warm=40):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
  cap = 1_000_000
  g = torch.Generator(device=DEV).manual_seed(1)
  ds = TensorTupleDataset(size=cap, device=DEV)
  for lo in range(0, cap, 1 << 18):
    k = min(1 << 18, cap - lo)
    ds.extend(torch.rand(k, 3, device=DEV, generator=g), torch.rand(k, 1, device=DEV, generator=g),
              torch.rand(k, 1, device=DEV, generator=g), torch.rand(k, 3, device=DEV, generator=g),
              (torch.rand(k, 1, device=DEV, generator=g) < 0.01).float())
  ctl = DDPGController(3, 1, -2.0, 2.0, device=DEV)
  batch = ds.sample_device(B)
  rows = [('randperm_prefix (host only)', lambda: randperm_prefix(cap, B)),
          ('learn only', lambda: ctl.learn(*batch)),
          ('sample_device', lambda: ds.sample_device(B)),
          ('sample_device + learn', lambda: ctl.learn(*ds.sample_device(B)))]
  for zc in ('1', '0'):
    def samp(zc=zc):
      from torchrl_amd import config
      config.index_zero_copy = zc == '1'
      return ds.sample(B)
    rows.append(('sample zero_copy=%s' % zc, samp))
    rows.append(('sample zero_copy=%s + learn' % zc, lambda samp=samp: ctl.learn(*samp())))
  for name, fn in rows:
    wall, host = timeit(fn)
    print('B=%d %-32s %7.1f us/step wall, host issue %7.1f us' % (B, name, wall, host), flush=True)


if __name__ == '__main__':
  main()
