#!/usr/bin/env python3
"""Sweep / ablate the pipelined GAE kernel on the GPU box (diagnostic, not a test).

    python tools/gae_sweep.py [T N]

For each (E envs per workgroup, PF prefetch depth, ablation mask) sets TRL_GAE_TUNE and
times back-to-back launches with events.  Ablations (wrong results, timing only):
1 no chain, 2 no output stores, 4 no staging pass, 8 no DMA.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import functional as F  # noqa: E402


def time_us(fn, iters=30):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) * 1e3 / iters


def main():
  T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
  N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
  dev = torch.device('cuda:0')
  r, v, b = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.randn(N, device=dev)
  out = (torch.empty(T, N, device=dev), torch.empty(T, N, device=dev))
  out_nt = (torch.empty(N, T, device=dev), torch.empty(N, T, device=dev))
  algo = 16 * T * N
  src, dst = torch.empty(algo // 8, device=dev), torch.empty(algo // 8, device=dev)
  us = time_us(lambda: dst.copy_(src))
  print('torch copy of the same bytes: %.1f us  %.0f GB/s' % (us, algo / us / 1e3))
  for E in (16, 32, 64):
    for PF in (2, 3, 4):
      for dbg in (0, 1, 2, 4, 8, 15):
        os.environ['TRL_GAE_TUNE'] = '%d,%d,%d' % (E, PF, dbg)
        us = time_us(lambda: F.gae(r, v, b, 0.99, 0.95, out=out))
        line = 'E=%2d PF=%d dbg=%2d  TN %7.1f us %6.0f GB/s' % (E, PF, dbg, us, algo / us / 1e3)
        if dbg == 0:
          us2 = time_us(lambda: F.gae(r, v, b, 0.99, 0.95, out=out_nt, env_major_out=True))
          line += '   NT %7.1f us %6.0f GB/s' % (us2, algo / us2 / 1e3)
        print(line, flush=True)
  os.environ.pop('TRL_GAE_TUNE')


if __name__ == '__main__':
  main()
