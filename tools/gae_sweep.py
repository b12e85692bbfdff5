#!/usr/bin/env python3
"""Sweep / ablate the pipelined GAE kernel on the GPU box (diagnostic, not a test).

    python tools/gae_sweep.py [T N] [--warm]

Timing: launches are captured into one HIP graph and rotate over enough independent buffer
sets that every launch finds its inputs cold in the 256 MiB Infinity Cache (--warm reuses one
set, i.e. cache-resident inputs).  TRL_GAE_TUNE="E,PF,dbg" selects envs per workgroup,
prefetch depth and ablation mask (1 no chain, 2 no stores, 4 no staging, 8 no DMA: wrong
results, timing only).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import functional as F  # noqa: E402


def graph_time_us(launches, reps=5):
  """launches: list of callables run once each per replay; returns us per launch."""
  for fn in launches:
    fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(reps):
      for fn in launches:
        fn()
  g.replay()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  g.replay()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) * 1e3 / (reps * len(launches))


def main():
  args = [a for a in sys.argv[1:] if not a.startswith('--')]
  warm = '--warm' in sys.argv
  T = int(args[0]) if len(args) > 0 else 2048
  N = int(args[1]) if len(args) > 1 else 4096
  dev = torch.device('cuda:0')
  algo = 16 * T * N
  nsets = 1 if warm else max(2, int(3 * (256 << 20) / algo) + 1)
  sets = []
  for _ in range(nsets):
    sets.append((torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.randn(N, device=dev),
                 (torch.empty(T, N, device=dev), torch.empty(T, N, device=dev))))
  print('T=%d N=%d  %d buffer sets (%s)' % (T, N, nsets, 'warm' if warm else 'cold'))
  cp = [(lambda s=s: s[3][0].copy_(s[0])) for s in sets]
  us = graph_time_us(cp)
  print('torch copy r -> ret (8 B/elem): %.1f us  %.0f GB/s' % (us, 8 * T * N / us / 1e3))
  configs = [(16, PF, d, 2) for PF in (2, 3) for d in (0, 1, 2, 3, 8, 9, 10, 11)] + [(32, 2, 0, 2), (32, 3, 0, 2)]
  configs += [(16, PF, d, 1) for PF in (2, 3, 4) for d in (0, 1, 11)]
  for E, PF, dbg, U in configs:
    os.environ['TRL_GAE_TUNE'] = '%d,%d,%d,%d' % (E, PF, dbg, U)
    try:
      us = graph_time_us([(lambda s=s: F.gae(s[0], s[1], s[2], 0.99, 0.95, out=s[3])) for s in sets])
    except Exception as e:  # noqa
      print('E=%2d PF=%d dbg=%2d  failed: %s' % (E, PF, dbg, str(e)[:80]))
      continue
    print('E=%2d PF=%d dbg=%2d U=%d  %7.1f us %6.0f GB/s' % (E, PF, dbg, U, us, algo / us / 1e3), flush=True)
  os.environ.pop('TRL_GAE_TUNE')


if __name__ == '__main__':
  main()
