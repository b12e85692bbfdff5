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
os.environ.setdefault('TRL_LIB', 'tune')   # the -DTRL_TUNING diagnostic build (python -m torchrl_amd._build --tuning)
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
  if '--blocked' in sys.argv:
    configs = [(16, 2, 0, 1), (16, 2, 16, 1), (16, 2, 32, 1), (16, 2, 48, 1)]
  for E, PF, dbg, U in configs:
    os.environ['TRL_GAE_TUNE'] = '%d,%d,%d,%d' % (E, PF, dbg, U)
    try:
      us = graph_time_us([(lambda s=s: F.gae(s[0], s[1], s[2], 0.99, 0.95, out=s[3])) for s in sets])
    except Exception as e:  # noqa
      print('E=%2d PF=%d dbg=%2d  failed: %s' % (E, PF, dbg, str(e)[:80]))
      continue
    print('E=%2d PF=%d dbg=%2d U=%d  %7.1f us %6.0f GB/s' % (E, PF, dbg, U, us, algo / us / 1e3), flush=True)
  os.environ.pop('TRL_GAE_TUNE')
  # ---- blocked layout [N / E][T][E]: the same kernel on flat per-group streams
  only_blocked = '--blocked' in sys.argv
  for E in (16, 8):
    for pad in (0, 4, 20, 68):       # rows of padding per env group (group stride = (T + pad) * E floats)
      bsets = []
      for s in sets:
        planes = []
        for x in s[:2]:
          buf = torch.empty(N // E, T + pad, E, device=dev)
          buf[:, :T] = F.to_blocked(x, E)
          planes.append(buf[:, :T])
        bsets.append((planes[0], planes[1], s[2], (torch.empty(N // E, T, E, device=dev), torch.empty(N // E, T, E, device=dev))))
      cfgs = [(2, 0, 1), (3, 0, 1), (4, 0, 1), (3, 16, 1), (3, 32, 1), (3, 48, 1), (4, 48, 1), (3, 0, 2), (3, 48, 2)]
      if pad == 0:
        cfgs += [(3, 1, 1), (3, 2, 1), (3, 3, 1), (3, 9, 1), (3, 11, 1), (3, 15, 1), (3, 17, 1), (3, 49, 1)]
      for PF, dbg, U in cfgs:
        if E == 8 and U == 2:
          continue
        os.environ['TRL_GAE_TUNE'] = '0,%d,%d,%d' % (PF, dbg, U)
        try:
          us = graph_time_us([(lambda s=s: F.gae_blocked(s[0], s[1], s[2], 0.99, 0.95, out=s[3])) for s in bsets])
        except Exception as e:  # noqa
          print('blocked E=%2d pad=%d PF=%d dbg=%2d U=%d  failed: %s' % (E, pad, PF, dbg, U, str(e)[:80]))
          continue
        print('blocked E=%2d pad=%2d PF=%d dbg=%2d U=%d  %7.1f us %6.0f GB/s' % (E, pad, PF, dbg, U, us, algo / us / 1e3), flush=True)
      os.environ.pop('TRL_GAE_TUNE')
      if pad == 0:
        nt = [(s[0], s[1], s[2], (torch.empty(N, T, device=dev), torch.empty(N, T, device=dev))) for s in bsets]
        us = graph_time_us([(lambda s=s: F.gae_blocked(s[0], s[1], s[2], 0.99, 0.95, env_major_out=True, out=s[3])) for s in nt])
        print('blocked E=%2d in, env-major [N, T] out (defaults)  %7.1f us %6.0f GB/s' % (E, us, algo / us / 1e3), flush=True)
        del nt
      del bsets


if __name__ == '__main__':
  main()
