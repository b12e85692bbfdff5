#!/usr/bin/env python3
"""Probe how PyTorch-ROCm handles the skinny weight-gradient GEMMs of the small MLPs
(dW[out, in] = dY[B, out]^T X[B, in] with B = 65536): default library choice vs rocBLAS vs a
split-K formulation with a batched matmul."""
import sys
import torch


def t_us(fn, iters=50):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(iters):
      fn()
  g.replay(); torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record(); g.replay(); b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) * 1e3 / iters


def main():
  dev = 'cuda:0'
  for B in (65536, 4096, 256):
    for (i, o) in ((3, 64), (64, 64), (64, 1), (256, 256), (4, 128)):
      X = torch.randn(B, i, device=dev)
      dY = torch.randn(B, o, device=dev)
      W = torch.randn(o, i, device=dev)
      res = {}
      res['fwd addmm'] = t_us(lambda: torch.addmm(W[:, 0], X, W.t()))
      res['dX mm'] = t_us(lambda: dY @ W)
      res['dW mm'] = t_us(lambda: dY.t() @ X)
      for S in (16, 64, 256):
        if B % S == 0 and B // S >= 16:
          res['dW bmm S=%d' % S] = t_us(lambda: torch.bmm(dY.view(S, B // S, o).transpose(1, 2),
                                                          X.view(S, B // S, i)).sum(0))
      res['db sum'] = t_us(lambda: dY.sum(0))
      print('B=%6d in=%3d out=%3d  ' % (B, i, o) + '  '.join('%s %.1f' % kv for kv in res.items()), flush=True)
  for lib in ('hipblas', 'hipblaslt'):
    try:
      torch.backends.cuda.preferred_blas_library(lib)
      X = torch.randn(65536, 64, device=dev); dY = torch.randn(65536, 64, device=dev)
      print(lib, 'dW 64x64 B=65536: %.1f us' % t_us(lambda: dY.t() @ X),
            ' dW 3: %.1f' % t_us(lambda: dY.t() @ X[:, :3].contiguous()))
    except Exception as e:  # noqa
      print(lib, 'failed', e)


if __name__ == '__main__':
  main()
