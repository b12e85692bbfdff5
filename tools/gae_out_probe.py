#!/usr/bin/env python3
"""GAE kernel time, cold inputs, by output layout ([T, N] vs env-major [N, T]) and with / without the
advantage plane (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
bench._load()
from torchrl_amd import functional as F  # noqa: E402

T, N = 2048, 4096
dev = torch.device('cuda:0')
nsets = 6
sets = [(torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.randn(N, device=dev),
         (torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)),
         (torch.empty(N, T, device=dev), torch.empty(N, T, device=dev))) for _ in range(nsets)]
rot = {'i': 0}


def make(em, adv=True):
  def launch():
    s = sets[rot['i'] % nsets]
    rot['i'] += 1
    o = s[4] if em else s[3]
    F.gae(s[0], s[1], s[2], 0.99, 0.95, env_major_out=em, out=o if adv else (o[0], None), want_adv=adv)
  return launch


for rep in range(2):
  for em in (False, True):
    for adv in (True, False):
      us = bench.kernel_time_us(make(em, adv), 48)
      nbytes = (16 if adv else 12) * T * N
      print('env_major_out=%-5s adv=%-5s %6.2f us  %5.0f GB/s' % (em, adv, us, nbytes / us / 1e3), flush=True)
