#!/usr/bin/env python3
"""Sweep the persistent wide-gather grid (TRL_GATHER_WG_PER_CU) on Atari-shaped records (diagnostic)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TRL_LIB', 'tune')   # the -DTRL_TUNING diagnostic build (python -m torchrl_amd._build --tuning)
import bench  # noqa: E402


def main():
  dev = torch.device('cuda:0')
  for un, wg in ((4, 0), (2, 0), (4, 2), (4, 8), (2, 16)):
    os.environ['TRL_GATHER_UN'] = str(un)
    os.environ["TRL_GATHER_WG_PER_CU"] = str(wg)   # 0 = the library default
    for B in (1024, 256):
      r = bench.gather_roofline(dev, 50, 200_000, B, atari=True)
      print('un=%d wg_per_cu=%-8d B=%-5d %6.2f us  %6.0f GB/s' % (un, wg, B, r['us_per_launch'], r['achieved']), flush=True)
  os.environ.pop('TRL_GATHER_WG_PER_CU')


if __name__ == '__main__':
  main()
