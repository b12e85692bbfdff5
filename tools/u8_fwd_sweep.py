#!/usr/bin/env python3
"""Split count of the uint8 first-layer forward (TRL_U8_FWD_WGS = target workgroups per launch), three
passes of 256 x 28,224 -> 128 as in the Double-DQN step (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TRL_LIB', 'tune')   # the -DTRL_TUNING diagnostic build (python -m torchrl_amd._build --tuning)
import bench  # noqa: E402
bench._load()
from torchrl_amd import _lib  # noqa: E402
from torchrl_amd.contrib.controllers import DQNController  # noqa: E402

dev = torch.device('cuda:0')
B, K = 256, 28224
ctl = DQNController(K, 6, double_dqn=True, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
batch = [torch.randint(0, 256, (B, K), dtype=torch.uint8, device=dev, generator=g), torch.randint(0, 6, (B, 1), device=dev, generator=g).float(),
         torch.randn(B, 1, device=dev, generator=g), torch.randint(0, 256, (B, K), dtype=torch.uint8, device=dev, generator=g),
         torch.zeros(B, 1, device=dev)]
for wgs in (768, 512, 384, 256, 192, 1536):
  os.environ['TRL_U8_FWD_WGS'] = str(wgs)
  us = bench.kernel_time_us(lambda: ctl._fused(*batch, 0.99, ctl.q_net_optim, double_dqn=True, update=False), 30)
  print('target workgroups %5d: whole chain (gradients only) %7.1f us per step' % (wgs, us), flush=True)
