#!/usr/bin/env python3
"""A few Double-DQN steps on Atari-shaped uint8 frames (B = 256, 28,224 -> 128 -> 128 -> 6), for rocprofv3 --pmc
passes over the uint8 first layer's kernels (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import DQNController  # noqa: E402

dev = torch.device('cuda:0')
B, K = 256, 28224
ctl = DQNController(K, 6, double_dqn=True, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
batch = [torch.randint(0, 256, (B, K), dtype=torch.uint8, device=dev, generator=g), torch.randint(0, 6, (B, 1), device=dev, generator=g).float(),
         torch.randn(B, 1, device=dev, generator=g), torch.randint(0, 256, (B, K), dtype=torch.uint8, device=dev, generator=g),
         torch.zeros(B, 1, device=dev)]
for _ in range(6):
  ctl._fused(*batch, 0.99, ctl.q_net_optim, double_dqn=True, update=False)
torch.cuda.synchronize()
print('done')
