#!/usr/bin/env python3
"""An env-loop shaped run: push n_envs host transitions, sample + learn, repeat (DQN CartPole shapes)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import DQNController  # noqa: E402
from torchrl_amd.utils.storage import TransitionTupleDataset, Transition  # noqa: E402

DEV = torch.device('cuda:0')
n_envs, B = 16, 256
buf = TransitionTupleDataset(size=100_000, device=DEV)
ctl = DQNController(4, 2, double_dqn=True, device=DEV)
rs = np.random.RandomState(0)
tl = [Transition(rs.randn(4).astype(np.float32), int(rs.randint(2)), float(rs.randn()), rs.randn(4).astype(np.float32),
                 False) for _ in range(n_envs)]
for _ in range(300):
  buf.extend(tl)


def step():
  buf.extend(tl)
  ctl.learn(*buf.sample(B))


for _ in range(50):
  step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
  step()
torch.cuda.synchronize()
print('push(%d) + sample(%d) + learn: %.1f us per iteration' % (n_envs, B, (time.perf_counter() - t0) / 500 * 1e6))
