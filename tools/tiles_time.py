"""learn() at 1 M rows on shapes outside the register-chained kernels: the tile pipeline (trl_ppo_learn_tiles) against the
layer-by-layer path the same controller takes without it.  python tools/tiles_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

dev = torch.device('cuda:0')
n = 1 << 20
for obs_size, A, hidden in [(12, 3, (256,)), (8, 4, (128,)), (15, 1, (64, 64)), (10, 2, (256,)), (3, 1, (64, 64)), (3, 1, (256,))]:
  torch.manual_seed(0)
  ctl = PPOController(obs_size, A, lr=3e-4, beta=1e-3, device=dev, hidden=hidden)
  g = torch.Generator(device=dev).manual_seed(1)
  cols = [torch.randn(n, obs_size, device=dev, generator=g), torch.randn(n, A, device=dev, generator=g), None, None, None,
          torch.randn(n, 1, device=dev, generator=g), torch.randn(n, A, device=dev, generator=g) - 1.0,
          torch.randn(n, 1, device=dev, generator=g)]
  res = {}
  for mode in ('tiles', 'default'):
    if mode == 'tiles':
      fn = lambda: ctl._epoch.learn_tiles(cols[0], cols[1], cols[5], cols[6], cols[7], 0.2, 0.5, 1e-3, ctl.ac_net_optim)
    else:
      saved = ctl._epoch
      if ctl._fused is None:
        ctl._epoch = None        # -> layer by layer
      fn = lambda: ctl.learn(*cols)
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
      fn()
    torch.cuda.synchronize()
    res[mode] = (time.perf_counter() - t0) / 10 * 1e6
    if mode == 'default':
      ctl._epoch = saved
  print('obs %2d  A %d  hidden %-9s 1M rows: tiles %8.1f us   %s %8.1f us' % (
      obs_size, A, hidden, res['tiles'], 'register-chained kernel' if ctl._fused is not None else 'layer by layer        ', res['default']))
