#!/usr/bin/env python3
"""One learn() over a 1 M-row batch for the shapes outside the register-chained whole-step kernels: the tile pipeline
spread over the chip (trl_ppo_learn_tiles / trl_a2c_learn_tiles; the wide form beyond 15 observations / four actions)
against the layer-by-layer path (fused=False) on the same controller shapes (diagnostic)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import PPOController, A2CController  # noqa: E402

dev = 'cuda:0'
B = 1 << 20


def timed(fn):
  for _ in range(3):
    info = fn()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(10):
    info = fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / 10 * 1e3, {k: round(float(v), 5) for k, v in info.items()}


CASES = (('ppo', 8, 4, (256,)), ('ppo', 15, 4, (256,)), ('ppo', 12, 3, (256,)), ('ppo', 8, 4, (128,)),
          ('ppo', 17, 6, (256,)), ('ppo', 24, 4, (256,)), ('ppo', 31, 6, (256,)), ('ppo', 17, 6, (64,)),
                          ('a2c', 8, 4, (256,)), ('a2c', 17, 6, (256,)), ('a2c', 31, 6, (256,)))
if len(sys.argv) >= 5:      # one shape: kind obs actions hidden   (e.g. under rocprofv3 --kernel-trace --stats)
  CASES = ((sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), tuple(int(h) for h in sys.argv[4].split(','))),)
for kind, obs, A, hid in CASES:
  o = torch.randn(B, obs, device=dev)
  r = torch.randn(B, 1, device=dev)
  line = '%s obs %2d actions %d hidden %s:' % (kind, obs, A, hid)
  for fused in (True, False):
    if kind == 'ppo':
      c = PPOController(obs, A, device=dev, hidden=hid, fused=fused)
      a, l, ad = torch.randn(B, A, device=dev), torch.randn(B, A, device=dev) - 1, torch.randn(B, 1, device=dev)
      fn = lambda: c.learn(o, a, None, None, None, r, l, ad)  # noqa: E731
    else:
      c = A2CController(obs, A, device=dev, hidden=hid, fused=fused)
      a = torch.randint(0, A, (B, 1), device=dev).float()
      fn = lambda: c.learn(o, a, None, None, None, r)  # noqa: E731
    path = 'layers' if not fused else ('whole-step' if c._fused is not None else ('tiles' if c._tiles is not None else 'layers'))
    try:
      ms, info = timed(fn)
      line += '  %s %.3f ms' % (path, ms)
    except Exception as e:  # noqa: BLE001
      line += '  %s error %s' % (path, str(e)[:120])
  print(line, flush=True)
