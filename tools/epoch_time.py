"""Time PPOController.learn_epoch (one persistent launch over K minibatches) against K learn() calls.
  python tools/epoch_time.py [hidden] [minibatch] [K] [obs] [actions]"""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

hidden = tuple(int(h) for h in (sys.argv[1] if len(sys.argv) > 1 else '64,64').split(','))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
OBS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ACT = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device('cuda:0')
torch.manual_seed(0)
ctl = PPOController(OBS, ACT, lr=3e-4, alpha=0.5, beta=1e-3, clip_ratio=0.2, max_grad_norm=1.0, device=dev, hidden=hidden)
n = B * K
g = torch.Generator(device=dev).manual_seed(1)
cols = [torch.randn(n, OBS, device=dev, generator=g), torch.randn(n, ACT, device=dev, generator=g), None, None, None,
        torch.randn(n, 1, device=dev, generator=g), torch.randn(n, ACT, device=dev, generator=g) - 1.0,
        torch.randn(n, 1, device=dev, generator=g)]
for _ in range(2):
  ctl.learn_epoch(cols, B, 1)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
info = ctl.learn_epoch(cols, B, 1)
b.record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
us = a.elapsed_time(b) * 1e3 / K
print('learn_epoch hidden=%s B=%d K=%d: %.3f us per minibatch (device), %.2f M transitions/s; wall %.3f us per minibatch; '
      'losses finite: %s' % (hidden, B, K, us, B / us, wall * 1e6 / K, bool(torch.isfinite(info['actor_loss']).all())))
ncalls = min(K, 2000)
for i in range(50):
  ctl.learn(*[c[i * B:(i + 1) * B] if c is not None else None for c in cols])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(ncalls):
  ctl.learn(*[c[i * B:(i + 1) * B] if c is not None else None for c in cols])
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / ncalls * 1e6
print('learn() x %d: %.2f us per call, %.2f M transitions/s  -> learn_epoch is %.1fx' % (ncalls, per, B / per, per / (wall * 1e6 / K)))
