"""Time DDPGController.learn / DQNController.learn with and without the fused step (eager)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.contrib.controllers import DDPGController, DQNController  # noqa: E402

DEV = 'cuda:0'


def timeit(fn, n=300, warm=30):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
  rs = np.random.RandomState(0)
  for fused in ('1', '0'):
    from torchrl_amd import config
    config.fused_learn = fused == '1'
    ctl = DDPGController(3, 1, -2.0, 2.0, device=DEV)
    batch = [torch.from_numpy(rs.randn(B, k).astype(np.float32)).to(DEV) for k in (3, 1, 1, 3, 1)]
    wall, host = timeit(lambda: ctl.learn(*batch))
    print('ddpg  B=%d fused=%s: %.1f us/step wall, host issue %.1f us' % (B, fused, wall, host))
    q = DQNController(4, 2, double_dqn=True, device=DEV)
    qb = [torch.from_numpy(rs.randn(B, 4).astype(np.float32)).to(DEV),
          torch.from_numpy(rs.randint(0, 2, (B, 1)).astype(np.float32)).to(DEV),
          torch.from_numpy(rs.randn(B, 1).astype(np.float32)).to(DEV),
          torch.from_numpy(rs.randn(B, 4).astype(np.float32)).to(DEV),
          torch.zeros(B, 1, device=DEV)]
    wall, host = timeit(lambda: q.learn(*qb))
    print('ddqn  B=%d fused=%s: %.1f us/step wall, host issue %.1f us' % (B, fused, wall, host))


if __name__ == '__main__':
  main()
