"""Timeline of k_ac_learn_pair from a DIAGNOSTIC build (TRL_LEARN_STAMP=1 python -m torchrl_amd._build --tuning
--force): absolute shader-clock stamps of strips 8..11 of the four pairs of workgroup 0 (critic), wave A and B."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TRL_LIB', 'tune')
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
dev = torch.device('cuda', 0)
torch.manual_seed(0)
ctl = PPOController(3, 1, device=dev, hidden=(64, 64))
obs, act = torch.randn(B, 3, device=dev), torch.randn(B, 1, device=dev)
ret, adv, old = torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev) - 1
f = ctl._fused
for _ in range(20):
  f(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
torch.cuda.synchronize()
ws = f._ws.buf.view(torch.uint8)
pstride = (4481 + 8 + 3) // 4 * 4
off = 2 * 128 * pstride * 4 + 2 * 128 * 8 * 8 * 3 * 8
tr = ws[off:off + 2 * 4 * 4 * 2 * 8 * 8].cpu().numpy().view(np.uint64).reshape(2, 4, 4, 2, 8).astype(np.int64)
names = {0: ['top', 'x in, fetch issued', 'forward MFMAs issued', 'loss + last bwd done', 'B released tiles', 'tiles posted', 'db1 done', 'g2 tile written'],
         1: ['top', 'tiles seen', 'tiles read + released', 'dW1 MFMAs issued', 'dX1 MFMAs issued', 'mask + g1 tile', 'dW0 done, posted', '-']}
for pair in range(int(os.environ.get('PAIRS', 1))):
  t0 = tr[0, pair, 0, 0, 0]
  ev = []
  for it in range(4):
    for role in range(2):
      for slot in range(8 if role == 0 else 7):
        ev.append((tr[0, pair, it, role, slot] - t0, 'AB'[role], it + 8, names[role][slot]))
  ev.sort()
  print('pair', pair)
  prev = {'A': None, 'B': None}
  for t, r, it, name in ev:
    d = '' if prev[r] is None else '+%d' % (t - prev[r])
    prev[r] = t
    print('%8d  %s%s strip %2d  %-28s %s' % (t, '' if r == 'A' else '                                        ', r, it, name, d))
