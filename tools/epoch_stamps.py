"""Phase timeline of one update inside k_ppo_epoch (diagnostic build: TRL_LIB=tune).  python tools/epoch_stamps.py [hidden] [B]"""
import ctypes as C
import os
import sys

os.environ['TRL_LIB'] = 'tune'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torchrl_amd import _lib  # noqa: E402
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

hidden = tuple(int(h) for h in (sys.argv[1] if len(sys.argv) > 1 else '64,64').split(','))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = 512
dev = torch.device('cuda:0')
torch.manual_seed(0)
ctl = PPOController(3, 1, lr=3e-4, beta=1e-3, device=dev, hidden=hidden)
n = B * K
cols = [torch.randn(n, 3, device=dev), torch.randn(n, 1, device=dev), None, None, None, torch.randn(n, 1, device=dev),
        torch.randn(n, 1, device=dev) - 1.0, torch.randn(n, 1, device=dev)]
for _ in range(2):
  ctl.learn_epoch(cols, B, 1)
torch.cuda.synchronize()
out = (C.c_ulonglong * 256)()
lib = _lib.lib()
lib.trl_tune_epoch_stamps.argtypes = [C.c_void_p]
assert lib.trl_tune_epoch_stamps(out) == 0
names = ['layer 0 (+ barrier)', 'layer 1 + last-layer dots (+ barrier)', 'loss, wave 0 (+ barrier)', 'lane = feature pass (+ barrier)', 'dW1',
         'g1 (+ barrier)', "dW0'", 'input swap', 'gradient / partial / loss-sum flush (+ barrier)', '-', '-',
         'clip + Adam: small vectors, layer 0, W1 tiles (one stream)', 'barrier']
for role in range(2):
  st = [[out[(role * 8 + w) * 16 + j] for j in range(14)] for w in range(8)]
  print('critic' if role == 0 else 'actor', 'total cycles (s_memtime ticks), wave 0:', st[0][13] - st[0][0])
  print('  %-48s %s' % ('phase \\ wave', ' '.join('%6d' % w for w in range(8))))
  for j in range(13):
    print('  %-48s %s' % (names[j], ' '.join('%6d' % (st[w][j + 1] - st[w][j]) for w in range(8))))
