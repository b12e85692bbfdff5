"""Phase timeline of one 64-row tile inside k_ppo_tiles (slab 0's middle tile; diagnostic build: TRL_LIB=tune).
python tools/tiles_stamps.py [obs] [actions] [hidden] [rows]"""
import ctypes as C
import os
import sys

os.environ['TRL_LIB'] = 'tune'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torchrl_amd import _lib  # noqa: E402
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

obs_size = int(sys.argv[1]) if len(sys.argv) > 1 else 17
A = int(sys.argv[2]) if len(sys.argv) > 2 else 6
hidden = tuple(int(h) for h in (sys.argv[3] if len(sys.argv) > 3 else '256').split(','))
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
dev = torch.device('cuda:0')
torch.manual_seed(0)
ctl = PPOController(obs_size, A, lr=3e-4, beta=1e-3, device=dev, hidden=hidden)
assert ctl._fused is None and ctl._tiles is not None, 'a shape the tile pipeline serves'
cols = [torch.randn(B, obs_size, device=dev), torch.randn(B, A, device=dev), None, None, None, torch.randn(B, 1, device=dev),
        torch.randn(B, A, device=dev) - 1.0, torch.randn(B, 1, device=dev)]
for _ in range(3):
  ctl.learn(*cols)
torch.cuda.synchronize()
out = (C.c_ulonglong * 256)()
lib = _lib.lib()
lib.trl_tune_epoch_stamps.argtypes = [C.c_void_p]
assert lib.trl_tune_epoch_stamps(out) == 0
names = ['store x(t+1), prefetch x(t+2)', 'layer 0 + last-layer dots (+ barrier)', 'layer 1 (two hidden layers)', 'loss, wave 0 (+ barrier)',
         'lane = feature pass (+ barrier)', 'dW1', 'g1 (+ barrier)', "dW0'", 'input swap', 'barrier, top of the next tile']
# stamp j is taken BEFORE phase j: [0] top of the tile's products (behind the x stores), [1] after layer 0 + barrier, ...
for role in range(2):
  st = [[out[(role * 8 + w) * 16 + j] for j in range(10)] for w in range(8)]
  print('critic' if role == 0 else 'actor', 'one tile, s_memtime ticks, wave 0:', st[0][9] - st[0][0])
  print('  %-48s %s' % ('phase \\ wave', ' '.join('%6d' % w for w in range(8))))
  for j in range(9):
    print('  %-48s %s' % (names[j + 1], ' '.join('%6d' % (st[w][j + 1] - st[w][j]) for w in range(8))))
  # inside the layer-0 phase: [10] behind the tiles' products and epilogues, [11] behind the row sums of the last layer's dots
  sx = [[out[(role * 8 + w) * 16 + j] for j in (0, 10, 11, 1)] for w in range(8)]
  for j, nm in enumerate(('  layer 0: tiles (products, relu, stores, dot FMAs)', '  layer 0: row sums of the dots -> LDS', '  layer 0: barrier')):
    print('  %-48s %s' % (nm, ' '.join('%6d' % (sx[w][j + 1] - sx[w][j]) for w in range(8))))
