"""Ablations of the pipelined GAE kernel in the blocked layout (diagnostic build; timing only)."""
import os, sys, torch
os.environ.setdefault('TRL_LIB', 'tune')   # the -DTRL_TUNING diagnostic build (python -m torchrl_amd._build --tuning)
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from torchrl_amd import functional as F
from gae_sweep import graph_time_us
T, N = 2048, 4096
dev = torch.device('cuda:0')
algo = 16 * T * N
nsets = 7
for E in (16, 8):
  sets = [(F.to_blocked(torch.randn(T, N, device=dev), E), F.to_blocked(torch.randn(T, N, device=dev), E),
           torch.randn(N, device=dev), (torch.empty(N // E, T, E, device=dev), torch.empty(N // E, T, E, device=dev)))
          for _ in range(nsets)]
  for PF, U in ((3, 1),):
    for dbg, ck in ((48, 0), (48, 1), (49, 1), (14, 0), (14, 1), (6, 1), (50, 1)):
      os.environ['TRL_GAE_TUNE'] = '0,%d,%d,%d,%d' % (PF, dbg, U, ck)
      us = graph_time_us([(lambda s=s: F.gae_blocked(s[0], s[1], s[2], 0.99, 0.95, out=s[3])) for s in sets])
      print('blocked E=%2d PF=%d U=%d ckpt=%d dbg=%2d [%s]  %6.1f us' % (
          E, PF, U, ck, dbg, ' '.join(n for b, n in ((1, 'nochain'), (2, 'nostore'), (4, 'nostage'), (8, 'nodma'), (16, 'ntst'), (32, 'ntld')) if dbg & b), us), flush=True)
    for dbg, ck in ((48 + 256 + 2, 0), (48 + 256 + 2, 1), (14 + 256, 1)):   # stamps (stores off: the stamps go into adv)
      os.environ['TRL_GAE_TUNE'] = '0,%d,%d,%d,%d' % (PF, dbg, U, ck)
      s = sets[0]
      F.gae_blocked(s[0], s[1], s[2], 0.99, 0.95, out=s[3])
      torch.cuda.synchronize()
      st = s[3][1].reshape(-1)[:4 * (N // E)].reshape(-1, 4).median(0).values.tolist()
      print('  stamps ckpt=%d dbg=%d: scan %.0f cycles (%.1f / step), wave life %.0f, until first tile %.0f, first tile -> end %.0f'
            % (ck, dbg, st[0], st[0] / T, st[1], st[2], st[3]))
