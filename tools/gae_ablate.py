import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from torchrl_amd import functional as F
from gae_sweep import time_us
T, N = 2048, 4096
dev = torch.device('cuda:0')
r, v, b = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.randn(N, device=dev)
out = (torch.empty(T, N, device=dev), torch.empty(T, N, device=dev))
for E, PF in ((16, 2), (16, 3)):
  for dbg in (0, 14, 6, 10, 12, 7, 11, 13, 1, 15):
    os.environ['TRL_GAE_TUNE'] = '%d,%d,%d' % (E, PF, dbg)
    us = time_us(lambda: F.gae(r, v, b, 0.99, 0.95, out=out), 50)
    print('E=%d PF=%d dbg=%2d %7.1f us' % (E, PF, dbg, us), flush=True)
# empty kernel launch floor
x = torch.zeros(1, device=dev)
print('tiny torch kernel back-to-back: %.1f us' % time_us(lambda: x.add_(1), 200))
