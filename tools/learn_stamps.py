"""In-kernel timeline of k_ac_learn from a DIAGNOSTIC build (TRL_LEARN_STAMP=1 python -m
torchrl_amd._build --tuning --force): per wave, shader-clock and 100 MHz stamps at entry, after the weight
staging, after the strip loop and at the end.  Prints the medians and the in-kernel clock."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TRL_LIB', 'tune')   # the -DTRL_TUNING diagnostic build (python -m torchrl_amd._build --tuning)
from torchrl_amd import functional as F  # noqa: E402
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
NW = 4     # waves per workgroup (k_ac_learn; run with TRL_LEARN_NW=4: the pair kernel has its own trace tool)
os.environ.setdefault('TRL_LEARN_NW', '4')
dev = torch.device('cuda', 0)
torch.manual_seed(0)
HIDDEN = tuple(int(x) for x in os.environ.get('HIDDEN', '64,64').split(','))
ctl = PPOController(3, 1, device=dev, hidden=HIDDEN if len(HIDDEN) > 1 else HIDDEN[0])
obs, act = torch.randn(B, 3, device=dev), torch.randn(B, 1, device=dev)
ret, adv, old = torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev) - 1
f = ctl._fused
for _ in range(int(os.environ.get('ITERS', 3000))):          # >= 2 s of back-to-back launches
  f(obs, act, ret, old, adv, 0.2, 0.5, 1e-3)
torch.cuda.synchronize()
ws = f._ws.buf.view(torch.uint8)
np_c = sum(p.numel() for p in ctl.ac_net.critic.parameters())
np_a = sum(p.numel() for p in ctl.ac_net.actor.parameters())
pstride = (max(np_c, np_a) + 8 + 3) // 4 * 4
nwg = min(128, (B + 32 * NW - 1) // (32 * NW))
raw = ws[2 * 128 * pstride * 4:2 * 128 * pstride * 4 + 2 * nwg * NW * 64].cpu().numpy().view(np.uint64)
st = raw.reshape(2 * nwg * NW, 8).astype(np.float64)
clk, rt = st[:, 0::2], st[:, 1::2]
for i, name in enumerate(('staging', 'strip loop', 'partials out')):
  dc, dr = clk[:, i + 1] - clk[:, i], rt[:, i + 1] - rt[:, i]
  print('%-13s median %8.0f shader cycles  %6.2f us' % (name, np.median(dc), np.median(dr) / 100.0))
for net, name in enumerate(('critic', 'actor')):      # rows [net * nwg * 4, (net + 1) * nwg * 4) are that network's waves
  sl = slice(net * nwg * NW, (net + 1) * nwg * NW)
  dr = rt[sl, 2] - rt[sl, 1]
  print('strip loop, %-6s waves: median %7.2f us  max %7.2f us' % (name, np.median(dr) / 100.0, dr.max() / 100.0))
dc, dr = clk[:, 3] - clk[:, 0], rt[:, 3] - rt[:, 0]
print('whole wave    median %8.0f shader cycles  %6.2f us; clock %.2f GHz' % (np.median(dc), np.median(dr) / 100.0,
                                                                             np.median(dc / dr) * 0.1))
print('first entry -> last exit over all waves: %.2f us' % ((rt[:, 3].max() - rt[:, 0].min()) / 100.0))
print('entry spread: %.2f us' % ((rt[:, 0].max() - rt[:, 0].min()) / 100.0))
dc, dr = clk[:, 2] - clk[:, 1], rt[:, 2] - rt[:, 1]
print('loop clock %.2f GHz' % (np.median(dc / dr) * 0.1))
off = 2 * 128 * pstride * 4 + 2 * 128 * 8 * 64
ph = ws[off:off + 2 * nwg * NW * 128].cpu().numpy().view(np.uint64)
ph = ph.reshape(2 * nwg * NW, 16).astype(np.float64)
nst = (B + 31) // 32 / (nwg * NW)
names = {5: 'loop top -> inputs in registers', 6: 'x tile write + fence', 7: 'layer 0 forward + relu', 8: 'layer 1 forward + relu',
         0: 'last layer forward', 1: 'loss + last layer backward', 2: 'layer 1 backward (tiles, dW1, g0, mask)',
         9: 'layer 0 backward: db0, tile, fence', 3: 'layer 0 backward: dW0 MFMAs'}
if len(HIDDEN) == 1 and HIDDEN[0] > 64:       # k_ac_learn_wide
  names = {5: 'loop top -> inputs in registers', 6: 'x tile write, fence, operand read', 7: 'forward: layer 0, relu, last layer dots',
           0: 'last layer finish', 1: 'loss, gout rows write, fence, reads', 3: 'backward (lane = feature)', 8: '-', 2: '-', 9: '-'}
for k in (5, 6, 7, 8, 0, 1, 2, 9, 3):
  print('  phase %-42s %8.0f cycles per strip' % (names[k], np.median(ph[:, k]) / nst))
