import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loss_oracle
from torchrl_amd.contrib.controllers import PPOController
DEV = 'cuda:0'
def run(obs_size, A, hidden, B):
  torch.manual_seed(obs_size + 10 * A + B)
  ctl = PPOController(obs_size, A, lr=3e-4, alpha=0.5, beta=1e-2, clip_ratio=0.2, max_grad_norm=1e9, device=DEV, hidden=hidden)
  ref = copy.deepcopy(ctl.ac_net).cpu()
  ref_opt = torch.optim.Adam(ref.parameters(), lr=3e-4)
  g = torch.Generator().manual_seed(7 + B)
  obs = torch.randn(B, obs_size, generator=g); act = torch.randn(B, A, generator=g)
  ret = torch.randn(B, 1, generator=g); adv = torch.randn(B, 1, generator=g)
  with torch.no_grad():
    mean = ref.actor(obs)
    old = torch.distributions.Normal(mean, ref.log_std.exp().expand_as(mean)).log_prob(act) + 0.3 * torch.randn(B, A, generator=g)
  exp = loss_oracle.ppo_learn(ref, ref_opt, obs, act, None, None, None, ret, old, adv, clip=0.2, alpha=0.5, beta=1e-2, max_grad_norm=1e9)
  got = ctl.learn(obs.to(DEV), act.to(DEV), None, None, None, ret.to(DEV), old.to(DEV), adv.to(DEV))
  print('case', obs_size, A, hidden, B, {k: (float(exp[k]), float(got[k])) for k in exp})
  for (name, p), q in zip(ref.named_parameters(), ctl.ac_net.parameters()):
    e, v = p.grad.numpy().reshape(-1), q.grad.cpu().numpy().reshape(-1)
    d = np.abs(e - v)
    bad = d > 1e-4 * np.abs(e) + 2e-5 * np.abs(e).max()
    print('  %-18s n=%5d max|e|=%.3e maxdiff=%.3e bad=%d %s' % (name, e.size, np.abs(e).max(), d.max(), bad.sum(), np.nonzero(bad)[0][:12]))
for c in [(4, 2, (64, 64), 9000), (17, 2, (48, 33), 5000), (40, 1, (64, 64), 3000), (64, 2, (64, 64), 2048), (3, 1, (64,), 8192), (5, 4, (64,), 7001), (33, 3, (20,), 300), (3, 1, (64, 64), 256)]:
  run(*c)
