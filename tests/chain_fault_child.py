"""Child process of tests/test_gpu_offpolicy.py::test_ddpg_chain_barrier_timeout_is_reported (not collected by pytest).

Runs on the -DTRL_TUNING diagnostic build (TRL_LIB=tune) with TRL_CHAIN_WITHHOLD=3 (the third one-launch DDPG step on the
stream loses a workgroup, so its first device-wide barrier can never complete) and a short TRL_CHAIN_SPIN.  Checks:
the lost step writes NaN losses and applies no update; the NEXT call returns an error that names the barrier; the
steps after it are right again (bit for bit the launch-per-phase chain's)."""
import sys

import numpy as np
import torch

from torchrl_amd import _lib
from torchrl_amd.contrib.controllers import DDPGController

DEV = 'cuda:0'
S, A, hidden, B = 3, 1, 64, 128


def batch(rs):
  f = lambda a: torch.from_numpy(a.astype(np.float32)).to(DEV)
  return [f(rs.randn(B, S)), f(rs.uniform(-1, 1, size=(B, A))), f(rs.uniform(-2, 0, size=(B, 1))), f(rs.randn(B, S)),
          f((rs.rand(B, 1) < 0.1))]


def make():
  torch.manual_seed(5)
  return DDPGController(S, A, -1.0, 1.0, actor_lr=1e-3, critic_lr=1e-3, tau=0.05, device=DEV, hidden=hidden)


def state(c):
  return [t.data.clone() for t in (c._actor, c._critic, c._target_actor, c._target_critic)] + \
         [c.actor_optim.exp_avg.clone(), c.critic_optim.exp_avg_sq.clone()]


def main():
  assert _lib.LIB_PATH.endswith('libtrl_hip_tune.so'), 'run with TRL_LIB=tune'
  rs = np.random.RandomState(3)
  batches = [batch(rs) for _ in range(7)]
  lib = _lib.lib()

  # the yardstick: one launch per phase, steps on batches 0, 1, 3, 4, 5 (batch 2 is the step that gets lost)
  lib.trl_offpolicy_set_chain(0)
  ref = make()
  ref_losses = [ref.learn(*batches[i]) for i in (0, 1)]
  torch.cuda.synchronize()
  ref_after2 = state(ref)
  for i in (3, 4, 5):
    ref_losses.append(ref.learn(*batches[i]))
  torch.cuda.synchronize()
  ref_end = state(ref)

  lib.trl_offpolicy_set_chain(1)
  ctl = make()
  for i in (0, 1):
    ctl.learn(*batches[i])
  lost = ctl.learn(*batches[2])        # chain launch number 3 of the stream: a workgroup never arrives
  torch.cuda.synchronize()
  assert all(np.isnan(float(v)) for v in lost.values()), {k: float(v) for k, v in lost.items()}
  for a, b in zip(state(ctl), ref_after2):
    assert torch.equal(a, b), 'the aborted step must not have applied any update'
  try:
    ctl.learn(*batches[3])
  except _lib.TrlError as e:
    msg = str(e)
    assert 'barrier' in msg and 'phase 0' in msg, msg
  else:
    raise AssertionError('the call after the lost step did not report it')
  # the caller's recovery: the optimizers counted the lost step and the refused call; put the counts back
  ctl.actor_optim.step_count = ctl.critic_optim.step_count = 2
  got = [ctl.learn(*batches[i]) for i in (3, 4, 5)]
  torch.cuda.synchronize()
  for g, e in zip(got, ref_losses[2:]):
    for k in e:
      assert float(g[k]) == float(e[k]), (k, float(g[k]), float(e[k]))
  for a, b in zip(state(ctl), ref_end):
    assert torch.equal(a, b), 'steps after the reported time-out must be right again'
  print('chain fault child ok')


if __name__ == '__main__':
  main()
  sys.exit(0)
