"""The no-allocation entry point trl_ppo_loss_static (include/trl_hip.h: the reduction scratch lives in a
file-scope device array of the library) against trl_ppo_loss with a caller-owned workspace: same kernels, so the
losses and the three backward seeds must be bit-identical -- on two calls in a row, the second on a smaller batch
(the array keeps the first call's partial sums), and again after other work has used the GPU.  Called through the
C ABI with raw device pointers, as a caller under graph capture would."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import loss_oracle
from torchrl_amd import _lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _inputs(B, A, seed):
  g = torch.Generator(device=DEV).manual_seed(seed)
  mean = torch.randn(B, A, device=DEV, generator=g)
  log_std = 0.3 * torch.randn(A, device=DEV, generator=g)
  value, adv, ret = (torch.randn(B, device=DEV, generator=g) for _ in range(3))
  action = mean + torch.randn(B, A, device=DEV, generator=g)
  old = torch.randn(B, A, device=DEV, generator=g) - 1.0
  return mean, log_std, value, action, old, adv, ret


def _call(static, x, B, A, ws=None):
  lib = _lib.lib()
  mean, log_std, value, action, old, adv, ret = x
  losses = torch.full((4,), float('nan'), device=DEV)
  dmean, dls, dv = torch.empty(B, A, device=DEV), torch.empty(A, device=DEV), torch.empty(B, device=DEV)
  st = _lib.stream_ptr(torch.device(DEV))
  args = [mean.data_ptr(), log_std.data_ptr(), value.data_ptr(), action.data_ptr(), old.data_ptr(), adv.data_ptr(),
          ret.data_ptr(), B, A, 0.2, 0.5, 1e-2, losses.data_ptr(), dmean.data_ptr(), dls.data_ptr(), dv.data_ptr()]
  if static:
    _lib.check(lib.trl_ppo_loss_static(*args, st))
  else:
    _lib.check(lib.trl_ppo_loss(*args, ws.data_ptr(), st))
  return losses, dmean, dls, dv


def test_static_scratch_entry_equals_workspace_entry_two_calls_in_a_row():
  ws = torch.empty(_lib.lib().trl_loss_workspace_bytes(), dtype=torch.uint8, device=DEV)
  # large (1024 blocks of partials), then smaller (20 blocks), then one block, then large again
  for i, (B, A) in enumerate([(300_000, 2), (5_000, 2), (200, 1), (70_001, 3)]):
    x = _inputs(B, A, 10 + i)
    got = _call(True, x, B, A)
    exp = _call(False, x, B, A, ws)
    torch.cuda.synchronize()
    for g_, e_, name in zip(got, exp, ('losses', 'dmean', 'dlog_std', 'dvalue')):
      assert torch.equal(g_, e_), (B, A, name)
    assert bool(torch.isfinite(got[0]).all())
    if B <= 5000:       # ... and both are the oracle's numbers (torch-CPU restatement of ppo_controller.py:55-68)
      mean, log_std, value, action, old, adv, ret = (t.cpu() for t in x)
      a_, c_, e_ = loss_oracle.ppo_losses(value.view(-1, 1), mean, log_std.view(1, -1), action, old, adv.view(-1, 1),
                                          ret.view(-1, 1), 0.2)
      for v, e in zip(got[0][:3].cpu().tolist(), (a_, c_, e_)):
        assert abs(v - float(e)) <= 1e-6 * abs(float(e)) + 2e-7, (B, A, v, float(e))


def test_static_scratch_entry_rejects_what_it_cannot_serve():
  lib = _lib.lib()
  x = _inputs(64, 1, 0)
  p = [t.data_ptr() for t in x]
  out = torch.empty(4, device=DEV)
  rc = lib.trl_ppo_loss_static(*p, 64, 65, 0.2, 0.5, 0.0, out.data_ptr(), None, None, None, None)
  assert rc == -1 and b'static scratch' in lib.trl_last_error()
  rc = lib.trl_ppo_loss_static(*p, (1 << 30) + 1, 1, 0.2, 0.5, 0.0, out.data_ptr(), None, None, None, None)
  assert rc == -1 and b'beyond the limit' in lib.trl_last_error()
  rc = lib.trl_ppo_loss_static(*p, 0, 1, 0.2, 0.5, 0.0, out.data_ptr(), None, None, None, None)
  assert rc == -1
