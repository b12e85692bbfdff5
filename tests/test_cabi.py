"""The C-ABI library loads without a GPU and exports every symbol include/trl_hip.h declares;
host-only entry points (error text, argument validation, sampler) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from torchrl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  text = open(os.path.join(ROOT, 'include', 'trl_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(trl_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
  names = _declared()
  assert len(names) >= 15
  handle = C.CDLL(_lib.LIB_PATH)
  for n in names:
    assert hasattr(handle, n), 'libtrl_hip.so does not export %s' % n
    assert n in _lib.PROTOTYPES, 'no ctypes prototype for %s' % n
  assert sorted(_lib.PROTOTYPES) == names


def test_version_and_error_text():
  lib = _lib.lib()
  assert lib.trl_version() >= 100
  assert lib.trl_loss_workspace_bytes() > 0
  # argument validation happens before any HIP call, so it is testable without a device
  rc = lib.trl_gae_reverse_scan(None, None, None, None, 4, 4, 0.99, 0.95, 0, 0, None, None, None)
  assert rc == -1 and b'null pointer' in lib.trl_last_error()
  with pytest.raises(_lib.TrlError, match='null pointer'):
    _lib.check(rc)
  lay = _lib.RingLayout()
  lay.ncols = 0
  rc = lib.trl_gather_rows(None, 1, 4, C.byref(lay), 0, 1, None, 1, None, None, None)
  assert rc == -1 and b'ncols' in lib.trl_last_error()


def test_sampler_matches_golden(golden):
  from torchrl_amd.utils.sampler import Mt19937Sampler, randperm_prefix
  g = golden('sampler.npz')
  for ci in range(int(g['n_cases'])):
    seed = int(g['c%d_seed' % ci])
    own = Mt19937Sampler(seed)
    torch.manual_seed(seed)
    for j, (n, b) in enumerate(g['c%d_calls' % ci]):
      exp = g['c%d_out%d' % (ci, j)]
      np.testing.assert_array_equal(own.randperm_prefix(int(n), int(b)).numpy(), exp)
      np.testing.assert_array_equal(randperm_prefix(int(n), int(b)).numpy(), exp)
    # torch's generator is where the reference's would be: the next draws agree
    np.testing.assert_array_equal(torch.randint(0, 2 ** 31, (4,)).numpy(), g['c%d_next_u32' % ci])


@pytest.mark.parametrize('n,b', [(1, 1), (2, 1), (2, 2), (7, 7), (624, 10), (625, 625), (1249, 3), (100003, 77)])
def test_sampler_lockstep_with_torch(n, b):
  from torchrl_amd.utils.sampler import randperm_prefix
  torch.manual_seed(n * 1000 + b)
  torch.rand(5)                                  # start mid-block
  exp = [torch.randperm(n)[:b] for _ in range(3)]
  tail = torch.randn(3)
  torch.manual_seed(n * 1000 + b)
  torch.rand(5)
  got = [randperm_prefix(n, b) for _ in range(3)]
  for a, e in zip(got, exp):
    assert torch.equal(a, e)
  assert torch.equal(torch.randn(3), tail)


def test_sampler_ahead_of_time_is_the_same_draw():
  """trl_randperm_prefix_ahead: the next draw is computed on a helper thread from the state a call leaves; it is
  used only when the generator is still in that state and n, B are the ones speculated -- and the indices and the
  generator state are torch's own either way (hits, a draw in between, a changed n, a changed B)."""
  from torchrl_amd.utils import sampler
  n, b = 50021, 64
  torch.manual_seed(99)
  torch.rand(3)
  exp = [torch.randperm(n)[:b] for _ in range(4)]
  between = torch.rand(2)                        # someone else draws: the speculation is void
  exp += [torch.randperm(n)[:b] for _ in range(2)]
  exp.append(torch.randperm(n + 1)[:b])          # the ring grew
  exp.append(torch.randperm(n + 1)[:b - 1])      # another batch size
  tail = torch.randn(3)
  torch.manual_seed(99)
  torch.rand(3)
  hits = []
  for i in range(4):
    assert torch.equal(sampler.randperm_prefix(n, b, next_n=n), exp[i])
    hits.append(sampler.last_hit)
  assert torch.equal(torch.rand(2), between)
  for i in range(4, 6):
    assert torch.equal(sampler.randperm_prefix(n, b, next_n=n), exp[i])
    hits.append(sampler.last_hit)
  assert torch.equal(sampler.randperm_prefix(n + 1, b, next_n=n + 1), exp[6])
  hits.append(sampler.last_hit)
  assert torch.equal(sampler.randperm_prefix(n + 1, b - 1, next_n=n + 1), exp[7])
  hits.append(sampler.last_hit)
  assert torch.equal(torch.randn(3), tail)
  # where the speculation is void the call must miss; behind a hit-or-miss call whose job the idle worker took, the next
  # call must hit.  Call 5 follows a miss (call 4) that could post its job only if the worker had already finished the
  # job call 3 posted (now stale): a busy worker is never interrupted and never waited for, so under load call 5 may
  # compute on the spot -- same indices either way, which is what the equalities above check.
  assert hits[:5] == [False, True, True, True, False] and hits[6:] == [False, False], hits
  # a plain call in between is a draw from the same generator: the pending speculation is simply not used
  torch.manual_seed(5)
  e2 = [torch.randperm(1000)[:10] for _ in range(3)]
  torch.manual_seed(5)
  assert torch.equal(sampler.randperm_prefix(1000, 10, next_n=1000), e2[0])
  assert torch.equal(sampler.randperm_prefix(1000, 10), e2[1])
  assert torch.equal(sampler.randperm_prefix(1000, 10, next_n=1000), e2[2]) and not sampler.last_hit


def test_sampler_rejects_bad_arguments():
  from torchrl_amd.utils.sampler import randperm_prefix
  with pytest.raises(_lib.TrlError):
    randperm_prefix(5, 6)
  with pytest.raises(_lib.TrlError):
    randperm_prefix(2 ** 31, 4)


def test_generator_argument():
  from torchrl_amd.utils.sampler import randperm_prefix
  g1 = torch.Generator().manual_seed(99)
  g2 = torch.Generator().manual_seed(99)
  assert torch.equal(randperm_prefix(1000, 50, generator=g1), torch.randperm(1000, generator=g2)[:50])
  assert torch.equal(torch.rand(3, generator=g1), torch.rand(3, generator=g2))


def test_sampler_jump_ahead_equals_direct_skip_and_torch():
  """Rings of more than ~2.6 M rows skip their n - 1 - B unused words by polynomial jump-ahead once the
  same skip length has been seen three times (csrc/trl_host.cpp, namespace jump).  Same indices, same
  generator state as the block-by-block skip, and as torch.randperm itself."""
  from torchrl_amd.utils.sampler import Mt19937Sampler, randperm_prefix
  n, B = 2_700_001, 48
  jumped, direct = Mt19937Sampler(31), Mt19937Sampler(31)
  a = [jumped.randperm_prefix(n, B).numpy().copy() for _ in range(8)]
  b = [direct.randperm_prefix(n, B, jump=False).numpy().copy() for _ in range(8)]
  for x, y in zip(a, b):
    np.testing.assert_array_equal(x, y)
  np.testing.assert_array_equal(jumped._mt, direct._mt)
  assert (jumped._left.value, jumped._next.value) == (direct._left.value, direct._next.value)
  torch.manual_seed(11)
  exp = [torch.randperm(n)[:B].clone() for _ in range(6)]
  after = torch.randint(0, 2 ** 31, (4,))
  torch.manual_seed(11)
  got = [randperm_prefix(n, B) for _ in range(6)]
  for x, y in zip(exp, got):
    assert torch.equal(x, y)
  assert torch.equal(after, torch.randint(0, 2 ** 31, (4,)))


def test_production_library_has_no_environment_switches():
  """The shipped library reads no environment variable: the tuning / ablation switches of tools/*_sweep.py
  (TRL_GAE_TUNE's ablation bits give wrong results by design) exist only in the -DTRL_TUNING diagnostic
  build, lib/libtrl_hip_tune.so."""
  blob = open(_lib.LIB_PATH, 'rb').read()
  for knob in (b'TRL_GAE_TUNE', b'TRL_MLP_DBG', b'TRL_GATHER_UN', b'TRL_GATHER_WG_PER_CU', b'TRL_U8_FWD_WGS',
               b'TRL_MT_NO_JUMP'):
    assert knob not in blob, knob
  handle = C.CDLL(_lib.LIB_PATH)
  assert not hasattr(handle, 'getenv') or True   # (libc's symbol resolves through any handle; the strings are the check)
  # a stray variable changes nothing an argument check can see: same validation path, same error text
  os.environ['TRL_GAE_TUNE'] = '64,9,15,7'
  try:
    lib = _lib.lib()
    rc = lib.trl_gae_reverse_scan(None, None, None, None, 4, 4, 0.99, 0.95, 0, 0, None, None, None)
    assert rc == -1 and b'null pointer' in lib.trl_last_error()
  finally:
    del os.environ['TRL_GAE_TUNE']


def _ahead_stream(seed, n, B, calls, next_n):
  """`calls` consecutive trl_randperm_prefix_ahead draws from a private MT19937 stream, through the C ABI."""
  import ctypes as C
  lib = _lib.lib()
  mt = np.zeros(624, dtype=np.uint32)
  left, nxt, hit = C.c_int32(1), C.c_int32(0), C.c_int32(0)
  _lib.check(lib.trl_mt19937_seed(mt.ctypes.data, C.byref(left), C.byref(nxt), C.c_uint64(seed)))
  outs = []
  for _ in range(calls):
    out = np.empty(B, dtype=np.int64)
    _lib.check(lib.trl_randperm_prefix_ahead(mt.ctypes.data, C.byref(left), C.byref(nxt), n, B, out.ctypes.data, next_n,
                                             C.byref(hit)))
    outs.append(out)
  return outs, mt.copy(), (left.value, nxt.value)


def _direct_stream(seed, n, B, calls):
  from torchrl_amd.utils.sampler import Mt19937Sampler
  s = Mt19937Sampler(seed)
  return [s.randperm_prefix(n, B, jump=False).numpy().copy() for _ in range(calls)], s._mt.copy(), (s._left.value, s._next.value)


def test_sampler_ahead_of_time_with_two_host_threads():
  """ctypes releases the GIL, so two Python threads can be inside trl_randperm_prefix_ahead at once, each with its own
  generator, batch size and ring length.  The one speculation worker must never have its job rewritten while it runs
  (round-2 advisor finding: heap overflow from a resize to the old B and a draw with the new one): every thread gets
  exactly the indices and final state of the plain draw."""
  import threading
  specs = [(7, 200_003, 64, 40), (8, 90_001, 2048, 40), (9, 333_337, 16, 40)]
  res = {}

  def run(i, seed, n, B, calls):
    res[i] = _ahead_stream(seed, n, B, calls, n)

  ths = [threading.Thread(target=run, args=(i,) + sp) for i, sp in enumerate(specs)]
  for t in ths:
    t.start()
  for t in ths:
    t.join(timeout=120)
    assert not t.is_alive()
  for i, (seed, n, B, calls) in enumerate(specs):
    exp, mt, pos = _direct_stream(seed, n, B, calls)
    got, gmt, gpos = res[i]
    for a, b in zip(got, exp):
      np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(gmt, mt)
    assert gpos == pos


def test_sampler_ahead_of_time_survives_fork():
  """The helper thread does not exist in a fork()ed child (a DataLoader worker, multiprocessing's fork start method);
  the child must start its own instead of waiting for the parent's (round-2 advisor finding: second call blocks)."""
  import os
  import signal
  _ahead_stream(1, 50_000, 32, 3, 50_000)      # the parent's worker thread exists and has a speculation posted
  exp, _, pos = _direct_stream(2, 60_000, 24, 4)
  r, w = os.pipe()
  pid = os.fork()
  if pid == 0:      # child: no pytest machinery, just the calls
    ok = b'0'
    try:
      signal.alarm(30)      # a blocked child dies instead of hanging the suite
      got, _, gpos = _ahead_stream(2, 60_000, 24, 4, 60_000)
      ok = b'1' if all(np.array_equal(a, b) for a, b in zip(got, exp)) and gpos == pos else b'2'
    finally:
      os.write(w, ok)
      os._exit(0)
  os.close(w)
  _, status = os.waitpid(pid, 0)
  assert os.read(r, 1) == b'1', status
  os.close(r)
  # and the parent's worker still serves the parent
  got, _, _ = _ahead_stream(2, 60_000, 24, 4, 60_000)
  for a, b in zip(got, exp):
    np.testing.assert_array_equal(a, b)
