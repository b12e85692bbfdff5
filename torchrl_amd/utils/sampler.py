"""Index samplers for the replay ring.

`randperm_prefix(n, B)` is the drop-in for `torch.randperm(n)[:B]` (examples/dqn.py:40,
examples/ddpg.py:40): same indices, and torch's default CPU generator is left in the state
the reference's call would leave it in, so everything downstream of it (weight init, noise)
stays in lock-step.  The work is done by `trl_randperm_prefix` in libtrl_hip.so (host C++,
O(n) draws, O(B) memory); this module only moves the MT19937 state in and out of torch's
serialized generator state.

Layout of torch.get_rng_state() for the CPU generator (5056 bytes, little endian),
at::CPUGeneratorImplState: u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] |
f64 normal_x, normal_y, normal_rho | i32 normal_is_valid (+pad) | f32 next_float_normal |
bool valid (+pad).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib

_STATE_BYTES = 5056
_OFF_LEFT, _OFF_NEXT, _OFF_MT, _N = 8, 16, 24, 624


class Mt19937Sampler:
  """A private MT19937 stream (torch's engine and seeding), e.g. one per replay shard."""

  def __init__(self, seed=0):
    self._mt = np.zeros(_N, dtype=np.uint32)
    self._left = C.c_int32(1)
    self._next = C.c_int32(0)
    self.seed(seed)

  def seed(self, seed):
    _lib.check(_lib.lib().trl_mt19937_seed(self._mt.ctypes.data, C.byref(self._left),
                                           C.byref(self._next), C.c_uint64(int(seed) & (2 ** 64 - 1))))

  def randperm_prefix(self, n, batch_size, jump=True):
    """jump=False: the block-by-block skip only (trl_randperm_prefix_direct); same indices and state."""
    out = np.empty(int(batch_size), dtype=np.int64)
    fn = _lib.lib().trl_randperm_prefix if jump else _lib.lib().trl_randperm_prefix_direct
    _lib.check(fn(self._mt.ctypes.data, C.byref(self._left), C.byref(self._next), int(n), int(batch_size),
                  out.ctypes.data))
    return torch.from_numpy(out)


last_hit = False     # whether the latest randperm_prefix(..., next_n=...) used the result computed ahead of time


def randperm_prefix(n, batch_size, generator=None, next_n=None):
  """`torch.randperm(n, generator=generator)[:batch_size]`, bit for bit, in O(batch) memory.

  next_n: the `n` the caller expects to ask for next (a learner sampling a full ring every step: the same n).
  The library then computes that draw on a helper thread from the state this call leaves, and the next call only
  copies it if the generator is still in that state (trl_randperm_prefix_ahead); same indices, same state."""
  gen = torch.default_generator if generator is None else generator
  state = gen.get_state().numpy().copy()
  if state.size != _STATE_BYTES:
    raise _lib.TrlError('unexpected CPU generator state of %d bytes (torch %s)' % (state.size, torch.__version__))
  mt64 = state[_OFF_MT:_OFF_MT + 8 * _N].view(np.uint64)
  mt = mt64.astype(np.uint32)
  left = C.c_int32(int(state[_OFF_LEFT:_OFF_LEFT + 4].view(np.int32)[0]))
  nxt = C.c_int32(int(state[_OFF_NEXT:_OFF_NEXT + 8].view(np.uint64)[0]))
  out = np.empty(int(batch_size), dtype=np.int64)
  if next_n is None:
    _lib.check(_lib.lib().trl_randperm_prefix(mt.ctypes.data, C.byref(left), C.byref(nxt), int(n),
                                              int(batch_size), out.ctypes.data))
  else:
    global last_hit
    hit = C.c_int32(0)
    _lib.check(_lib.lib().trl_randperm_prefix_ahead(mt.ctypes.data, C.byref(left), C.byref(nxt), int(n),
                                                    int(batch_size), out.ctypes.data, int(next_n), C.byref(hit)))
    last_hit = bool(hit.value)
  mt64[:] = mt
  state[_OFF_LEFT:_OFF_LEFT + 4].view(np.int32)[0] = left.value
  state[_OFF_NEXT:_OFF_NEXT + 8].view(np.uint64)[0] = nxt.value
  gen.set_state(torch.from_numpy(state))
  return torch.from_numpy(out)
