"""Device-resident on-policy rollout staging ([T, N, .] tensors).

The reference keeps one TransitionTupleDataset per env, appends one transition per env per step
(`torch.cat` of the whole column each time), and at train time walks the envs in Python:
`buffer[:]`, `compute_return`, `torch.cat` over envs (examples/ppo.py:28-35, 43-68; a2c.py the
same).  Here one env step over all N envs is ONE row of preallocated `[T, N, .]` device
tensors (one scatter launch reading the packed transitions from pinned memory), the return/advantage pass is one batched call,
and minibatches come out in the reference's order: env-major, sequential, last partial batch
kept (`DataLoader(ds, batch_size)` at examples/ppo.py:68).

`layout='blocked'` keeps the same rollout as `[N / E, T, E, .]` (E = 8 or 16 consecutive envs per group,
a group's T steps contiguous): the value pass then writes its values in that order and the return /
advantage scan streams one flat run per env group (csrc/trl_gae.hip, trl_gae_reverse_scan_blocked)
instead of a 4 * E-byte segment of every `[T, N]` row.  Minibatches of a blocked rollout are
consecutive slices of that order (a group's steps, E envs interleaved) -- the same samples per
epoch as the reference's env-major order, cut into minibatches differently; use the default layout
with `env_major()` where the reference's exact minibatch composition matters.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


class RolloutBuffer:
  def __init__(self, n_envs, device=None, initial_steps=32, layout='tn', block_envs=16):
    if layout not in ('tn', 'blocked'):
      raise ValueError("layout must be 'tn' or 'blocked'")
    if layout == 'blocked' and (block_envs not in (8, 16) or n_envs % block_envs):
      raise ValueError('the blocked layout needs block_envs 8 or 16 dividing n_envs=%d' % n_envs)
    self.layout, self.block_envs = layout, block_envs
    self.n_envs = n_envs
    self._device = torch.device(device) if device is not None else None
    self._cap = initial_steps
    self._t = 0
    self._cols = None          # obs, action, reward, done  [cap, N, .]
    self._last_next_obs = None
    self._pin = None

  def __len__(self):
    return self._t

  def _bind(self):
    if self._device is None:
      if not torch.cuda.is_available():
        raise _lib.TrlError('RolloutBuffer needs a HIP device (no CPU fallback)')
      self._device = torch.device('cuda', torch.cuda.current_device())
    if self._device.type != 'cuda':
      raise _lib.TrlError('RolloutBuffer lives on a HIP device, not %s' % self._device)

  def _alloc(self, widths, cap):
    if self.layout == 'blocked':
      E = self.block_envs
      return [torch.empty((self.n_envs // E, cap, E, w), dtype=torch.float32, device=self._device) for w in widths]
    return [torch.empty((cap, self.n_envs, w), dtype=torch.float32, device=self._device) for w in widths]

  def extend(self, transition_list):
    """One Transition per env (env order), i.e. one time step; fields become float32 like
    TransitionTupleDataset.extend (storage.py:65-67)."""
    if len(transition_list) != self.n_envs:
      raise ValueError('expected %d transitions (one per env), got %d' % (self.n_envs, len(transition_list)))
    self._bind()
    fields = [np.asarray(f, dtype=np.float32).reshape(self.n_envs, -1) for f in zip(*transition_list)]
    obs, action, reward, next_obs, done = fields
    row = np.concatenate([obs, action, reward, done, next_obs], axis=1)
    if self._cols is None:
      self._widths = [obs.shape[1], action.shape[1], 1, 1]
      self._cols = self._alloc(self._widths, self._cap)
      self._last_next_obs = torch.empty((self.n_envs, obs.shape[1]), dtype=torch.float32, device=self._device)
      self._pin = [torch.empty(row.shape, dtype=torch.float32).pin_memory() for _ in range(8)]
      self._pin_ev = [None] * 8
      cols = self._push_cols = _lib.RolloutCols()
      cols.ncols = 5
      for i, w in enumerate(self._widths + [obs.shape[1]]):
        cols.width[i] = w
      cols.dst[4] = self._last_next_obs.data_ptr()           # next_obs of the step: dense [N, O], overwritten every step
      cols.block_envs[4], cols.group_stride[4] = 1, obs.shape[1]
    if self._t == self._cap:
      new = self._alloc(self._widths, 2 * self._cap)
      for n, o in zip(new, self._cols):
        (n[:, :self._cap] if self.layout == 'blocked' else n[:self._cap]).copy_(o)
      self._cols, self._cap = new, 2 * self._cap
    slot = self._t % len(self._pin)
    if self._pin_ev[slot] is not None:
      self._pin_ev[slot].synchronize()
    self._pin[slot].copy_(torch.from_numpy(row))
    # ONE launch: the push kernel reads the packed rows straight out of pinned memory and scatters every
    # column to its place at step t (csrc/trl_act.hip; no DMA-engine copy, no per-column slice copies)
    cols = self._push_cols
    E = self.block_envs
    for i, (col, w) in enumerate(zip(self._cols, self._widths)):
      if self.layout == 'blocked':
        cols.dst[i] = col.data_ptr() + 4 * self._t * E * w
        cols.block_envs[i], cols.group_stride[i] = E, self._cap * E * w
      else:
        cols.dst[i] = col.data_ptr() + 4 * self._t * self.n_envs * w
        cols.block_envs[i], cols.group_stride[i] = 1, w
    with torch.cuda.device(self._device):
      _lib.check(_lib.lib().trl_rollout_push(self._pin[slot].data_ptr(), self.n_envs, C.byref(cols),
                                             _lib.stream_ptr(self._device)))
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(self._device))
    self._pin_ev[slot] = ev
    self._t += 1

  def truncate(self):
    self._t = 0

  def tensors(self):
    """(obs [T,N,O], action [T,N,A], reward [T,N], done [T,N], last_next_obs [N,O]); blocked layout:
    (obs [N/E,T,E,O], action [N/E,T,E,A], reward [N/E,T,E], done [N/E,T,E], last_next_obs [N,O])."""
    T = self._t
    if self.layout == 'blocked':
      # (width-1 columns: dropping the unit axis leaves strides (capacity * E, E, 1), what the scan takes)
      obs, action, reward, done = (c[:, :T] for c in self._cols)
      return obs, action, reward[..., 0], done[..., 0], self._last_next_obs
    obs, action, reward, done = (c[:T] for c in self._cols)
    return obs, action, reward[..., 0], done[..., 0], self._last_next_obs


def env_major(x):
  """[T, N, ...] -> [N*T, ...] in the reference's concatenation order (env 0's T rows first)."""
  T, N = x.shape[:2]
  return x.transpose(0, 1).reshape((N * T,) + tuple(x.shape[2:]))


def iter_minibatches(columns, batch_size):
  """Sequential minibatches over equally long columns, last partial batch kept."""
  n = columns[0].shape[0]
  for s in range(0, n, batch_size):
    yield [c[s:s + batch_size] for c in columns]
