"""Device-resident replay storage with the reference's surface.

Drop-in for torchrl/utils/storage.py: `TensorTupleDataset(size, x).extend / len / [index] /
truncate`, `Transition`, `TransitionTupleDataset.extend(list[Transition])`.  Behind it the
reference's "cat the whole column and slice" FIFO (storage.py:7-16, O(capacity) per push)
is a preallocated HBM ring of fixed-stride records (layout: csrc/trl_replay.hip), and
`buf[LongTensor]` / `buf[:]` are HIP gather kernels that return dense per-column batches on
the device.  Logical index 0 is always the oldest surviving row, as in the reference.

Differences a caller can observe (all additive):
  * rows live on a HIP device; results come back on that device (`.to(device)` in
    examples/dqn.py:41 is then a no-op).  There is no CPU storage path.
  * columns are stored as float32 (what `torch.Tensor(b)` yields at storage.py:65-67), or,
    opt-in, as uint8 with an exact cast back to float32 at gather time.  A column first pushed
    as float64 is kept float64, as `torch.cat` promotion does in the reference (storage.py:35 +
    :10); the one promotion not reproduced is a float64 push into a column that already holds
    float32 rows (the reference re-types the whole column then; here that push raises).
  * `sample(batch_size)` = `self[torch.randperm(len(self))[:batch_size]]` with the same
    indices and the same effect on torch's CPU generator, without the O(n) permutation.
"""
import ctypes as C
import os
from collections import namedtuple
from typing import List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import _lib
from .. import config
from .sampler import randperm_prefix

Transition = namedtuple('Transition', ['obs', 'action', 'reward', 'next_obs', 'done'])

_WIDE_BYTES = 1024        # must match WIDE_MIN_BYTES in csrc/trl_replay.hip
_MIN_UNBOUNDED_ROWS = 1024
_ITEMSIZE = {torch.uint8: 1, torch.float32: 4, torch.float64: 8}
_F32_PROMOTABLE = (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.uint8, torch.int8,
                   torch.int16, torch.int32, torch.int64, torch.bool)


def _default_device():
  if not torch.cuda.is_available():
    raise _lib.TrlError('torchrl_amd replay storage needs a HIP device; none is visible '
                        '(there is no CPU fallback)')
  return torch.device('cuda', torch.cuda.current_device())


class _Layout:
  """Record layout of one ring: offsets/widths per column + the ctypes mirror."""

  def __init__(self, feat_shapes, storage_dtypes):
    if len(feat_shapes) > _lib.TRL_MAX_COLS:
      raise ValueError('at most %d columns per dataset, got %d' % (_lib.TRL_MAX_COLS, len(feat_shapes)))
    self.feat_shapes = [tuple(s) for s in feat_shapes]
    self.storage_dtypes = list(storage_dtypes)
    self.widths = [int(np.prod(s, dtype=np.int64)) if len(s) else 1 for s in self.feat_shapes]
    self.col_bytes = [w * _ITEMSIZE[d] for w, d in zip(self.widths, self.storage_dtypes)]
    off, offs, wide = 0, [], False
    for nbytes in self.col_bytes:
      is_wide = nbytes >= _WIDE_BYTES and nbytes % 16 == 0
      wide = wide or is_wide
      align = 16 if is_wide else 4
      off = (off + align - 1) // align * align
      offs.append(off)
      off += (nbytes + 3) // 4 * 4
    align = 16 if wide else 4
    self.stride = (off + align - 1) // align * align
    # A narrow record (CartPole 44 B, Pendulum 36 B) is padded to the next power of two: the ring's base is 256 B
    # aligned, so a record then never straddles a 64 B memory request -- a sampled row is ONE request instead of one or
    # two (measured, rocprofv3 FETCH_SIZE over 65,536 random rows of a 1 M-row ring: 5.47 -> 4.48 MB fetched, 64 B per
    # row + the indices; tools/narrow_pad_experiment.sh).  Costs ring capacity (36 -> 64 MB per million rows), nothing else.
    self.payload = self.stride
    if not wide and 16 < self.stride <= 128:
      p2 = 32
      while p2 < self.stride:
        p2 *= 2
      self.stride = p2
    self.offsets = offs
    self.has_wide = wide
    self.c = _lib.RingLayout()
    self.c.ncols = len(offs)
    for i, (o, b, d) in enumerate(zip(offs, self.col_bytes, self.storage_dtypes)):
      self.c.col_off[i] = o
      self.c.col_bytes[i] = b
      # (a float64 column is 2 raw dwords per element to the byte-moving kernels: TRL_F32 = "no cast")
      self.c.col_dtype[i] = _lib.TRL_U8 if d == torch.uint8 else _lib.TRL_F32

  def same_as(self, other):
    return (other is not None and self.feat_shapes == other.feat_shapes and
            self.storage_dtypes == other.storage_dtypes)

  @property
  def ncols(self):
    return len(self.offsets)


class TensorTupleDataset(Dataset):
  """Vectorized tuples of tensors in an HBM ring (reference: storage.py:19-48).

  size:            maximum number of rows kept (FIFO); -1 = unbounded (the ring doubles).
  x:               optional list of [n, ...] tensors to start from (examples/ppo.py:66).
  device:          HIP device; default = device of the first tensors pushed if they are on
                   one, else the current HIP device.  Bound lazily because the examples
                   build the buffer before they know the device (examples/dqn.py:22-25).
  storage_dtypes:  per-column torch.float32 (default) or torch.uint8.
  gather_dtypes:   per-column dtype of gathered batches; default float32 for every column
                   (the reference hands out float32), torch.uint8 allowed for uint8 columns.
  """

  def __init__(self, size: int = -1, x: Optional[List] = None, device=None,
               storage_dtypes=None, gather_dtypes=None):
    self.size = size
    self._device = torch.device(device) if device is not None else None
    self._storage_dtypes = storage_dtypes
    self._gather_dtypes = gather_dtypes
    self._layout = None
    self._ring = None
    self._capacity = 0
    self._head = 0
    self._count = 0
    self._pin = None
    self._packed_layout = None
    self._stage = None
    if x is not None:
      self.extend(*x)

  # ------------------------------------------------------------------ introspection
  @property
  def device(self):
    return self._device

  @property
  def capacity(self):
    return self._capacity

  @property
  def row_bytes(self):
    return 0 if self._layout is None else self._layout.stride

  def __len__(self) -> int:
    return self._count

  # ------------------------------------------------------------------ push
  def _bind(self, tensors):
    if self._device is None:
      on_dev = [t.device for t in tensors if t.is_cuda]
      self._device = on_dev[0] if on_dev else _default_device()
    if self._device.type != 'cuda':
      raise _lib.TrlError('torchrl_amd replay storage lives on a HIP device, not %s' % self._device)

  def _make_layout(self, tensors):
    if self._storage_dtypes is not None:
      sd = self._storage_dtypes
    elif self._layout is not None and self._count > 0:
      sd = self._layout.storage_dtypes        # a column keeps the dtype it was created with
    else:
      # zeros(0, ...) float32 cat the pushed tensor (storage.py:35, :10): float64 stays float64,
      # everything else torch.cat promotes to float32
      sd = [torch.float64 if t.dtype == torch.float64 else torch.float32 for t in tensors]
    if len(sd) != len(tensors):
      raise ValueError('storage_dtypes has %d entries for %d columns' % (len(sd), len(tensors)))
    for d, t in zip(sd, tensors):
      if d not in (torch.float32, torch.uint8, torch.float64):
        raise TypeError('columns are stored as float32, float64 or uint8, not %s' % d)
      if t.dtype == torch.float64 and d != torch.float64:
        raise TypeError('a float64 push into a %s column: the reference would re-type the whole column to '
                        'float64 here (torch.cat promotion); torchrl_amd keeps a column\'s dtype -- push '
                        'float64 from the start or cast the rows' % d)
    return _Layout([t.shape[1:] for t in tensors], sd)

  def _allocate(self, rows):
    nbytes = rows * self._layout.stride
    return torch.empty(nbytes, dtype=torch.uint8, device=self._device)

  def _grow(self, need):
    """Unbounded mode: double the ring, oldest row moves to record 0."""
    new_cap = max(_MIN_UNBOUNDED_ROWS, self._capacity)
    while new_cap < need:
      new_cap *= 2
    new_ring = self._allocate(new_cap)
    if self._count:
      st = self._layout.stride
      old = self._ring.view(self._capacity, st)
      new = new_ring.view(new_cap, st)
      first = min(self._count, self._capacity - self._head)
      new[:first].copy_(old[self._head:self._head + first])
      if first < self._count:
        new[first:self._count].copy_(old[:self._count - first])
    self._ring, self._capacity, self._head = new_ring, new_cap, 0

  def extend(self, *tensor_list: List[torch.Tensor]):
    """FIFO-append k rows per column; keeps the last `size` rows (storage.py:27-37)."""
    tensors = [torch.as_tensor(t) for t in tensor_list]
    if self._count > 0 and len(tensors) != self._layout.ncols:   # storage.py:31
      raise AssertionError('expected %d columns, got %d' % (self._layout.ncols, len(tensors)))
    if not tensors:
      return
    k = tensors[0].shape[0]
    for t in tensors:
      if t.dim() < 1 or t.shape[0] != k:
        raise ValueError('every column must be [k, ...] with the same k')
      if t.dtype not in _F32_PROMOTABLE:
        raise TypeError('column dtype %s is not one torchrl_amd stores (float32 / float64 / uint8 columns)' % t.dtype)
    self._bind(tensors)
    layout = self._make_layout(tensors)
    if self._layout is None or self._count == 0:
      if not layout.same_as(self._layout):     # first push, or new shapes after truncate()
        if self._layout is None or layout.stride != self._layout.stride:
          self._ring, self._capacity = None, 0
        self._layout = layout
      self._head = 0
    elif not layout.same_as(self._layout):
      raise ValueError('column shapes %s do not match the stored %s' %
                       (layout.feat_shapes, self._layout.feat_shapes))
    if k == 0:
      return
    if self.size > 0 and k > self.size:      # only the last `size` rows survive (storage.py:12-14)
      tensors = [t[k - self.size:] for t in tensors]
      k = self.size
    if self._ring is None:
      self._capacity = self.size if self.size > 0 else max(_MIN_UNBOUNDED_ROWS, k)
      self._ring = self._allocate(self._capacity)
    if self.size <= 0 and self._count + k > self._capacity:
      self._grow(self._count + k)

    tail = (self._head + self._count) % self._capacity
    if all(not t.is_cuda for t in tensors):
      self._push_host(tensors, k, tail)
    else:
      self._push_device(tensors, k, tail)
    over = self._count + k - self._capacity
    if over > 0:
      self._head = (self._head + over) % self._capacity
      self._count = self._capacity
    else:
      self._count += k

  def _push_device(self, tensors, k, tail):
    """Rows already on a device (or mixed): scatter kernel from dense per-column sources."""
    src = []
    for t, sdt in zip(tensors, self._layout.storage_dtypes):
      t = t.reshape(k, -1)
      if t.dtype != sdt:
        t = t.to(sdt)
      t = t.to(self._device, non_blocking=True).contiguous()
      if t.data_ptr() % 16:     # a slice of the caller's tensor: the kernel wants aligned rows
        t = t.clone()
      src.append(t)
    srcp = (C.c_void_p * len(src))(*[s.data_ptr() for s in src])
    with torch.cuda.device(self._device):
      _lib.check(_lib.lib().trl_ring_push(self._ring.data_ptr(), self._capacity, self._layout.stride,
                                          C.byref(self._layout.c), tail, srcp, k,
                                          _lib.stream_ptr(self._device)))
    for s_ in src:   # keep staging alive until the stream has consumed it
      s_.record_stream(torch.cuda.current_stream(self._device))

  _STAGE_BYTES = 32 << 20
  _KERNEL_PUSH_BYTES = 4 << 20

  def _push_host(self, tensors, k, tail):
    """Rows that are host data (what store() hands over every env step, examples/dqn.py:27-28):
    packed on the host straight into the ring's record layout in pinned memory, then ONE launch of
    the push kernel, which reads the pinned records itself (no per-column copies, no DMA-engine copy
    in the learner's stream).  Bulk loads (more than _KERNEL_PUSH_BYTES per slice) go by async H2D
    copies straight into the ring, in slices of the pinned staging size."""
    st = self._layout.stride
    rows_per = max(1, self._STAGE_BYTES // st)
    cols = []
    for t, sdt in zip(tensors, self._layout.storage_dtypes):
      t = t.reshape(k, -1)
      cols.append((t if t.dtype == sdt else t.to(sdt)).contiguous().numpy())
    ring2d = self._ring.view(self._capacity, st)
    if self._stage is None:
      nbytes = min(self._STAGE_BYTES, max(1 << 16, 1 << (k * st - 1).bit_length()))
      self._stage = [torch.empty(max(nbytes, st), dtype=torch.uint8).pin_memory() for _ in range(4)]
      self._stage_ev = [None] * 4
      self._stage_next = 0
    for lo in range(0, k, rows_per):
      n = min(rows_per, k - lo)
      if self._stage[0].numel() < n * st:        # grow (rare): the pushes got bigger
        for ev in self._stage_ev:
          if ev is not None:
            ev.synchronize()
        nbytes = min(max(self._STAGE_BYTES, st), 1 << (n * st - 1).bit_length())
        self._stage = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(4)]
        self._stage_ev = [None] * 4
      slot = self._stage_next
      self._stage_next = (slot + 1) % 4
      if self._stage_ev[slot] is not None:
        self._stage_ev[slot].synchronize()       # the copy issued four pushes ago
      rec = self._stage[slot][:n * st].view(n, st)
      recn = rec.numpy()
      for a, off, nb in zip(cols, self._layout.offsets, self._layout.col_bytes):
        recn[:, off:off + nb] = a[lo:lo + n].view(np.uint8).reshape(n, nb)
      t0 = (tail + lo) % self._capacity
      if n * st <= self._KERNEL_PUSH_BYTES and config.kernel_push:
        # the per-env-step case: the push kernel reads the packed records straight out of the pinned
        # buffer (one launch, wrap included).  An async H2D copy of a few hundred bytes is a DMA-engine
        # job with a cross-queue dependency on either side: ~175 us of stall in the learner's stream.
        if self._packed_layout is None:
          lay = _lib.RingLayout()
          lay.ncols, lay.col_off[0], lay.col_bytes[0], lay.col_dtype[0] = 1, 0, st, _lib.TRL_U8
          self._packed_layout = lay
        srcp = (C.c_void_p * 1)(rec.data_ptr())
        with torch.cuda.device(self._device):
          _lib.check(_lib.lib().trl_ring_push(self._ring.data_ptr(), self._capacity, st,
                                              C.byref(self._packed_layout), t0, srcp, n,
                                              _lib.stream_ptr(self._device)))
      else:
        first = min(n, self._capacity - t0)
        ring2d[t0:t0 + first].copy_(rec[:first], non_blocking=True)
        if first < n:
          ring2d[:n - first].copy_(rec[first:], non_blocking=True)
      ev = torch.cuda.Event()
      ev.record(torch.cuda.current_stream(self._device))
      self._stage_ev[slot] = ev

  # ------------------------------------------------------------------ gather
  def _out_dtypes(self):
    sd = self._layout.storage_dtypes
    gd = self._gather_dtypes or [torch.float64 if s == torch.float64 else torch.float32 for s in sd]
    for g, s in zip(gd, sd):
      if g != s and not (g == torch.float32 and s == torch.uint8):
        raise TypeError('cannot gather a %s column as %s' % (s, g))
    return gd

  def _alloc_out(self, B):
    gd = self._out_dtypes()
    outs = [torch.empty((B,) + shp, dtype=g, device=self._device)
            for shp, g in zip(self._layout.feat_shapes, gd)]
    dstp = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
    odt = (C.c_int32 * len(outs))(*[_lib.TRL_U8 if g == torch.uint8 else _lib.TRL_F32 for g in gd])
    return outs, dstp, odt

  def _gather_range(self, start, B):
    outs, dstp, odt = self._alloc_out(B)
    if B:
      with torch.cuda.device(self._device):
        _lib.check(_lib.lib().trl_gather_range(self._ring.data_ptr(), self._capacity, self._layout.stride,
                                               C.byref(self._layout.c), self._head, self._count, start, B,
                                               dstp, odt, _lib.stream_ptr(self._device)))
    return outs

  def _gather_index(self, idx, out=None):
    """idx: int64 tensor [B] on the device; out: (outs, dstp, odt) from _alloc_out to reuse."""
    B = idx.numel()
    outs, dstp, odt = out if out is not None else self._alloc_out(B)
    if B:
      with torch.cuda.device(self._device):
        _lib.check(_lib.lib().trl_gather_rows(self._ring.data_ptr(), self._capacity, self._layout.stride,
                                              C.byref(self._layout.c), self._head, self._count,
                                              idx.data_ptr(), B, dstp, odt, _lib.stream_ptr(self._device)))
    return outs

  # Host indices of at most this many rows are read by the gather kernel straight out of pinned host
  # memory (one coalesced 8 B-per-lane load per wave over PCIe); larger index arrays are copied first.
  # A small H2D copy is a DMA-engine job behind a cross-queue dependency on both sides: tens of
  # microseconds in the middle of a step that is otherwise ~100 us of kernels.
  ZERO_COPY_MAX_ROWS = 8192

  def _stage_host_index(self, index, zero_copy):
    """A HOST int64 index tensor -> something a kernel can read: a slot of a small ring of pinned staging buffers
    (reused only after the launch that read it has finished).  zero_copy: hand the pinned slot itself to the kernel
    (mapped into the device's address space: one coalesced read over PCIe); else a copy kernel reads it into device
    memory first (trl_stage_indices: still no DMA-engine copy in the stream).  Returns (idx, event): record the event
    behind the consuming launch."""
    n = index.numel()
    if self._pin is None or self._pin[0][0].numel() < n:
      if self._pin is not None:
        # a queued gather may still be reading its indices out of the old pinned buffers (zero-copy):
        # wait for every slot that was ever used before the buffers go back to the host allocator
        for (_, ev), used in zip(self._pin, self._pin_used):
          if used:
            ev.synchronize()
      cap = max(4096, 1 << (max(n, 1) - 1).bit_length())
      self._pin = [(torch.empty(cap, dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(4)]
      self._pin_next = 0
      self._pin_used = [False] * 4
    slot = self._pin_next
    self._pin_next = (slot + 1) % len(self._pin)
    buf, ev = self._pin[slot]
    if self._pin_used[slot]:
      ev.synchronize()            # the launch issued 4 calls ago has long finished
    buf[:n].copy_(index)
    self._pin_used[slot] = True
    if zero_copy and n <= self.ZERO_COPY_MAX_ROWS and config.index_zero_copy:
      return buf[:n], ev          # pinned host memory is mapped into the device's address space
    idx = torch.empty(n, dtype=torch.int64, device=self._device)
    with torch.cuda.device(self._device):
      _lib.check(_lib.lib().trl_stage_indices(buf.data_ptr(), idx.data_ptr(), n, _lib.stream_ptr(self._device)))
    return idx, ev

  def _gather_host_index(self, index):
    """Gather the rows listed by a HOST int64 tensor."""
    # (zero-copy for narrow records only: there one wave per 64 rows reads the indices once; the wide kernel reads a
    # row's index in every workgroup that copies one of its chunks, which must come from HBM / L2)
    idx, ev = self._stage_host_index(index, zero_copy=not self._layout.has_wide)
    outs = self._gather_index(idx)
    ev.record(torch.cuda.current_stream(self._device))
    return outs

  # ------------------------------------------------------------------ rows in place (no gather)
  def sample_indices(self, batch_size, generator=None):
    """The index draw of sample() alone: `torch.randperm(len(self), generator=generator)[:batch_size]` (examples/dqn.py:40)
    bit for bit, as a host int64 tensor -- for a learner that reads the sampled rows in place
    (DQNController.learn_indexed) instead of gathering them first."""
    ahead = self._count if (generator is None and self._count == self._capacity) else None
    return randperm_prefix(self._count, batch_size, generator, next_n=ahead)

  def ring_view(self, index):
    """What a kernel needs to read the rows `index` (int64 tensor, host or device, logical row numbers as
    `self[index]` takes them) where they lie: dict(ring, capacity, stride, layout, head, count, idx, done) --
    idx a tensor the device can read; call done() right after enqueueing the consumer."""
    if self._layout is None or self._count == 0:
      raise TypeError("'NoneType' object is not iterable")
    if not torch.is_tensor(index) or index.dtype != torch.int64 or index.dim() != 1:
      raise TypeError('ring_view takes a 1-D int64 tensor of row numbers')
    n = self._count
    if index.is_cuda:
      idx, done = index.to(self._device).contiguous(), (lambda: None)
    else:
      if index.numel() and (int(index.min()) < -n or int(index.max()) >= n):
        raise IndexError('index out of range for %d rows' % n)
      idx, ev = self._stage_host_index(index.contiguous(), zero_copy=True)
      done = lambda: ev.record(torch.cuda.current_stream(self._device))
    return dict(ring=self._ring, capacity=self._capacity, stride=self._layout.stride, layout=self._layout, head=self._head,
                count=self._count, idx=idx, done=done)

  def __getitems__(self, indices):
    """torch's batched-fetch hook (torch.utils.data's fetcher calls it with a minibatch's list of row
    numbers when it exists): a `DataLoader(ds, batch_size)` minibatch (examples/ppo.py:66-73) is ONE
    gather launch -- a range gather for the sequential sampler's consecutive rows -- instead of one
    launch per row.  Returns the per-row samples the default collate expects; they are views of the
    one gathered batch, so collation is a single stack per column."""
    idx = [int(i) for i in indices]
    n = self._count
    if not idx:
      return []
    if idx[-1] - idx[0] == len(idx) - 1 and idx[0] >= 0 and idx[-1] < n and \
        all(b - a == 1 for a, b in zip(idx, idx[1:])):
      cols = self._gather_range(idx[0], len(idx))
    else:
      cols = self[torch.tensor(idx, dtype=torch.int64)]
    return [[c[r] for c in cols] for r in range(len(idx))]

  def __getitem__(self, index) -> List[torch.Tensor]:
    if self._layout is None or self._count == 0:
      if isinstance(index, slice) and self._layout is not None:
        return self._gather_range(0, 0)
      raise TypeError("'NoneType' object is not iterable")   # what the reference raises when empty
    n = self._count
    if isinstance(index, slice):
      start, stop, step = index.indices(n)
      if step == 1:
        return self._gather_range(start, max(0, stop - start))
      index = torch.arange(start, stop, step, dtype=torch.int64)
    if isinstance(index, (int, np.integer)):
      i = int(index)
      if i < -n or i >= n:
        raise IndexError('index %d is out of bounds for dimension 0 with size %d' % (i, n))
      return [o[0] for o in self._gather_range(i % n, 1)]
    if isinstance(index, (list, tuple, np.ndarray)):
      index = torch.as_tensor(np.asarray(index))
    if not torch.is_tensor(index) or index.dtype not in (torch.int64, torch.int32, torch.int16, torch.uint8,
                                                         torch.int8):
      raise TypeError('index must be an int, a slice or an integer tensor, got %r' % (type(index),))
    if index.dim() == 0:
      return self.__getitem__(int(index))
    if index.dim() != 1:
      raise IndexError('only 1-D index tensors are supported')
    if index.dtype != torch.int64:
      index = index.to(torch.int64)
    if index.is_cuda:
      # device-resident indices are not validated (no host sync); out-of-range clamps
      idx = index.to(self._device).contiguous()
    else:
      if index.numel() and (int(index.min()) < -n or int(index.max()) >= n):
        raise IndexError('index out of range for %d rows' % n)
      return self._gather_host_index(index.contiguous())
    return self._gather_index(idx)

  def sample(self, batch_size, generator=None):
    """`self[torch.randperm(len(self), generator=generator)[:batch_size]]` (examples/dqn.py:40-41):
    identical indices, identical generator state afterwards."""
    # a full ring is sampled with the same n every step: let the library draw the next prefix ahead of time
    # (default generator only: the helper keeps one speculation per process)
    ahead = self._count if (generator is None and self._count == self._capacity) else None
    idx = randperm_prefix(self._count, batch_size, generator, next_n=ahead)
    return self._gather_host_index(idx)

  def sample_device(self, batch_size, seed=None):
    """Throughput mode: `batch_size` distinct rows drawn uniformly ON THE DEVICE (trl_sample_indices,
    Philox) and gathered, with no host work per call beyond two launches -- the step can be queued
    or graph-replayed.  Same distribution as `sample()`, NOT the same indices as
    `torch.randperm` (for parity use `sample()`).  `seed` (first call only) defaults to
    torch.initial_seed().  Returns the batch; the indices stay in `self.last_indices`."""
    if self._layout is None or self._count == 0:
      raise TypeError("'NoneType' object is not iterable")
    B = int(batch_size)
    if getattr(self, '_dev_rng', None) is None:
      self._dev_seed = int(torch.initial_seed() if seed is None else seed) & (2 ** 64 - 1)
      self._dev_rng = torch.zeros(1, dtype=torch.int64, device=self._device)     # the call counter
    idx = torch.empty(B, dtype=torch.int64, device=self._device)
    with torch.cuda.device(self._device):
      _lib.check(_lib.lib().trl_sample_indices(self._count, B, C.c_uint64(self._dev_seed), self._dev_rng.data_ptr(),
                                               idx.data_ptr(), _lib.stream_ptr(self._device)))
    self.last_indices = idx
    return self._gather_index(idx)

  def sample_indices_device(self, batch_size, seed=None):
    """The index draw of sample_device() alone (trl_sample_indices: distinct, uniform, Philox on the device -- throughput
    mode, not torch.randperm's indices): an int64 device tensor for DQNController.learn_indexed."""
    if self._layout is None or self._count == 0:
      raise TypeError("'NoneType' object is not iterable")
    B = int(batch_size)
    if getattr(self, '_dev_rng', None) is None:
      self._dev_seed = int(torch.initial_seed() if seed is None else seed) & (2 ** 64 - 1)
      self._dev_rng = torch.zeros(1, dtype=torch.int64, device=self._device)     # the call counter
    idx = torch.empty(B, dtype=torch.int64, device=self._device)
    with torch.cuda.device(self._device):
      _lib.check(_lib.lib().trl_sample_indices(self._count, B, C.c_uint64(self._dev_seed), self._dev_rng.data_ptr(),
                                               idx.data_ptr(), _lib.stream_ptr(self._device)))
    self.last_indices = idx
    return idx

  def iter_batches(self, batch_size):
    """Sequential minibatches, last partial batch kept: the order of
    `DataLoader(ds, batch_size=...)` at examples/ppo.py:68 without per-row collation."""
    for start in range(0, self._count, batch_size):
      yield self._gather_range(start, min(batch_size, self._count - start))

  def truncate(self):
    """Drop every row (storage.py:47-48); the allocation is kept for the next rollout."""
    self._head = 0
    self._count = 0


class TransitionTupleDataset(TensorTupleDataset):
  """Replay of `Transition`s (reference: storage.py:60-72).

  obs_dtype=torch.uint8 stores obs/next_obs as uint8 (4x the capacity for Atari-shaped
  frames); gathered batches are float32 either way, bit-identical to what the reference
  returns for uint8 observations.
  """

  def __init__(self, size: int = -1, device=None, obs_dtype=None, gather_obs_dtype=None):
    sd = gd = None
    if obs_dtype is not None:
      sd = [obs_dtype, torch.float32, torch.float32, obs_dtype, torch.float32]
      g = gather_obs_dtype or torch.float32
      gd = [g, torch.float32, torch.float32, g, torch.float32]
    super().__init__(size=size, device=device, storage_dtypes=sd, gather_dtypes=gd)

  def extend(self, transition_list: List[Transition]):  # pylint: disable=arguments-differ
    if len(transition_list) == 0:
      return
    cols = []
    for ci, field in enumerate(zip(*transition_list)):
      if torch.is_tensor(field[0]):
        b = torch.stack([f.detach() for f in field])
      else:
        b = torch.from_numpy(np.asarray(field))
      want = self._storage_dtypes[ci] if self._storage_dtypes else torch.float32
      if b.dtype != want:
        b = b.to(want)       # torch.Tensor(b) -> float32 (storage.py:65-67)
      if b.dim() == 1:
        b = b.unsqueeze(-1)  # storage.py:68-71
      cols.append(b)
    super().extend(*cols)
