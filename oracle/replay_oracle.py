"""FIFO replay columns in numpy (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows torchrl/utils/storage.py: truncated_cat (:7-16) concatenates and keeps the LAST
`maxsize` rows when maxsize > 0; TensorTupleDataset.extend (:27-37) applies it per column,
creating float32 columns lazily; __len__ (:39-42); __getitem__ (:44-45) indexes every
column; truncate (:47-48) drops them; TransitionTupleDataset.extend (:60-72) converts each
field of the transitions to float32 and gives 1-D fields a trailing unit axis.
"""
import numpy as np


def truncated_cat(a, b, maxsize=-1):
  res = np.concatenate([a, b], axis=0)
  diff = res.shape[0] - maxsize
  if maxsize > 0 and diff > 0:
    res = res[diff:]
  return res


class FifoColumns:
  def __init__(self, size=-1):
    self.size = size
    self.cols = None

  def extend(self, *arrays):
    if self.cols is None:
      self.cols = [None] * len(arrays)
    assert len(self.cols) == len(arrays)
    for i, new in enumerate(arrays):
      new = np.asarray(new)
      if self.cols[i] is None:
        self.cols[i] = np.zeros((0,) + new.shape[1:], dtype=np.float32)
      # torch.cat's promotion of the float32 column with the pushed dtype (:35, :10): float64 wins (the
      # whole column becomes float64), every integer / bool / half dtype gives float32
      dt = np.float64 if np.float64 in (new.dtype, self.cols[i].dtype) else np.float32
      self.cols[i] = truncated_cat(self.cols[i].astype(dt), new.astype(dt), self.size)

  def extend_transitions(self, transitions):
    if len(transitions) == 0:
      return
    batch = [np.asarray(f).astype(np.float32) for f in zip(*transitions)]
    batch = [b[:, None] if b.ndim == 1 else b for b in batch]
    self.extend(*batch)

  def __len__(self):
    return 0 if self.cols is None else self.cols[0].shape[0]

  def __getitem__(self, index):
    return [c[index] for c in self.cols]

  def truncate(self):
    self.cols = None
