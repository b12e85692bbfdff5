"""Minimal observation/action space descriptions (the attributes the examples read:
`.shape`, `.n`, `.low`, `.high`, `.sample()`; examples/dqn.py:31-32, examples/ddpg.py:29-32)."""
import numpy as np


class Box:
  def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
    self.dtype = dtype
    self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape if shape is not None else np.shape(low)).copy()
    self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.low.shape).copy()
    self.shape = self.low.shape
    self._rs = np.random.RandomState(seed)

  def seed(self, seed):
    self._rs = np.random.RandomState(seed)

  def sample(self):
    lo = np.where(np.isfinite(self.low), self.low, -1.0)
    hi = np.where(np.isfinite(self.high), self.high, 1.0)
    return self._rs.uniform(lo, hi).astype(self.dtype)

  def contains(self, x):
    x = np.asarray(x)
    return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete:
  def __init__(self, n, seed=None):
    self.n = int(n)
    self.shape = ()
    self._rs = np.random.RandomState(seed)

  def seed(self, seed):
    self._rs = np.random.RandomState(seed)

  def sample(self):
    return int(self._rs.randint(self.n))

  def contains(self, x):
    return 0 <= int(x) < self.n
