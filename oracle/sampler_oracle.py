"""`torch.randperm(n)[:B]` restated on numpy's MT19937 (TEST INFRASTRUCTURE).

The reference samples replay indices with torch.randperm on the default CPU generator
(examples/dqn.py:40, examples/ddpg.py:40).  torch (third party, the installed 2.10.0; the
reference leaves it unpinned, setup.py:42) implements that for n < 2^32/20 as
    for i in range(n - 1): z = mt19937_u32() % (n - i); swap(r[i], r[i + z])
seeded by init_genrand(seed & 0xffffffff) -- the same engine and seeding as
numpy.random.RandomState(seed), whose raw 32-bit words `random_raw` exposes.
"""
import numpy as np


class Mt19937Stream:
  def __init__(self, seed):
    self._rs = np.random.RandomState(int(seed) & 0xffffffff)

  def words(self, k):
    return self._rs._bit_generator.random_raw(int(k)).astype(np.uint64) if k else np.zeros(0, np.uint64)


def randperm(stream, n):
  """Full permutation, exactly torch's loop (n - 1 draws)."""
  r = np.arange(n, dtype=np.int64)
  if n > 1:
    w = stream.words(n - 1)
    z = (w % (n - np.arange(n - 1, dtype=np.uint64))).astype(np.int64)
    for i in range(n - 1):
      j = i + z[i]
      r[i], r[j] = r[j], r[i]
  return r


def randperm_prefix(stream, n, B):
  """First B entries of randperm(n), consuming n - 1 draws like the full call."""
  if n <= 1:
    return np.arange(n, dtype=np.int64)[:B]
  w = stream.words(n - 1)
  live = min(B, n - 1)
  sparse = {}
  out = np.empty(B, dtype=np.int64)
  for i in range(live):
    j = i + int(w[i] % np.uint64(n - i))
    vi, vj = sparse.get(i, i), sparse.get(j, j)
    out[i] = vj
    sparse[j] = vi
  for i in range(live, B):
    out[i] = sparse.get(i, i)
  return out
