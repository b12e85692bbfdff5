"""HIP replay ring vs the reference's FIFO semantics: golden fixtures, the numpy oracle, and
size-independent properties at large sizes.  Bit-exact throughout (gathers move bytes)."""
import numpy as np
import pytest
import torch

from oracle import replay_oracle
from torchrl_amd.utils.storage import TensorTupleDataset, TransitionTupleDataset, Transition
from tests.helpers import FIELDS, transitions_from

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cpu(cols):
  return [c.cpu().numpy() for c in cols]


def test_golden_scenarios(golden):
  g = golden('storage.npz')
  for si in range(int(g['n_scenarios'])):
    p = 's%d_' % si
    ds = TransitionTupleDataset(size=int(g[p + 'size']), device=DEV)
    assert len(ds) == 0
    for pi, _k in enumerate(g[p + 'pushes']):
      ds.extend(transitions_from(g, '%spush%d_' % (p, pi)))
      assert len(ds) == int(g[p + 'lens'][pi])
      if (p + 'after%d_obs' % pi) in g.files:
        got = _cpu(ds[:])
        for fi, f in enumerate(FIELDS):
          exp = g['%safter%d_%s' % (p, pi, f)]
          assert got[fi].dtype == np.float32 and got[fi].shape == exp.shape, (si, pi, f)
          np.testing.assert_array_equal(got[fi], exp)
    idx = torch.from_numpy(np.array(g[p + 'idx']))
    for index in (idx, idx.to(DEV)):
      got = _cpu(ds[index])
      for fi, f in enumerate(FIELDS):
        np.testing.assert_array_equal(got[fi], g[p + 'gather_' + f])
    got = _cpu(ds[0])
    for fi, f in enumerate(FIELDS):
      np.testing.assert_array_equal(got[fi], g[p + 'item0_' + f])
    ds.truncate()
    assert len(ds) == 0
  ds = TransitionTupleDataset(size=3, device=DEV)
  ds.extend([])
  assert len(ds) == 0


def test_uint8_storage_matches_float_reference(golden):
  """Scenario 5 pushes uint8 frames; stored as uint8 here, gathered as float32 == reference."""
  g = golden('storage.npz')
  p = 's5_'
  ds = TransitionTupleDataset(size=int(g[p + 'size']), device=DEV, obs_dtype=torch.uint8)
  for pi, _k in enumerate(g[p + 'pushes']):
    ds.extend(transitions_from(g, '%spush%d_' % (p, pi)))
    got = _cpu(ds[:])
    for fi, f in enumerate(FIELDS):
      np.testing.assert_array_equal(got[fi], g['%safter%d_%s' % (p, pi, f)])
  raw = TransitionTupleDataset(size=6, device=DEV, obs_dtype=torch.uint8, gather_obs_dtype=torch.uint8)
  raw.extend(transitions_from(g, p + 'push0_'))
  assert raw[:][0].dtype == torch.uint8
  np.testing.assert_array_equal(raw[:][0].cpu().numpy(), g[p + 'push0_obs'])


@pytest.mark.parametrize('n', [10, 20, 30, 40, 50])
def test_unbounded_shapes(n):
  """The reference's own test_storage.py::test_unbounded, on the device ring."""
  ds = TensorTupleDataset(size=-1, device=DEV)
  assert len(ds) == 0
  x = [torch.randn(n, 2, 19), torch.randn(n, 4), torch.randn(n, 8), torch.randn(n, 16), torch.randn(n, 32)]
  ds.extend(*x)
  assert len(ds) == n
  i = int(np.random.randint(0, n))
  for v, ref in zip(ds[i], x):
    assert v.dim() == len(ref.shape[1:]) and v.size(-1) == ref.size(-1)
    assert torch.equal(v.cpu(), ref[i])
  ds.truncate()
  assert len(ds) == 0


@pytest.mark.parametrize('n,size', [(10, 9), (20, 100), (30, 10), (40, 40), (50, 12)])
def test_bounded_shapes(n, size):
  """The reference's own test_storage.py::test_bounded, plus FIFO values."""
  ds = TensorTupleDataset(size=size, device=DEV)
  assert ds.size == size and len(ds) == 0
  x = [torch.randn(n, 2), torch.randn(n, 4), torch.randn(n, 8, 12), torch.randn(n, 16), torch.randn(n, 32)]
  ds.extend(*x)
  assert len(ds) == min(n, size)
  got = ds[:]
  for v, ref in zip(got, x):
    assert torch.equal(v.cpu(), ref[-size:] if n > size else ref)
  ds.truncate()
  assert len(ds) == 0


def _random_transitions(rs, k, obs_shape, u8):
  out = []
  for _ in range(k):
    if u8:
      o, no = rs.randint(0, 256, obs_shape).astype(np.uint8), rs.randint(0, 256, obs_shape).astype(np.uint8)
    else:
      o, no = rs.randn(*obs_shape).astype(np.float32), rs.randn(*obs_shape).astype(np.float32)
    out.append(Transition(o, np.int64(rs.randint(0, 6)), float(rs.randn()), no, bool(rs.rand() < 0.1)))
  return out


@pytest.mark.parametrize('obs_shape,u8,size', [((3,), False, 1000), ((4,), False, 257), ((84, 84, 4), True, 96),
                                               ((5,), True, 50), ((300,), False, 64), ((7, 5), True, 33)])
def test_random_pushes_vs_oracle(obs_shape, u8, size):
  """Many wrapping pushes of varying k against the numpy FIFO, incl. Atari-shaped uint8
  records (wide-column kernel) and ragged uint8 widths."""
  rs = np.random.RandomState(hash((obs_shape, size)) & 0xffff)
  ds = TransitionTupleDataset(size=size, device=DEV, obs_dtype=torch.uint8 if u8 else None)
  ref = replay_oracle.FifoColumns(size=size)
  for pi, k in enumerate([1, 3, size // 2, size - 1, 2, size, 5, size + 7, 1]):
    tl = _random_transitions(rs, max(k, 1), obs_shape, u8)
    if pi % 2 == 0:
      ds.extend(tl)                       # host rows: pinned record staging + H2D copy
    else:                                 # device rows: the scatter kernel (k_ring_push / k_ring_push16)
      cols = []
      for ci, f in enumerate(zip(*tl)):
        a = torch.from_numpy(np.asarray(f))
        a = a if (u8 and ci in (0, 3)) else a.float()
        cols.append((a.unsqueeze(-1) if a.dim() == 1 else a).to(DEV))
      TensorTupleDataset.extend(ds, *cols)
    ref.extend_transitions(tl)
    assert len(ds) == len(ref)
    n = len(ref)
    idx = rs.randint(-n, n, size=2 * n + 3).astype(np.int64)
    for got, exp in zip(_cpu(ds[torch.from_numpy(idx)]), ref[idx]):
      np.testing.assert_array_equal(got, exp)
    for got, exp in zip(_cpu(ds[:]), ref[slice(None)]):
      np.testing.assert_array_equal(got, exp)
    for got, exp in zip(_cpu(ds[1:n:3]), ref[slice(1, n, 3)]):
      np.testing.assert_array_equal(got, exp)


def test_unbounded_growth_keeps_order():
  ds = TensorTupleDataset(size=-1, device=DEV)
  ref = replay_oracle.FifoColumns(size=-1)
  rs = np.random.RandomState(3)
  for k in [1, 1000, 30, 2000, 5000]:
    a, b = rs.randn(k, 3).astype(np.float32), rs.randn(k, 1).astype(np.float32)
    ds.extend(torch.from_numpy(a), torch.from_numpy(b))
    ref.extend(a, b)
  assert len(ds) == len(ref) == 8031
  for got, exp in zip(_cpu(ds[:]), ref[slice(None)]):
    np.testing.assert_array_equal(got, exp)


def test_sample_matches_reference_sampling_line():
  """buffer.sample(B) == buffer[torch.randperm(len(buffer))[:B]] (examples/dqn.py:40-41),
  same indices and same generator state afterwards."""
  rs = np.random.RandomState(0)
  ds = TransitionTupleDataset(size=5000, device=DEV)
  ds.extend(_random_transitions(rs, 5000, (4,), False))
  torch.manual_seed(123)
  idx = torch.randperm(len(ds))[:256]
  exp = _cpu(ds[idx])
  after = torch.rand(2)
  torch.manual_seed(123)
  got = _cpu(ds.sample(256))
  for a, b in zip(got, exp):
    np.testing.assert_array_equal(a, b)
  assert torch.equal(torch.rand(2), after)


def test_errors_are_loud():
  ds = TransitionTupleDataset(size=4, device=DEV)
  ds.extend(_random_transitions(np.random.RandomState(1), 3, (2,), False))
  with pytest.raises(IndexError):
    ds[3]
  with pytest.raises(IndexError):
    ds[torch.tensor([0, 5])]
  with pytest.raises(TypeError):        # float64 into an existing float32 column (the one promotion not reproduced)
    f32 = TensorTupleDataset(device=DEV)
    f32.extend(torch.zeros(2, 2))
    f32.extend(torch.zeros(2, 2, dtype=torch.float64))
  with pytest.raises(TypeError):
    TensorTupleDataset(device=DEV).extend(torch.zeros(2, 2, dtype=torch.complex64))
  with pytest.raises(Exception):
    TensorTupleDataset(device='cpu').extend(torch.zeros(2, 2))


def test_dtype_promotion_golden(golden):
  """storage.py:35 + torch.cat promotion: a float64 column stays float64, int64 becomes float32 (the
  reference's own outputs, tests/golden/storage.npz tt_a / tt_b: size 4, six rows pushed)."""
  g = golden('storage.npz')
  ds = TensorTupleDataset(size=4, device=DEV)
  ds.extend(torch.arange(12, dtype=torch.float64).reshape(6, 2), torch.arange(6, dtype=torch.int64).reshape(6, 1))
  a, b = ds[:]
  assert a.dtype == torch.float64 and b.dtype == torch.float32 and len(ds) == 4
  np.testing.assert_array_equal(a.cpu().numpy(), g['tt_a'])
  np.testing.assert_array_equal(b.cpu().numpy(), g['tt_b'])
  assert g['tt_a'].dtype == np.float64
  # wrapping pushes of a float64 column (values no float32 holds), from the host and from the device
  ref = replay_oracle.FifoColumns(size=7)
  ds = TensorTupleDataset(size=7, device=DEV)
  rs = np.random.RandomState(1)
  for pi, k in enumerate([3, 5, 2, 9, 1]):
    x, y = rs.randn(k, 3) + 1e-12, rs.randn(k, 1).astype(np.float32)
    ds.extend(torch.from_numpy(x).to(DEV) if pi % 2 else torch.from_numpy(x), torch.from_numpy(y))
    ref.extend(x, y)
    got = ds[:]
    assert got[0].dtype == torch.float64
    for gt, ex in zip(_cpu(got), ref[slice(None)]):
      np.testing.assert_array_equal(gt, ex)
    idx = rs.randint(0, len(ds), size=11).astype(np.int64)
    for gt, ex in zip(_cpu(ds[torch.from_numpy(idx)]), ref[idx]):
      np.testing.assert_array_equal(gt, ex)


def test_dataloader_contract_one_gather_per_minibatch(monkeypatch):
  """examples/ppo.py:66-73: `DataLoader(TensorTupleDataset(x=[8 columns]), batch_size)` -- sequential,
  un-shuffled, last partial batch kept.  Same minibatches as iter_batches(), one gather launch each
  (torch's batched-fetch hook __getitems__), not one per row; shuffled loaders take the index gather."""
  from torch.utils.data import DataLoader
  from torchrl_amd import _lib
  n, B = 1000, 64
  g = torch.Generator(device=DEV).manual_seed(3)
  cols = [torch.randn(n, w, device=DEV, generator=g) for w in (3, 1, 1, 3, 1, 1, 1, 1)]   # obs .. advantages
  ds = TensorTupleDataset(x=cols)
  assert len(ds) == n and ds.device.type == 'cuda'
  lib = _lib.lib()
  calls = {'range': 0, 'rows': 0}
  real_range, real_rows = lib.trl_gather_range, lib.trl_gather_rows

  def count_range(*a):
    calls['range'] += 1
    return real_range(*a)

  def count_rows(*a):
    calls['rows'] += 1
    return real_rows(*a)
  monkeypatch.setattr(lib, 'trl_gather_range', count_range)
  monkeypatch.setattr(lib, 'trl_gather_rows', count_rows)
  batches = list(DataLoader(ds, batch_size=B))
  nb = (n + B - 1) // B
  assert len(batches) == nb and calls == {'range': nb, 'rows': 0}
  for bi, (mb, ref) in enumerate(zip(batches, ds.iter_batches(B))):
    lo, hi = bi * B, min(n, (bi + 1) * B)
    assert len(mb) == 8
    for c, (got, exp) in enumerate(zip(mb, ref)):
      assert got.is_cuda and got.shape == (hi - lo, cols[c].shape[1])
      assert torch.equal(got, exp) and torch.equal(got, cols[c][lo:hi])
  # a shuffling sampler: one index gather per minibatch, rows as the sampler drew them
  calls.update(range=0, rows=0)
  order = torch.randperm(n, generator=torch.Generator().manual_seed(0)).tolist()
  shuffled = list(DataLoader(ds, batch_size=B, sampler=order))       # any order a sampler draws
  assert calls == {'range': 0, 'rows': nb}
  got0 = torch.cat([mb[0] for mb in shuffled])
  assert torch.equal(got0, cols[0][torch.tensor(order, device=DEV)])
  # single rows still work (test_storage.py:25 indexes with an int)
  assert torch.equal(ds[5][0], cols[0][5])


def test_large_ring_properties():
  """1M-row Pendulum-shaped ring (BASELINE config 3): every gathered row must equal the row
  that was pushed at that logical position, checked through a closed form of the content."""
  N, B = 1_000_000, 65536
  ds = TensorTupleDataset(size=N, device=DEV)
  base = torch.arange(N + 1234, dtype=torch.float32, device=DEV)
  def cols(lo, hi):
    r = base[lo:hi]
    return [torch.stack([r, r + 0.25, r + 0.5], 1), (r * 2).unsqueeze(1), (-r).unsqueeze(1),
            torch.stack([r + 1, r + 1.25, r + 1.5], 1), (r % 2).unsqueeze(1)]
  ds.extend(*cols(0, 600_000))
  ds.extend(*cols(600_000, N + 1234))          # wraps: oldest surviving row is 1234
  assert len(ds) == N
  torch.manual_seed(5)
  idx = torch.randint(0, N, (B,), device=DEV)
  obs, act, rew, nobs, done = ds[idx]
  r = (idx + 1234).float()
  assert torch.equal(obs[:, 0], r) and torch.equal(obs[:, 2], r + 0.5)
  assert torch.equal(act[:, 0], r * 2) and torch.equal(rew[:, 0], -r)
  assert torch.equal(nobs[:, 1], r + 1.25) and torch.equal(done[:, 0], r % 2)


def test_atari_ring_past_4gib_offsets_and_wrap():
  """BASELINE config 4 at size: a 100,000-row ring of Atari-shaped uint8 records (56,480 B each = 5.6 GB, so byte
  offsets pass 2^32 at record 76,045) that has wrapped.  Every gathered value is checked against a closed form of
  (global row id, column, position), at indices on both sides of the wrap, on both sides of the 4 GiB offset, at the
  ends and at random: index gather (device and host indices), range gather across the boundary, the uint8 -> float32
  cast path and the 12 B narrow tail that rides in the same launch."""
  cap, extra, K = 100_000, 3_333, 28224
  free, _ = torch.cuda.mem_get_info()
  if free < 9 * (1 << 30):
    pytest.skip('needs 9 GiB of free HBM')
  dt = [torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32]
  ds = TensorTupleDataset(size=cap, device=DEV, storage_dtypes=dt, gather_dtypes=dt)
  j = torch.arange(K, device=DEV, dtype=torch.int64).view(1, K)

  def frames(rows, a, b):       # rows: int64 global row ids [k] -> uint8 [k, 84, 84, 4]
    return ((rows.view(-1, 1) * a + j * b + (j >> 8)) & 255).to(torch.uint8).view(-1, 84, 84, 4)

  def tail(rows):
    r = rows.view(-1, 1).float()
    return (r % 6), (r % 3) - 1.0, (rows.view(-1, 1) % 200 == 0).float()

  for lo in range(0, cap + extra, 4096):
    rows = torch.arange(lo, min(lo + 4096, cap + extra), device=DEV, dtype=torch.int64)
    act, rew, done = tail(rows)
    ds.extend(frames(rows, 7, 13), act, rew, frames(rows, 11, 5), done)
  assert len(ds) == cap and ds.row_bytes * cap > (1 << 32) and ds._head == extra
  stride = ds.row_bytes
  p32 = (1 << 32) // stride                       # the physical record whose bytes straddle 2^32
  special = []
  for phys in (0, 1, cap - 1, cap - 2, p32 - 1, p32, p32 + 1, p32 + 2):
    special.append((phys - extra) % cap)          # logical index of that physical record
  special += [0, 1, cap - 1, -1, -cap]
  g = torch.Generator().manual_seed(11)
  idx = torch.cat([torch.tensor(special, dtype=torch.int64), torch.randint(0, cap, (1500,), generator=g)])
  rows = (idx % cap) + extra                      # global row id of logical index i (0 = oldest survivor)
  exp = (frames(rows.to(DEV), 7, 13),) + tail(rows.to(DEV))[:2] + (frames(rows.to(DEV), 11, 5), tail(rows.to(DEV))[2])
  for index in (idx.to(DEV), idx):                # device indices, then host indices (staged through pinned memory)
    got = ds[index]
    for c, (a, b) in enumerate(zip(got, exp)):
      assert a.dtype == b.dtype and torch.equal(a, b), 'column %d' % c
  # a contiguous logical range across the 4 GiB boundary and across the wrap
  for lo in ((p32 - extra) % cap - 40, cap - extra - 40):
    got = ds[lo:lo + 80]
    rr = torch.arange(lo, lo + 80, device=DEV, dtype=torch.int64) + extra
    assert torch.equal(got[0], frames(rr, 7, 13)) and torch.equal(got[3], frames(rr, 11, 5))
    assert torch.equal(got[1], tail(rr)[0])
  # the cast path (frames handed out as float32, the reference's dtype) on the special indices
  ds._gather_dtypes = [torch.float32] * 5
  got = ds[idx[:len(special)].to(DEV)]
  assert got[0].dtype == torch.float32 and torch.equal(got[0], exp[0][:len(special)].float())
  assert torch.equal(got[3], exp[3][:len(special)].float())
  del ds
  torch.cuda.empty_cache()


def test_device_sampler_distinct_uniform_reproducible():
  """Throughput-mode sampler (trl_sample_indices): distinct indices in range, a function of
  (seed, call number) only, uniform over the rows; the gathered batch is ds[indices]."""
  dev = torch.device(DEV)
  n = 5000
  ds = TensorTupleDataset(size=n, device=dev)
  cols = [torch.arange(n, dtype=torch.float32, device=dev).view(n, 1), torch.randn(n, 3, device=dev)]
  ds.extend(*cols)
  runs = []
  for _ in range(2):
    ds._dev_rng = None
    seq = []
    for _call in range(3):
      batch = ds.sample_device(256, seed=1234)
      idx = ds.last_indices.cpu()
      assert idx.numel() == 256 and idx.unique().numel() == 256 and int(idx.min()) >= 0 and int(idx.max()) < n
      assert torch.equal(batch[0].view(-1).cpu(), idx.float())           # row i of column 0 holds i
      assert torch.equal(batch[1].cpu(), cols[1].cpu()[idx])
      seq.append(idx)
    runs.append(seq)
  for a, b in zip(*runs):
    assert torch.equal(a, b)                                             # same seed, same call number
  assert not torch.equal(runs[0][0], runs[0][1])                          # successive calls differ
  # every row when B == n (maximal collision pressure), and the largest batch
  small = TensorTupleDataset(size=300, device=dev)
  small.extend(torch.arange(300, dtype=torch.float32, device=dev).view(300, 1))
  small.sample_device(300, seed=7)
  assert torch.equal(small.last_indices.sort().values.cpu(), torch.arange(300))
  ds.sample_device(4096, seed=5)
  assert ds.last_indices.unique().numel() == 4096
  with pytest.raises(Exception):
    ds.sample_device(4097)
  # uniformity: 400 calls x 250 draws over 50 buckets of 100 rows, chi-square with 49 dof
  counts = torch.zeros(50)
  ds._dev_rng = None
  for _ in range(400):
    ds.sample_device(250, seed=99)
    counts += torch.bincount((ds.last_indices // 100).cpu(), minlength=50).float()
  expect = 400 * 250 / 50
  chi2 = float(((counts - expect) ** 2 / expect).sum())
  assert chi2 < 95.0, chi2        # p ~ 1e-4 for 49 dof
