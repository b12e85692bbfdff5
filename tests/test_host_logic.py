"""Host-side pieces that need no device: schedules, exploration helpers, record layout,
controller construction, and the data-parallel gradient sync over gloo (world_size 2)."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torchrl_amd.utils.schedule import LinearSchedule, ExpDecaySchedule
from torchrl_amd.utils.storage import _Layout, TransitionTupleDataset
from torchrl_amd.contrib.controllers import DQNController, DDPGController, A2CController, PPOController
from torchrl_amd.contrib.controllers.dqn_controller import epsilon_greedy
from torchrl_amd.contrib.controllers.ddpg_controller import OUNoise, polyak_average_
from torchrl_amd.controllers import RandomController
from torchrl_amd.optim import FlatParams
from torchrl_amd.parallel import GradSync, shard_range
from torchrl_amd import _lib


def test_linear_schedule_reference_values():
  # the assertions of the reference's test_schedule.py:23-41
  s = LinearSchedule(min_val=0.1, max_val=1.0, num_steps=10)
  vals = [s.value for _ in range(6)]
  assert vals[0] == 0.1 and abs(vals[5] - 0.55) < 1e-12
  for _ in range(20):
    s.step()
  assert s.value == 1.0
  s = LinearSchedule(min_val=0.1, max_val=1.0, num_steps=10, invert=True)
  assert s.value == 1.0
  for _ in range(20):
    s.step()
  assert s.value == 0.1


def test_exp_decay_schedule_reference_values():
  s = ExpDecaySchedule(start=1.0, end=0.01, num_steps=5)
  assert s.value == 1.0
  assert abs(s.value - (0.01 + 0.99 * math.exp(-1 / 5))) < 1e-15
  for _ in range(1000):
    s.step()
  assert abs(s.value - 0.01) < 1e-6
  assert repr(s) == str(s.wrapped_value)


def test_epsilon_greedy_consumes_numpy_stream_like_reference():
  np.random.seed(3)
  got = epsilon_greedy(4, np.array([0, 3, 2, 1, 0]), eps=0.3)
  np.random.seed(3)
  pmf = np.ones((5, 4), dtype=np.float32) * 0.3 / 4
  pmf[np.arange(5), [0, 3, 2, 1, 0]] += 0.7
  exp = np.array([np.random.choice(np.arange(4), p=row) for row in pmf])
  assert got.shape == (5, 1) and np.array_equal(got[:, 0], exp)
  assert np.array_equal(epsilon_greedy(3, np.array([2, 1]), eps=0.0)[:, 0], [2, 1])


def test_ou_noise_recurrence():
  np.random.seed(0)
  n = OUNoise(2)
  a = n.get_action(np.zeros((3, 2)))
  np.random.seed(0)
  x = np.zeros(2)
  x = x + (0.15 * (0.0 - x) + 0.3 * np.random.randn(2))
  assert np.allclose(a, np.broadcast_to(x, (3, 2))) and n.step == 1
  n.reset()
  assert n.step == 0 and np.all(n.state == 0)


def test_polyak_modules_and_random_controller():
  a, b = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
  exp = 0.1 * a.weight.data + 0.9 * b.weight.data
  polyak_average_(a, b, 0.1)
  assert torch.equal(b.weight.data, exp)
  with pytest.raises(AssertionError):
    polyak_average_(a, 3)

  class Space:
    def sample(self):
      return 1
  assert RandomController(Space()).act([0, 0, 0]) == [1, 1, 1]


def test_record_layouts():
  cart = _Layout([(4,), (1,), (1,), (4,), (1,)], [torch.float32] * 5)
  assert cart.payload == 44 and cart.stride == 64 and cart.offsets == [0, 16, 20, 24, 40]      # narrow records: padded to one 64 B request
  pend = _Layout([(3,), (1,), (1,), (3,), (1,)], [torch.float32] * 5)
  assert pend.payload == 36 and pend.stride == 64
  atari = _Layout([(84, 84, 4), (1,), (1,), (84, 84, 4), (1,)],
                  [torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32])
  assert atari.col_bytes[0] == 28224 and atari.stride % 16 == 0
  assert atari.offsets[0] % 16 == 0 and atari.offsets[3] % 16 == 0
  assert 56460 <= atari.stride <= 56460 + 48
  ragged = _Layout([(5,), (1,)], [torch.uint8, torch.float32])
  assert ragged.offsets == [0, 8] and ragged.stride == 12
  with pytest.raises(ValueError):
    _Layout([(1,)] * 9, [torch.float32] * 9)


def test_controllers_construct_and_act_on_cpu_but_refuse_to_learn():
  torch.manual_seed(0)
  np.random.seed(0)
  d = DQNController(4, 2, device='cpu')
  assert len(d.act([[0.0] * 4] * 3)) == 3
  g = DDPGController(3, 1, -2.0, 2.0, device='cpu')
  assert g.act([[0.0] * 3] * 2).shape == (2, 1)
  a = A2CController(4, 2, device='cpu')
  assert a.act([[0.0] * 4] * 2).shape == (2,)
  p = PPOController(3, 1, device='cpu')
  assert p.act([[0.0] * 3] * 2).shape == (2, 1)
  x = torch.zeros(4, 4)
  with pytest.raises(_lib.TrlError):
    d.learn(x, x[:, :1], x[:, :1], x, x[:, :1])
  with pytest.raises(_lib.TrlError):
    a.compute_return(x, x[:, :1], x[:, :1], x, x[:, :1])
  # target nets start as copies (dqn_controller.py:37)
  assert torch.equal(d._q.data, d._target.data) and d._q.data.data_ptr() != d._target.data.data_ptr()
  # parameters are views of the flat buffer, in parameters() order
  for prm, off in zip(d.q_net.parameters(), d._q.offsets):
    assert prm.data_ptr() == d._q.data.data_ptr() + 4 * off and off % 4 == 0   # 16 B-aligned starts
  assert torch.equal(d._q.dense(), torch.cat([p_.data.reshape(-1) for p_ in d.q_net.parameters()]))


def test_storage_without_device_fails_loudly():
  if torch.cuda.is_available():
    pytest.skip('a HIP device is present')
  ds = TransitionTupleDataset(size=4)
  with pytest.raises(_lib.TrlError, match='HIP device'):
    ds.extend([(np.zeros(2, np.float32), 0, 0.0, np.zeros(2, np.float32), False)])


def test_shard_range_partitions():
  for n, w in [(4096, 8), (10, 3), (7, 8), (1_000_000, 8)]:
    spans = [shard_range(n, r, w) for r in range(w)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _dp_worker(rank, world, port, out):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    torch.manual_seed(100 + rank)                      # ranks start from different weights
    net = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
    flat = FlatParams(net)
    sync = GradSync()
    sync.broadcast_params(flat)                        # ... and end up with rank 0's
    torch.manual_seed(7)
    x = torch.randn(8, 3)
    y = torch.randn(8, 1)
    lo, hi = shard_range(8, rank, world)               # each rank learns on its shard
    flat.zero_grad()
    torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
    scale = sync.all_reduce(flat)
    out[rank] = (flat.data.clone(), flat.grad.clone() * scale)
  finally:
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2_equals_full_batch():
  world, port = 2, _free_port()
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
  (w0, g0), (w1, g1) = out[0], out[1]
  assert torch.equal(w0, w1) and torch.equal(g0, g1)
  # single-process reference on the whole batch with the same weights
  net = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
  flat = FlatParams(net)
  flat.data.copy_(w0)
  torch.manual_seed(7)
  x = torch.randn(8, 3)
  y = torch.randn(8, 1)
  flat.zero_grad()
  torch.nn.functional.mse_loss(net(x), y).backward()
  assert torch.allclose(flat.grad, g0, rtol=1e-5, atol=1e-7)


def test_flat_params_fresh_grads_equal_accumulated():
  """optim.FlatParams(fresh_grads=True): detached .grad + one multi-tensor copy gives the flat
  gradient buffer that zero + autograd accumulation gives; unused parameters come out zero."""
  import copy
  from torchrl_amd.optim import FlatParams
  torch.manual_seed(0)
  net_a = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2), torch.nn.Linear(2, 2))
  net_b = copy.deepcopy(net_a)
  fa, fb = FlatParams(net_a), FlatParams(net_b, fresh_grads=True)
  x = torch.randn(7, 3)
  for f, net in ((fa, net_a), (fb, net_b)):
    for _ in range(2):                       # second pass: no leftovers from the first
      f.zero_grad()
      net[2](net[1](net[0](x))).pow(2).sum().backward()      # the last Linear is unused
      f.collect_grads()
  assert torch.equal(fa.grad, fb.grad) and float(fa.grad.abs().max()) > 0
  assert all(p.grad.data_ptr() == fb.grad.data_ptr() + 4 * off for p, off in zip(fb.params, fb.offsets))
  assert float(fb.params[-1].grad.abs().max()) == 0.0


# ---------------------------------------------------------------- bench.py --gpus N: the launcher path
def _fresh_bench():
  import importlib, sys as _sys
  _sys.modules.pop('bench', None)
  return importlib.import_module('bench')


def test_bench_gpus_n_starts_n_ranks_before_the_gpu_stack_loads(monkeypatch, capsys):
  """`python bench.py --gpus N` as ONE process (the driver's command shape) must become the launcher of N rank
  processes -- torch.distributed.run, loopback rendezvous -- before bench.py has imported torch or the package
  (nothing GPU-side may be loaded in a process that then starts the ranks), relay rank 0's JSON line and exit 0."""
  bench = _fresh_bench()
  monkeypatch.delenv('WORLD_SIZE', raising=False)
  seen = {}

  class FakeProc:
    def __init__(self, cmd, env=None, stdout=None, text=None):
      seen['cmd'], seen['env'] = cmd, env
      seen['torch_loaded_in_launcher'] = bench.torch is not None
      self.stdout = iter(['some library chatter\n', '{"metric": "m", "value": 1.0, "n_gpus": 4}\n'])

    def wait(self):
      return 0

  monkeypatch.setattr(bench.subprocess, 'Popen', FakeProc)
  monkeypatch.setattr(bench, '_load', lambda: (_ for _ in ()).throw(AssertionError('the launcher loaded the GPU stack')))
  rc = bench.main(['--gpus', '4', '--steps', '3', '--warmup', '1'])
  assert rc == 0
  cmd = seen['cmd']
  assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nnodes=1' in cmd
  assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
  assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
  assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
  assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
  assert seen['torch_loaded_in_launcher'] is False
  out = capsys.readouterr()
  assert out.out.strip() == '{"metric": "m", "value": 1.0, "n_gpus": 4}'      # ONE line on stdout, the chatter on stderr
  assert 'some library chatter' in out.err


def test_bench_launcher_reports_a_failed_rank(monkeypatch, capsys):
  bench = _fresh_bench()
  monkeypatch.delenv('WORLD_SIZE', raising=False)

  class FakeProc:
    def __init__(self, *a, **k):
      self.stdout = iter([])

    def wait(self):
      return 1

  monkeypatch.setattr(bench.subprocess, 'Popen', FakeProc)
  assert bench.main(['--gpus', '2']) != 0
  assert capsys.readouterr().out.strip() == ''


def test_bench_is_a_rank_under_torch_distributed_run(monkeypatch):
  """With WORLD_SIZE in the environment (a rank started by torch.distributed.run, by the driver or by the launcher
  above) bench.py must NOT launch again."""
  bench = _fresh_bench()
  monkeypatch.setenv('WORLD_SIZE', '2')
  called = {}
  monkeypatch.setattr(bench, 'launch_ranks', lambda *a: called.setdefault('launched', True))
  monkeypatch.setattr(bench, '_load', lambda: (_ for _ in ()).throw(KeyboardInterrupt('rank path')))
  with pytest.raises(KeyboardInterrupt):
    bench.main(['--gpus', '2'])
  assert 'launched' not in called


def test_act_rng_streams_differ_per_instance_and_rank():
  """Default-seeded on-device exploration streams: two controllers of one process, and the same controller on two
  data-parallel ranks seeded alike, must get different Philox keys (round-2 advisor finding); an explicit seed is
  used as given, and a stream is reproducible from (torch seed, instance order, rank)."""
  from torchrl_amd import functional as F
  torch.manual_seed(123)
  a, b = F.ActRng(), F.ActRng()
  sa, sb = a.next(), b.next()
  assert sa[0] != sb[0] and sa[1] == sb[1] == 0
  assert a.next() == (sa[0], 1)
  r0, r1 = F.ActRng(rank=0), F.ActRng(rank=1)
  r1._instance = r0._instance      # the same controller, constructed in the same order, on two ranks
  assert r0.next()[0] != r1.next()[0]
  assert F.ActRng(7).next() == (7, 0) and F.ActRng(7).next() == (7, 0)
  x, y = F.ActRng(rank=0), F.ActRng(rank=0)
  y._instance = x._instance
  assert x.next() == y.next()


def test_python_layer_reads_no_environment_switches():
  """The product's path is never changed through the environment (round-2 finding: TRL_FUSED_LEARN & co. silently switched
  kernels): the only os.environ read under torchrl_amd/ is TRL_LIB in _lib.py (which build of the library to load) and
  the compiler lookup of _build.py."""
  import re
  root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'torchrl_amd')
  hits = []
  for d, _, files in os.walk(root):
    for f in files:
      if f.endswith('.py'):
        for i, line in enumerate(open(os.path.join(d, f)), 1):
          if re.search(r'os\.environ|getenv', line) and not line.lstrip().startswith('#'):
            hits.append((os.path.relpath(os.path.join(d, f), root), i, line.strip()))
  allowed = {'_lib.py', '_build.py'}
  assert all(h[0] in allowed for h in hits), [h for h in hits if h[0] not in allowed]
