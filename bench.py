#!/usr/bin/env python3
"""Benchmark of the learner hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ppo|ddpg|dqn_atari]

Default workload = BASELINE.json configs[1]: PPO on a synthetic rollout T=2048, N_envs=4096
per GPU, MLP 64x64.  One "step" is one full pass of the on-policy hot path over that rollout,
inputs already resident in HBM: value/policy forward over the rollout, the GAE reverse scan,
log-probs, then n_epochs x (T*N / minibatch) learn() calls (fused clipped-surrogate loss,
backward, [all-reduce], fused clip+Adam).  value = learner transitions/s = transitions fed
to learn() by all ranks / wall time; GAE steps/s is reported beside it.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): every rank owns its own
N_envs env shard (weak scaling) and the only collective is the flat-gradient all-reduce.

Rank 0 prints ONE JSON line.  `roofline` is the GAE kernel (HBM-bound), timed with events on
the stream it is launched on; `kernels` carries the same figures for the replay gather
kernels on BASELINE configs 3 and 4 shaped data; `cpu_baseline` is the oracle's port of the
reference path (per-env Python GAE loop + torch-CPU learn) on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

# torch, torch.distributed and the package are imported by _load() AFTER main() has decided whether this process is a
# rank or the launcher of N ranks: the launcher (`bench.py --gpus N` outside torch.distributed.run) never imports
# anything that could touch the GPU.
np = torch = dist = F = None
PPOController = A2CController = DDPGController = DQNController = TensorTupleDataset = None


def _load():
  global np, torch, dist, F, PPOController, A2CController, DDPGController, DQNController, TensorTupleDataset
  import numpy
  import torch as _torch
  import torch.distributed as _dist
  from torchrl_amd import functional as _F
  from torchrl_amd.contrib import controllers as _c
  from torchrl_amd.utils.storage import TensorTupleDataset as _T
  np, torch, dist, F = numpy, _torch, _dist, _F
  PPOController, A2CController, DDPGController, DQNController = (_c.PPOController, _c.A2CController, _c.DDPGController,
                                                                  _c.DQNController)
  TensorTupleDataset = _T

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290
BF16_MFMA_PEAK_TFLOPS = 2516.6   # v_mfma_f32_32x32x16_bf16, 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz (dense; MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD (MI355X_MICROARCH.md, matrix cores)


def parse(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=None)
  ap.add_argument('--warmup', type=int, default=None)
  ap.add_argument('--workload', default='ppo', choices=['ppo', 'a2c', 'ddpg', 'dqn_atari'])
  ap.add_argument('--T', type=int, default=2048)
  ap.add_argument('--N', type=int, default=4096)
  ap.add_argument('--hidden', default='64,64')
  ap.add_argument('--minibatch', type=int, default=None)
  ap.add_argument('--epochs', type=int, default=4)
  ap.add_argument('--capacity', type=int, default=None)
  ap.add_argument('--kernel-iters', type=int, default=50)
  ap.add_argument('--rollout-layout', default='blocked', choices=['blocked', 'tn'],
                  help='ppo: blocked = the rollout is kept as [N/E][T][E] (RolloutBuffer(layout="blocked"): flat per-group '
                       'streams for the value pass and the GAE scan); tn = [T, N], the parity layout, minibatches in the '
                       "reference's env-major order")
  ap.add_argument('--block-envs', type=int, default=16, choices=[8, 16])
  ap.add_argument('--only', default=None,
                  help='run ONE kernel-timing entry and print its JSON (for a rocprofv3 kernel trace that holds nothing else): '
                       'gae_cold_blocked, gae_warm_blocked, gae_cold_tn, gae_warm_tn, gather_wide_1024, gather_wide_256, '
                       'gather_narrow_65536, gather_narrow_128, learn_kernel, learn_kernel_bf16x6, learn_epoch, learn_epoch_256')
  ap.add_argument('--no-sweep', action='store_true', help='ppo: skip the minibatch sweep (64 / 65,536 / 1,048,576)')
  ap.add_argument('--sampler', default='mt19937', choices=['mt19937', 'device'],
                  help='off-policy loops: mt19937 = bit-exact torch.randperm prefix on the host (parity mode); '
                       'device = Philox sampler on the GPU (throughput mode, not index-identical)')
  ap.add_argument('--no-gemm-tuning', action='store_true',
                  help='off-policy loops: keep the BLAS default kernel picks (default: PyTorch TunableOp chooses per shape)')
  ap.add_argument('--f32-frames', action='store_true',
                  help='dqn_atari: gather the uint8 frames as float32 (the path before the uint8 first layer)')
  ap.add_argument('--dqn-chain', action='store_true',
                  help='dqn_atari: layer 1, layer 2, TD loss and input gradients of the uint8 step as ONE launch '
                       '(trl_offpolicy_set_chain(2)); default: four launches')
  ap.add_argument('--in-place', action='store_true',
                  help='dqn_atari: read the sampled frames in place out of the ring (learn_indexed(in_place=True)) instead of '
                       'gathering a dense batch first; bit-identical results, measured slower')
  ap.add_argument('--products', default='bf16x6', choices=['f32', 'bf16x6'],
                  help="ppo / a2c: the whole-step kernel's 64 x 64 products: exact f32 MFMA (parity mode) or bf16 MFMA on exactly "
                       "split operands (f32-equivalent, csrc/trl_learn.hip k_ac_learn_pair_bx)")
  ap.add_argument('--force-gradsync', action='store_true',
                  help='keep the gradient all-reduce in the step at world size 1 (rehearses the N-rank call sequence on one GPU)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extras', action='store_true')
  ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                  help='HIP-graph replay of learn(); auto = off wherever the whole step is a handful of fused launches '
                       '(ppo, a2c, ddpg, dqn_atari on uint8 frames), on for dqn_atari --f32-frames (BLAS + autograd path)')
  ap.add_argument('--no-graph', action='store_true', help='same as --graph off')
  return ap.parse_args(argv)


def launch_ranks(args, argv):
  """`bench.py --gpus N` started as ONE process (no WORLD_SIZE in the environment, the way the driver invokes it): start
  N fresh rank processes with torch.distributed.run, one per GPU over RCCL, relay rank 0's JSON line and the worst exit
  code.  Called before torch is imported: this process never touches the GPU (a process that has must not exec, and
  the ranks are children, not replacements)."""
  assert 'torch' not in sys.modules or torch is None, 'the launcher must run before the GPU stack is loaded'
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:     # a free rendezvous port on the loopback
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  env['TRL_BENCH_LAUNCHED'] = '1'
  print('bench.py: starting %d ranks: %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
  proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
  line = None
  for out in proc.stdout:          # rank 0 prints the one JSON line; anything else on stdout is passed through to stderr
    t = out.strip()
    if t.startswith('{') and t.endswith('}') and '"metric"' in t:
      line = t
    elif t:
      print(t, file=sys.stderr, flush=True)
  rc = proc.wait()
  if rc != 0:
    print('bench.py: a rank failed (torch.distributed.run exit code %d)' % rc, file=sys.stderr)
    return rc
  if line is None:
    print('bench.py: the ranks ended without a result line', file=sys.stderr)
    return 1
  print(line, flush=True)
  return 0


def setup_dist(args):
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  backend = os.environ.get('TRL_BENCH_BACKEND', 'nccl')   # 'gloo' lets a 1-GPU box rehearse N ranks
  ndev = torch.cuda.device_count()
  if world > 1 and backend == 'nccl' and ndev < world:
    raise SystemExit('bench.py: %d ranks over RCCL need %d GPUs, this node shows %d (TRL_BENCH_BACKEND=gloo rehearses '
                     'N ranks on fewer cards)' % (world, world, ndev))
  torch.cuda.set_device(local % max(1, ndev))
  local = local % max(1, ndev)
  ranks_seen = 1
  if world > 1 or args.force_gradsync:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
    # the world the collective library actually formed: a sum of ones over it
    one = torch.ones(1, device='cuda' if backend == 'nccl' else 'cpu')
    dist.all_reduce(one)
    ranks_seen = int(one.item())
  if args.gpus != world and rank == 0:
    print('note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
  args.ranks_seen = ranks_seen
  args.backend = backend if (world > 1 or args.force_gradsync) else None
  return rank, local, world


def trace(msg):
  if os.environ.get('TRL_BENCH_TRACE'):
    print('[rank %s] %s' % (os.environ.get('RANK', '0'), msg), file=sys.stderr, flush=True)


def barrier(world):
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()


def timed_steps(step_fn, steps, warmup, world):
  for _ in range(warmup):
    step_fn()
  barrier(world)
  t0 = time.perf_counter()
  for _ in range(steps):
    step_fn()
  barrier(world)
  dt = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([dt], device='cuda', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
  return dt


def kernel_time_us(launch, iters):
  """Average duration of one launch.  `iters` launches are captured into one HIP graph on the
  stream the C ABI launches on (torch's current stream during capture) and the replay is
  bracketed by events on that same stream, so host launch cost (ctypes, Python) is not in
  the figure; kernels of a few microseconds would otherwise be timed at the host's rate."""
  for _ in range(3):
    launch()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  # thread_local: other threads of the process (the collective watchdog of a multi-rank run) may
  # keep calling HIP while this one captures
  with torch.cuda.graph(g, capture_error_mode='thread_local'):
    for _ in range(iters):
      launch()
  g.replay()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  g.replay()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) * 1e3 / iters


PMC_FILES = ('r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01f_pmc_traffic.json')


def pmc_traffic(key):
  """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r02_pmc_traffic.json, made by
  tools/pmc_traffic.py from two --pmc passes over `bench.py --only <entry>`; counters cannot be read from inside
  this process), or None when that launch was not profiled."""
  for name in PMC_FILES:
    try:
      with open(os.path.join(ROOT, 'profiles', name)) as f:
        return int(json.load(f)[key]['hbm_bytes'])
    except (OSError, KeyError, ValueError):
      continue
  return None


def roofline(name, algo_bytes, us, note, traffic_key=None):
  gbs = algo_bytes / (us * 1e-6) / 1e9
  return {'kernel': name, 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
          'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(traffic_key) if traffic_key else None,
          'us_per_launch': round(us, 2),
          'algorithmic_bytes_per_launch': int(algo_bytes), 'note': note}


# ------------------------------------------------------------------------------ PPO (configs[1])
def make_rollout(T, N, obs_size, act_size, dev, seed):
  g = torch.Generator(device=dev).manual_seed(seed)
  obs = torch.randn(T, N, obs_size, device=dev, generator=g)
  action = torch.randn(T, N, act_size, device=dev, generator=g)
  reward = torch.randn(T, N, device=dev, generator=g)
  done = (torch.rand(T, N, device=dev, generator=g) < 0.01).float()
  last_next_obs = torch.randn(N, obs_size, device=dev, generator=g)
  return obs, action, reward, done, last_next_obs


def _tune_lib():
  """The -DTRL_TUNING diagnostic build, for its flat-copy probe only (never for a timed product kernel)."""
  import ctypes as C
  path = os.path.join(ROOT, 'torchrl_amd', 'lib', 'libtrl_hip_tune.so')
  if not os.path.exists(path):
    return None
  lib = C.CDLL(path)
  lib.trl_tune_flat_copy.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_void_p]
  lib.trl_tune_flat_copy.restype = C.c_int
  return lib


def gae_roofline(args, dev, T, N, layout, warm=False, with_ceilings=True):
  """The GAE kernel alone (graph replay, events on the launch stream, caller-owned outputs).
  Cold: the launches rotate over enough independent buffer sets (> 3 x 256 MiB) that every one reads its
  inputs from HBM, not from the Infinity Cache; that is the roofline figure.  Warm: one set re-used."""
  n = T * N
  E = args.block_envs
  algo = 16 * n + 4 * N          # r, v read + ret, adv written (4 B each) + bootstrap row
  nsets = 1 if warm else max(2, int(3 * (256 << 20) / algo) + 1)
  shape = (N // E, T, E) if layout == 'blocked' else (T, N)
  sets = [(torch.randn(shape, device=dev), torch.randn(shape, device=dev), torch.randn(N, device=dev),
           (torch.empty(shape, device=dev), torch.empty(shape, device=dev))) for _ in range(nsets)]
  rot = {'i': 0}
  fn = F.gae_blocked if layout == 'blocked' else F.gae

  def launch():
    s = sets[rot['i'] % nsets]
    rot['i'] += 1
    fn(s[0], s[1], s[2], 0.99, 0.95, out=s[3])

  iters = max(args.kernel_iters, 2 * nsets) // nsets * nsets
  us = kernel_time_us(launch, iters)
  lay = ('blocked [N/%d][T][%d] (flat per-group streams)' % (E, E)) if layout == 'blocked' else '[T, N] (parity layout: 64 B row segments per workgroup)'
  name = 'k_gae_pipe<%d,%d,false,1>' % ((E, 3) if layout == 'blocked' else (16, 2))
  key = 'gae_%s_%s' % ('warm' if warm else 'cold', layout)
  roof = roofline(name, algo, us, '16 B per (t, env) step x T*N = %d steps per launch, rollout layout %s; inputs %s'
                  % (n, lay, 'resident in the Infinity Cache (one buffer set re-used)' if warm else
                     'cold in HBM (%d rotating buffer sets)' % nsets),
                  traffic_key=key if (T, N) == (2048, 4096) else None)
  roof['entry'] = key
  roof['profile'] = 'profiles/r03_%s_kernel_stats.csv' % key
  if with_ceilings:
    def copy_torch():      # a plain device copy moving the same bytes over the same rotating sets (two launches)
      s = sets[rot['i'] % nsets]
      rot['i'] += 1
      s[3][0].copy_(s[0])
      s[3][1].copy_(s[1])
    us_copy = kernel_time_us(copy_torch, iters)
    roof['copy_ceiling'] = {'GB/s': round(algo / us_copy / 1e3, 1), 'us': round(us_copy, 2),
                            'note': 'torch device copy of the same bytes over the same buffer sets, same process (two launches)'}
    roof['frac_of_copy_ceiling'] = round(us_copy / us, 4)
    tune = _tune_lib()
    if tune is not None:
      def copy_flat():     # tools/copy_probe.hip's best flat copy: one launch, 256 persistent workgroups, 4 x 16 B per lane, nt
        s = sets[rot['i'] % nsets]
        rot['i'] += 1
        rc = tune.trl_tune_flat_copy(s[0].data_ptr(), s[1].data_ptr(), s[3][0].data_ptr(), s[3][1].data_ptr(), n, 256,
                                     torch.cuda.current_stream().cuda_stream)
        assert rc == 0
      us_flat = kernel_time_us(copy_flat, iters)
      roof['flat_copy_probe'] = {'GB/s': round(algo / us_flat / 1e3, 1), 'us': round(us_flat, 2),
                                 'note': 'one-launch fully coalesced non-temporal copy of both planes (diagnostic build, '
                                         'tools/copy_probe.hip best setting): what a read-plus-write pass of these bytes reaches'}
      roof['frac_of_flat_copy'] = round(us_flat / us, 4)
  del sets
  torch.cuda.empty_cache()
  return roof


def make_ppo_controller(args, dev, products=None):
  hidden = tuple(int(h) for h in args.hidden.split(','))
  torch.manual_seed(0)
  return PPOController(3, 1, gamma=0.99, lmbda=0.95, lr=3e-4, alpha=0.5, beta=1e-3, clip_ratio=0.2, max_grad_norm=1.0,
                       device=dev, hidden=hidden, graph=args.use_graph, products=products or args.products), hidden


DTYPE_NOTE = {'f32': 'f32',
              'bf16x6': 'f32-equivalent (bf16x6 split, f32 accumulate)'}


def bench_ppo(args, rank, world, dev):
  T, N, obs_size, act_size = args.T, args.N, 3, 1
  # 8 minibatches per epoch (T*N / 8 = 1,048,576 at the default shape): what PPO on thousands of
  # parallel envs is usually run with (4-8 minibatches per epoch), and 32 optimizer steps -- hence 32
  # gradient all-reduces on N > 1 ranks -- per rollout.  --minibatch overrides; the reference's own sizes
  # (64) and SURVEY's 65,536 are in `minibatch_sweep`
  B = args.minibatch or max(1, (T * N) // 8)
  steps = args.steps if args.steps is not None else 5
  warmup = args.warmup if args.warmup is not None else 2
  ctl, hidden = make_ppo_controller(args, dev)
  trace('controller built')
  obs, action, reward, done, last_next_obs = make_rollout(T, N, obs_size, act_size, dev, 1234 + rank)
  last_done = done[-1]
  n = T * N
  blocked = args.rollout_layout == 'blocked'
  if blocked:     # the rollout as RolloutBuffer(layout='blocked') holds it
    E = args.block_envs
    obs, action, reward = F.to_blocked(obs, E), F.to_blocked(action, E), F.to_blocked(reward, E)
  phase, keep = {}, {}

  def prepare():
    """compute_return over the rollout + the flat minibatch columns (blocked: consecutive slices of the blocked
    order; tn: the reference's env-major order, examples/ppo.py:61-68)"""
    returns, adv, logp, _values = ctl.compute_return_batched(obs, action, reward, last_next_obs, last_done,
                                                             env_major_out=not blocked)
    if blocked:
      cols = (obs.reshape(n, obs_size), action.reshape(n, act_size), returns.reshape(n, 1), logp.reshape(n, act_size),
              adv.reshape(n, 1))
    else:
      cols = (obs.transpose(0, 1).reshape(n, obs_size), action.transpose(0, 1).reshape(n, act_size),
              returns.reshape(n, 1), logp.transpose(0, 1).reshape(n, act_size), adv.reshape(n, 1))
    keep['cols'] = cols
    return cols

  def learn_slice(cols, s, e):
    o, a, r, l, ad = cols
    ctl.learn(o[s:e], a[s:e], None, None, None, r[s:e], l[s:e], ad[s:e])

  def step():
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e2 = torch.cuda.Event(enable_timing=True)
    e0.record()
    cols = prepare()
    e1.record()
    trace('compute_return done')
    for _ in range(args.epochs):
      for s in range(0, n, B):
        learn_slice(cols, s, min(n, s + B))
    e2.record()
    trace('learn epochs issued')
    phase['ev'] = (e0, e1, e2)

  dt = timed_steps(step, steps, warmup, world)
  trace('timed steps done')
  e0, e1, e2 = phase['ev']
  ms_return, ms_learn = e0.elapsed_time(e1), e1.elapsed_time(e2)
  learned = args.epochs * n * steps * world
  value = learned / dt
  # which arithmetic the 64 x 64 products of these learn() calls ran in (the bf16x6 variant exists for the role-split
  # kernel's shapes only: everything else is the exact-f32 kernels whatever --products says)
  bx = ctl._fused is not None and ctl._fused.bx6_takes(min(B, n))
  products_used = 'bf16x6' if bx else 'f32'
  products_note = ('bf16x6: the three 64 x 64 products per network (layer 1 forward, its weight and input gradients) on '
                   'v_mfma_f32_32x32x16_bf16, both operands split exactly into three bf16 parts, six of nine partial products, '
                   'f32 accumulators; everything else (layer 0, last layer, loss, GAE, reduction, clip, Adam) exact f32.  Not the '
                   'parity mode: `variants.exact_f32` is the same step on the exact-f32 MFMA (products="f32", the library default)'
                   if bx else 'f32: exact-f32 MFMA (the parity mode)')

  roof = gae_roofline(args, dev, T, N, args.rollout_layout)
  warm = gae_roofline(args, dev, T, N, args.rollout_layout, warm=True, with_ceilings=False)
  roof['note'] += '; cache-resident inputs: %.2f us = %.0f GB/s' % (warm['us_per_launch'], warm['achieved'])
  other = 'tn' if blocked else 'blocked'
  if N % args.block_envs == 0:
    roof['other_layout'] = gae_roofline(args, dev, T, N, other, with_ceilings=False)
  share_learn = ms_learn / max(ms_return + ms_learn, 1e-9)
  lk = 'k_ac_learn_pair_bx + k_learn_reduce: see learn_kernel_bf16x6, MFMA + VALU issue-bound' if bx else \
       'k_ac_learn + k_learn_reduce: see learn_kernel, MFMA-bound'
  extra = {'step_time_share': {'learn epochs (%s)' % lk: round(share_learn, 3),
                               'compute_return (k_ac_infer x2 + k_gae_pipe: see roofline, HBM-bound)': round(1 - share_learn, 3)},
           'gae_steps_per_sec_kernel': round(n / (roof['us_per_launch'] * 1e-6), 1),
           'gae_steps_per_sec_compute_return': round(n / (ms_return * 1e-3), 1),
           'unique_transitions_per_sec': round(value / args.epochs, 1),
           'ms_compute_return': round(ms_return, 3), 'ms_learn_epochs': round(ms_learn, 3)}
  res = {
      'metric': 'learner transitions/sec (PPO: GAE + clipped loss + update)', 'value': round(value, 1),
      'unit': 'transitions/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
      'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
      'vs_baseline': None, 'dtype': DTYPE_NOTE[products_used], 'data': 'synthetic',
      'config': {'workload': 'ppo_gae_clip: T=%d, N_envs=%d per GPU, MLP %s, obs 3, act 1' % (T, N, args.hidden),
                 'products': products_note,
                 'minibatch': B, 'n_epochs': args.epochs, 'learn_calls_per_step': args.epochs * ((n + B - 1) // B),
                 'value_counts': 'every transition once per epoch (value = n_epochs x T x N / step time; '
                                 'unique_transitions_per_sec = value / n_epochs)',
                 'rollout_layout': ('blocked [N/%d][T][%d]' % (args.block_envs, args.block_envs)) if blocked else
                 '[T, N], env-major minibatches',
                 'gamma': 0.99, 'lmbda': 0.95, 'clip_ratio': 0.2, 'parallelism': 'dp%d' % world,
                 'learn_step': ('hip-graph replay' if args.use_graph else 'eager') +
                 (': fused fwd+loss+bwd launch, reduce(+clip+Adam) launch' if ctl._fused is not None else '')},
      'roofline': roof,
  }
  res.update(extra)
  cols = keep['cols']
  if bx:
    # the same step with the exact-f32 products (the parity mode), timed the same way on every rank: both variants in the line
    main_ctl = ctl
    ctl, _ = make_ppo_controller(args, dev, products='f32')
    dt_x = timed_steps(step, steps, warmup, world)
    e0, e1, e2 = phase['ev']
    res['variants'] = {
        'bf16x6': {'value': round(value, 1), 'ms_per_step': round(dt / steps * 1e3, 3), 'ms_learn_epochs': round(ms_learn, 3),
                   'dtype': DTYPE_NOTE['bf16x6']},
        'exact_f32': {'value': round(learned / dt_x, 1), 'ms_per_step': round(dt_x / steps * 1e3, 3),
                      'ms_learn_epochs': round(e1.elapsed_time(e2), 3), 'dtype': 'f32',
                      'note': 'products="f32": k_ac_learn_pair on v_mfma_f32_32x32x2_f32, the parity mode (bit-identical to the '
                              'layer path / multi-rank sequence, tests/test_gpu_learn.py)'}}
    exact_ctl, ctl = ctl, main_ctl
    cols = prepare()
  if rank == 0 and world == 1 and not args.no_sweep:
    res['minibatch_sweep'] = minibatch_sweep(args, ctl, cols, n, ms_return, learn_slice, B, dt / steps * 1e3)
    if args.hidden != '256':
      # the same loop on the REFERENCE's own network (one hidden layer of 256, ppo_controller.py:14) at its own minibatch
      torch.manual_seed(0)
      ref_ctl = PPOController(3, 1, gamma=0.99, lmbda=0.95, lr=3e-4, alpha=0.5, beta=1e-3, clip_ratio=0.2, max_grad_norm=1.0,
                              device=dev, hidden=(256,))
      if ref_ctl._epoch is not None:
        res['learn_epoch_reference_net'] = [
            dict(minibatch=Bm, **learn_epoch_row(args, ref_ctl, cols, n, Bm, ms_return, (n + Bm - 1) // Bm,
                                                 note='ActorCriticNet(3, 1, 256), the network PPOController builds by default; '
                                                      'compute_return time taken from the headline run'))
            for Bm in (64, 32)]
  if ctl._fused is not None:
    if bx:
      res['learn_kernel'] = learn_kernel_roofline(args, exact_ctl, cols, min(B, n))
      res['learn_kernel_bf16x6'] = learn_kernel_roofline(args, ctl, cols, min(B, n))
    else:
      res['learn_kernel'] = learn_kernel_roofline(args, ctl, cols, min(B, n))
  return res, dict(T=T, hidden=hidden, B=B, epochs=args.epochs)


def minibatch_sweep(args, ctl, cols, n, ms_return, learn_slice, B_default, ms_default):
  """The same step at the reference's own minibatch (64: examples/ppo.py:92; its default is 32), at SURVEY's 65,536
  and at the bench default.  Small minibatches are timed on a bounded run of consecutive learn() calls and the
  step is extrapolated (a whole 4-epoch step at B = 64 is 524,288 calls)."""
  out = []
  for Bm in (64, 65536, 1048576):
    if Bm > n:
      continue
    calls_per_epoch = (n + Bm - 1) // Bm
    if Bm == B_default:
      ms_step, per_call, timed = ms_default, None, 'whole step (the headline run)'
    else:
      ncalls = min(calls_per_epoch, 4000 if Bm < 4096 else 256)
      for s in range(0, min(n, 50 * Bm), Bm):        # warm-up
        learn_slice(cols, s, min(n, s + Bm))
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(ncalls):
        s = i * Bm
        learn_slice(cols, s, min(n, s + Bm))
      torch.cuda.synchronize()
      per_call = (time.perf_counter() - t0) / ncalls * 1e3
      ms_step = ms_return + args.epochs * calls_per_epoch * per_call
      timed = '%d consecutive learn() calls, step extrapolated' % ncalls
    tps = args.epochs * n / (ms_step * 1e-3)
    row = {'minibatch': Bm, 'learn_calls_per_step': args.epochs * calls_per_epoch, 'ms_per_step': round(ms_step, 3),
           'transitions_per_sec': round(tps, 1), 'unique_transitions_per_sec': round(tps / args.epochs, 1), 'timed': timed}
    if Bm < 4096 and getattr(ctl, '_epoch', None) is not None:
      row['learn_epoch'] = learn_epoch_row(args, ctl, cols, n, Bm, ms_return, calls_per_epoch)
    if per_call is not None:
      row['us_per_learn_call'] = round(per_call * 1e3, 2)
      if Bm < 4096 and ctl._fused is not None:
        # the device side of one learn() alone (the launch pair replayed from a HIP graph, no host in between)
        o, a, r, l, ad = (c[:Bm] for c in cols)
        row['us_gpu_per_learn_call'] = round(kernel_time_us(lambda: ctl._fused(o, a, r, l, ad, 0.2, 0.5, 1e-3,
                                                                                adam=ctl.ac_net_optim), 200), 2)
        row['bound'] = ('latency, not throughput: a learn() is two dependent launches (k_ac_learn on %d strips of 32 samples = '
                        '%d of the chip\'s 1024 wavefronts, then k_learn_reduce + Adam); us_gpu_per_learn_call is the pair '
                        'replayed back to back from a HIP graph, us_per_learn_call adds the host (Python + ctypes).  '
                        'DESIGN.md section 5 has the breakdown and what a multi-minibatch launch would buy'
                        % ((Bm + 31) // 32, (Bm + 31) // 32))
    out.append(row)
  return out


def learn_epoch_row(args, ctl, cols, n, Bm, ms_return, calls_per_epoch, note=None):
  """The reference's minibatch loop (examples/ppo.py:66-73) as ONE persistent launch, PPOController.learn_epoch
  (csrc/trl_epoch.hip): timed on a bounded run of consecutive minibatches (one launch), the 4-epoch step extrapolated
  exactly as the learn()-per-minibatch row beside it."""
  K = min(calls_per_epoch, 8192)
  o, a, r, l, ad = (c[:K * Bm] for c in cols)
  ds = [o, a, None, None, None, r, l, ad]
  ctl.learn_epoch(ds, Bm, 1)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record()
  ctl.learn_epoch(ds, Bm, 1)
  e1.record()
  torch.cuda.synchronize()
  wall = (time.perf_counter() - t0) / K * 1e3          # ms per minibatch, host included
  dev_us = e0.elapsed_time(e1) * 1e3 / K
  ms_step = ms_return + args.epochs * calls_per_epoch * wall
  tps = args.epochs * n / (ms_step * 1e-3)
  row = {'ms_per_step': round(ms_step, 3), 'transitions_per_sec': round(tps, 1),
         'unique_transitions_per_sec': round(tps / args.epochs, 1), 'us_per_minibatch': round(wall * 1e3, 3),
         'us_gpu_per_minibatch': round(dev_us, 3),
         'timed': 'one launch over %d consecutive minibatches of %d (k_ppo_epoch: two persistent workgroups, critic and actor), '
                  'step extrapolated' % (K, Bm),
         'bound': 'one CU per network: exact-f32 MFMA products of a 64-sample tile + clip/Adam of every parameter, each update '
                  'waiting for the one before (DESIGN.md section 5)'}
  if note:
    row['note'] = note
  return row


def learn_kernel_roofline(args, ctl, cols, Bk):
  # the whole-step kernel pair alone (gradients only), graph-timed like the GAE kernel; FLOPs are the
  # networks' own (no padding): forward, weight gradients, and input gradients of the hidden layers
  f = ctl._fused
  o, a, r, l, ad = (c[:Bk].contiguous() for c in cols)
  us_l = kernel_time_us(lambda: f(o, a, r, l, ad, 0.2, 0.5, 1e-3), max(10, args.kernel_iters // 2))
  # the spread over single launch pairs (events around each): the graph figure above is their mean without host gaps
  singles = []
  for _ in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f(o, a, r, l, ad, 0.2, 0.5, 1e-3)
    e1.record()
    torch.cuda.synchronize()
    singles.append(e0.elapsed_time(e1) * 1e3)
  flops = 0
  for lin in (f.critic_lin, f.actor_lin):
    for i, m in enumerate(lin):
      flops += (4 if i == 0 else 6) * m.in_features * m.out_features
  flops *= Bk
  tf = flops / (us_l * 1e-6) / 1e12
  if f.bx6_takes(Bk):
    # the square products' FLOPs are issued six times over on the bf16 pipe; the rest stays on the f32 pipe / VALU
    sq = sum(6 * m.in_features * m.out_features for lin in (f.critic_lin, f.actor_lin) for m in lin[1:-1]) * Bk
    issued = 6 * sq / (us_l * 1e-6) / 1e12
    return {'kernel': 'k_ac_learn_pair_bx+k_learn_reduce', 'bound': 'mfma', 'achieved': round(tf, 1), 'peak': BF16_MFMA_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(issued / BF16_MFMA_PEAK_TFLOPS, 4), 'issued_bf16': round(issued, 1),
            'f32_equivalent_vs_f32_mfma_peak': round(tf / F32_MFMA_PEAK_TFLOPS, 4), 'us_per_launch': round(us_l, 2),
            'us_single_launch_pairs': {'mean': round(sum(singles) / len(singles), 2), 'min': round(min(singles), 2),
                                       'max': round(max(singles), 2)},
            'flops_per_launch': int(flops), 'entry': 'learn_kernel_bf16x6',
            'profile': 'profiles/r03_learn_kernel_bf16x6_kernel_stats.csv',
            'note': 'the same launch pair with products="bf16x6" over one minibatch of %d: `achieved` = the networks\' own %d FLOP per '
                    'sample (f32-equivalent work) per second; `issued_bf16` = 6 x the three 64 x 64 products on '
                    'v_mfma_f32_32x32x16_bf16, `frac` = issued_bf16 / the dense bf16 peak.  The kernel is bound by the SIMD\'s '
                    'issue of MFMA + VALU together (the operand splits are 4.5 VALU instructions per element and do not overlap '
                    'the matrix pipe on this part, DESIGN.md section 4.5)' % (Bk, flops // Bk)}
  out = {'kernel': 'k_ac_learn+k_learn_reduce', 'bound': 'mfma', 'achieved': round(tf, 1),
         'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / F32_MFMA_PEAK_TFLOPS, 4),
         'us_per_launch': round(us_l, 2), 'us_single_launch_pairs': {'mean': round(sum(singles) / len(singles), 2),
                                                                     'min': round(min(singles), 2), 'max': round(max(singles), 2),
                                                                     'note': 'events around each of 12 eager launch pairs (host gap between the two launches included); '
                                                                             'us_per_launch / frac are the mean of graph-replayed pairs'},
         'flops_per_launch': int(flops), 'entry': 'learn_kernel',
         'profile': 'profiles/r03_learn_kernel_kernel_stats.csv',
         'note': 'forward + PPO loss + backward of critic and actor over one minibatch of %d '
                 '(exact f32 MFMA); %d FLOP per sample' % (Bk, flops // Bk)}
  if len(f.critic_lin) == 2:
    # one hidden layer: no hidden x hidden product; the last layer's dot products, their transposes and
    # (fan-in <= 4) the layer-0 weight gradient run on the VALU, so the MFMA peak is not this shape's bound
    out['note'] += ('; single hidden layer: most of these FLOPs are VALU FMAs (f32 VALU peak '
                    '%.1f TFLOP/s), frac is still quoted against the MFMA peak' % (F32_MFMA_PEAK_TFLOPS / 2))
  return out


def ppo_info(args):
  hidden = tuple(int(h) for h in args.hidden.split(','))
  return dict(T=args.T, hidden=hidden, B=args.minibatch or max(1, (args.T * args.N) // 8), epochs=args.epochs)


def cpu_baseline_ppo(info):
  """Oracle port of the reference path on the host: per-env compute_return loop
  (ppo_controller.py:32-51) + learn() minibatches (ppo_controller.py:53-80).  Two rows: the bench's own minibatch
  (`value`) and the reference's minibatch of 64 (examples/ppo.py:92), the latter like-for-like with the GPU line's
  minibatch_sweep[64] / learn_epoch rows (learn() timed on a bounded run of calls, the step extrapolated the same way)."""
  from oracle import loss_oracle
  from torchrl_amd.contrib.models import ActorCriticNet
  torch.set_num_threads(1)
  T, hidden, B, epochs = info['T'], info['hidden'], info['B'], info['epochs']
  n_envs = 384     # ~10 s of host work
  torch.manual_seed(0)
  net = ActorCriticNet(3, 1, hidden)
  opt = torch.optim.Adam(net.parameters(), lr=3e-4)
  cols = [[] for _ in range(8)]
  t0 = time.perf_counter()
  for _ in range(n_envs):
    obs, act = torch.randn(T, 3), torch.randn(T, 1)
    rew, nobs, done = torch.randn(T, 1), torch.randn(T, 3), torch.zeros(T, 1)
    ret, lp, v = loss_oracle.ppo_compute_return(net, obs, act, rew, nobs, done, 0.99, 0.95)
    for c, x in zip(cols, (obs, act, rew, nobs, done, ret, lp, ret - v)):
      c.append(x)
  cols = [torch.cat(c, 0) for c in cols]
  t_return = time.perf_counter() - t0
  n = n_envs * T
  Bc = min(B, n)
  for _ in range(epochs):
    for s in range(0, n, Bc):
      loss_oracle.ppo_learn(net, opt, *[c[s:s + Bc] for c in cols], clip=0.2, alpha=0.5, beta=1e-3,
                            max_grad_norm=1.0)
  dt = time.perf_counter() - t0
  # the reference's own minibatch: 2,000 consecutive learn() calls of 64 rows, the 4-epoch step extrapolated
  ncalls, B64 = 2000, 64
  t1 = time.perf_counter()
  for i in range(ncalls):
    s = (i * B64) % (n - B64)
    loss_oracle.ppo_learn(net, opt, *[c[s:s + B64] for c in cols], clip=0.2, alpha=0.5, beta=1e-3, max_grad_norm=1.0)
  per_call = (time.perf_counter() - t1) / ncalls
  dt64 = t_return + epochs * ((n + B64 - 1) // B64) * per_call
  return {'value': round(epochs * n / dt, 1), 'unit': 'transitions/s', 'cores': 1, 'kind': 'port',
          'sample': '%d of the envs x T=%d: per-env torch-CPU GAE loop + %d epochs x minibatch %d learn(), '
                    '1 thread (the reference loops are single-threaded Python); %.1f s' % (n_envs, T, epochs, Bc, dt),
          'minibatch_64': {'value': round(epochs * n / dt64, 1), 'unit': 'transitions/s', 'us_per_learn_call': round(per_call * 1e6, 1),
                           'sample': 'the same %d envs; learn() at the reference\'s minibatch of 64 timed on %d consecutive calls, '
                                     'the %d-epoch step extrapolated (as the GPU line\'s minibatch_sweep does)' % (n_envs, ncalls, epochs)},
          'host_cpus': os.cpu_count()}



# ------------------------------------------------------------------------------ A2C (the other half of configs[4])
def bench_a2c(args, rank, world, dev):
  """A2C on the same synthetic rollout shape (CartPole-shaped: obs 4, 2 discrete actions): one step =
  value forward + return scan over [T, N], then ONE learn() over the whole rollout
  (examples/a2c.py:38-53: a single batch per rollout)."""
  T, N, obs_size, n_act = args.T, args.N, 4, 2
  hidden = tuple(int(h) for h in args.hidden.split(','))
  steps = args.steps if args.steps is not None else 10
  warmup = args.warmup if args.warmup is not None else 2
  torch.manual_seed(0)
  ctl = A2CController(obs_size, n_act, gamma=0.99, lmbda=1.0, lr=1e-3, alpha=0.5, beta=1e-3, device=dev,
                      hidden=hidden, graph=args.use_graph)
  g = torch.Generator(device=dev).manual_seed(4321 + rank)
  obs = torch.randn(T, N, obs_size, device=dev, generator=g)
  action = torch.randint(0, n_act, (T, N), device=dev, generator=g).float()
  reward = torch.randn(T, N, device=dev, generator=g)
  last_next_obs = torch.randn(N, obs_size, device=dev, generator=g)
  last_done = (torch.rand(N, device=dev, generator=g) < 0.01).float()
  n = T * N
  obs_em = obs.transpose(0, 1).reshape(n, obs_size).contiguous()
  act_em = action.transpose(0, 1).reshape(n, 1).contiguous()

  def step():
    returns, _adv, _v = ctl.compute_return_batched(obs, reward, last_next_obs, last_done, env_major_out=True)
    ctl.learn(obs_em, act_em, None, None, None, returns.reshape(n, 1))

  dt = timed_steps(step, steps, warmup, world)
  res = {
      'metric': 'learner transitions/sec (A2C: return scan + loss + update)', 'value': round(n * steps * world / dt, 1),
      'unit': 'transitions/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
      'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': 'a2c: T=%d, N_envs=%d per GPU, MLP %s, obs %d, %d actions, one learn() per rollout'
                             % (T, N, args.hidden, obs_size, n_act),
                 'parallelism': 'dp%d' % world,
                 'learn_step': 'fused fwd+loss+bwd launch' if ctl._fused is not None else 'general path'},
  }
  return res, {}

# ------------------------------------------------------------------------------ nets outside the register-chained kernels
def wide_nets(dev, B=1 << 20):
  """One learn() over a 1 M-row batch on the reference's default width (256) with more observations / actions than the
  whole-step kernels of csrc/trl_learn.hip take: the tile pipeline spread over the chip (csrc/trl_epoch.hip:
  trl_ppo_learn_tiles / trl_a2c_learn_tiles) against the layer-by-layer path (fused=False) of the same controller;
  and compute_return_batched over the same rows (trl_ac_forward_tiles against the module path)."""
  rows = []
  for kind, obs_size, A in (('ppo', 8, 4), ('ppo', 17, 6), ('a2c', 8, 4), ('a2c', 17, 6)):
    g = torch.Generator(device=dev).manual_seed(obs_size + A)
    o = torch.randn(B, obs_size, device=dev, generator=g)
    r = torch.randn(B, 1, device=dev, generator=g)
    row = {'net': '%s(%d, %d, 256)' % ('ActorCriticNet' if kind == 'ppo' else 'A2CNet', obs_size, A), 'rows': B}
    for fused in (True, False):
      if kind == 'ppo':
        c = PPOController(obs_size, A, device=dev, hidden=(256,), fused=fused)
        a, l, ad = (torch.randn(B, A, device=dev, generator=g), torch.randn(B, A, device=dev, generator=g) - 1,
                    torch.randn(B, 1, device=dev, generator=g))
        fn = lambda: c.learn(o, a, None, None, None, r, l, ad)  # noqa: E731
      else:
        c = A2CController(obs_size, A, device=dev, hidden=(256,), fused=fused)
        a = torch.randint(0, A, (B, 1), device=dev, generator=g).float()
        fn = lambda: c.learn(o, a, None, None, None, r)  # noqa: E731
      for _ in range(2):
        fn()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(5):
        fn()
      torch.cuda.synchronize()
      tiles = fused and getattr(c, '_tiles', None) is not None and c._fused is None
      key = 'ms_learn_tiles' if tiles else ('ms_learn' if fused else 'ms_learn_layer_path')
      row[key] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
      # compute_return over the same rows as a [T, N] rollout: the forward passes through trl_ac_forward_tiles / the modules
      T, N = 2048, B // 2048
      ob3, rew = o.view(T, N, obs_size), r.view(T, N)
      last_obs, last_done = o[:N], torch.zeros(N, device=dev)
      if kind == 'ppo':
        act3 = a.view(T, N, A)
        fr = lambda: c.compute_return_batched(ob3, act3, rew, last_obs, last_done)  # noqa: E731
      else:
        fr = lambda: c.compute_return_batched(ob3, rew, last_obs, last_done)  # noqa: E731
      for _ in range(2):
        fr()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(5):
        fr()
      torch.cuda.synchronize()
      row['ms_compute_return_tiles' if tiles else ('ms_compute_return' if fused else 'ms_compute_return_module_path')] = \
          round((time.perf_counter() - t0) / 5 * 1e3, 3)
      del c
    rows.append(row)
  return rows


# ------------------------------------------------------------------------------ replay gathers
def fill_ring(ds, n_rows, make_cols, chunk):
  for lo in range(0, n_rows, chunk):
    ds.extend(*make_cols(min(chunk, n_rows - lo)))


def gather_roofline(dev, iters, rows, B, atari):
  """Gather kernel(s) alone on BASELINE config 3 (Pendulum, 36 B rows) or 4 (Atari u8 frames)."""
  g = torch.Generator(device=dev).manual_seed(7)
  if atari:
    ds = TensorTupleDataset(size=rows, device=dev,
                            storage_dtypes=[torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32],
                            gather_dtypes=[torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32])
    def cols(k):
      return [torch.randint(0, 256, (k, 84, 84, 4), device=dev, dtype=torch.uint8, generator=g),
              torch.randint(0, 6, (k, 1), device=dev, generator=g).float(),
              torch.randint(-1, 2, (k, 1), device=dev, generator=g).float(),
              torch.randint(0, 256, (k, 84, 84, 4), device=dev, dtype=torch.uint8, generator=g),
              (torch.rand(k, 1, device=dev, generator=g) < 0.005).float()]
    fill_ring(ds, rows, cols, 4096)
    row_bytes = 2 * 28224 + 12
  else:
    ds = TensorTupleDataset(size=rows, device=dev)
    def cols(k):
      return [torch.rand(k, 3, device=dev, generator=g) * 2 - 1, torch.rand(k, 1, device=dev, generator=g) * 2 - 1,
              -16 * torch.rand(k, 1, device=dev, generator=g), torch.rand(k, 3, device=dev, generator=g) * 2 - 1,
              (torch.rand(k, 1, device=dev, generator=g) < 0.01).float()]
    fill_ring(ds, rows, cols, 1 << 18)
    row_bytes = 36
  idx = torch.randint(0, rows, (B,), device=dev, generator=g)
  out = ds._alloc_out(B)
  us = kernel_time_us(lambda: ds._gather_index(idx, out), iters)
  algo = B * (2 * row_bytes + 8)
  name = 'k_gather_wide (one launch: wide chunks + narrow tail)' if atari else 'k_gather_narrow'
  key = 'gather_%s_%d' % ('wide' if atari else 'narrow', B)
  r = roofline(name, algo, us, '(2 x %d B row + 8 B index) x B=%d rows gathered from a %d-row ring (%.1f GB)'
               % (row_bytes, B, rows, rows * ds.row_bytes / 1e9), traffic_key=key)
  r['transitions_per_sec'] = round(B / (us * 1e-6), 1)
  r['entry'] = 'gather_%s_%d' % ('wide' if atari else 'narrow', B)
  r['profile'] = 'profiles/r03_%s_kernel_stats.csv' % r['entry']
  del ds
  torch.cuda.empty_cache()
  return r


# ------------------------------------------------------------------------------ DDPG / DQN learner loops
def bench_offpolicy(args, rank, world, dev, atari):
  steps = args.steps if args.steps is not None else 400
  # the first ~0.1 s of host-bound stepping on a fresh box runs at half speed (seen on every first run of a
  # gpurun call, whatever the workload): warm up past it
  warmup = args.warmup if args.warmup is not None else 400
  if not args.no_gemm_tuning:
    from torchrl_amd.tuning import enable_gemm_tuning
    enable_gemm_tuning()     # 32-256-row GEMMs: the BLAS default pick is 4-5x off its best kernel
  g = torch.Generator(device=dev).manual_seed(100 + rank)
  torch.manual_seed(0)
  if atari:
    # the largest ring that fits one GPU (BASELINE config 4: 10 M transitions only fit sharded, 70.6 GB per GPU on 8):
    # 4,000,000 records x 56,480 B = 226 GB of the 288 GB, or what 85 % of the free HBM holds
    if args.capacity:
      cap = args.capacity
    else:
      free, _ = torch.cuda.mem_get_info()
      cap = max(8192, min(4_000_000, int(0.85 * free) // 56480 // 8192 * 8192))
    B = args.minibatch or 256
    # frames stay uint8 from the ring to the first layer's operand fetch (csrc/trl_linear_u8.hip); --f32-frames
    # hands the learner float32 frames instead (the cast fused into the gather), the earlier path
    fdt = torch.float32 if args.f32_frames else torch.uint8
    ds = TensorTupleDataset(size=cap, device=dev,
                            storage_dtypes=[torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32],
                            gather_dtypes=[fdt, torch.float32, torch.float32, fdt, torch.float32])
    def cols(k):
      return [torch.randint(0, 256, (k, 28224), device=dev, dtype=torch.uint8, generator=g),
              torch.randint(0, 6, (k, 1), device=dev, generator=g).float(),
              torch.randint(-1, 2, (k, 1), device=dev, generator=g).float(),
              torch.randint(0, 256, (k, 28224), device=dev, dtype=torch.uint8, generator=g),
              (torch.rand(k, 1, device=dev, generator=g) < 0.005).float()]
    fill_ring(ds, cap, cols, 8192)
    ctl = DQNController(28224, 6, double_dqn=True, gamma=0.99, lr=1e-3, n_update_interval=10, device=dev,
                        graph=args.use_graph)
    row_bytes, name = 2 * 28224 + 12, 'double_dqn_atari'
    wl = 'double_dqn: %d-row replay of 84x84x4 u8 frames per GPU (%.0f GB), flat-frame QNet 28224-128-128-6, %s' % (
        cap, cap * ds.row_bytes / 1e9, 'float32 frames to the learner' if args.f32_frames else 'uint8 frames to the first layer')
  else:
    cap = args.capacity or 1_000_000
    B = args.minibatch or 128
    ds = TensorTupleDataset(size=cap, device=dev)
    def cols(k):
      return [torch.rand(k, 3, device=dev, generator=g) * 2 - 1, torch.rand(k, 1, device=dev, generator=g) * 2 - 1,
              -16 * torch.rand(k, 1, device=dev, generator=g), torch.rand(k, 3, device=dev, generator=g) * 2 - 1,
              (torch.rand(k, 1, device=dev, generator=g) < 0.01).float()]
    fill_ring(ds, cap, cols, 1 << 18)
    ctl = DDPGController(3, 1, -2.0, 2.0, actor_lr=1e-4, critic_lr=1e-3, gamma=0.99, tau=1e-2, device=dev,
                         graph=args.use_graph)
    row_bytes, name = 36, 'ddpg_pendulum'
    wl = 'ddpg: %d-row Pendulum-shaped f32 replay per GPU, actor/critic MLP 256x256' % cap

  in_place = atari and not args.f32_frames and args.in_place
  if atari and args.dqn_chain:
    from torchrl_amd import _lib
    _lib.lib().trl_offpolicy_set_chain(2)
  if in_place:
    # --in-place: the sampled frames are read where the ring holds them (DQNController.learn_indexed(in_place=True): the
    # gather as the first layer's operand fetch).  Measured slower than the gathered batch, hence opt-in
    if args.sampler == 'device':
      def step():
        ctl.learn_indexed(ds, ds.sample_indices_device(B), in_place=True)
    else:
      def step():
        ctl.learn_indexed(ds, ds.sample_indices(B), in_place=True)
  elif args.sampler == 'device':
    def step():
      ctl.learn(*ds.sample_device(B))   # Philox sampler kernel + HIP gather + fused loss/update, no host RNG work
  else:
    def step():
      ctl.learn(*ds.sample(B))     # MT19937 randperm-prefix (host) + HIP gather + fused loss/update

  dt = timed_steps(step, steps, warmup, world)
  value = B * steps * world / dt
  idx = torch.randint(0, cap, (B,), device=dev, generator=g)
  out = ds._alloc_out(B)
  us = kernel_time_us(lambda: ds._gather_index(idx, out), args.kernel_iters)
  if atari and args.f32_frames:   # frames stored uint8, handed to the learner as float32 (exact cast in the gather)
    per_row = (2 * 28224 + 12) + 8 + (2 * 28224 * 4 + 12)
    roof = roofline('k_gather_wide (u8->f32 cast, narrow tail in the same launch)', B * per_row, us,
                    '(%d B u8 record read + 8 B index + %d B f32 batch written) x B=%d'
                    % (2 * 28224 + 12, 2 * 28224 * 4 + 12, B))
  elif atari:
    roof = roofline('k_gather_wide (one launch: wide chunks + narrow tail)', B * (2 * row_bytes + 8), us,
                    '(2 x %d B + 8 B) x B=%d, frames stay uint8' % (row_bytes, B))
  else:
    roof = roofline('k_gather_narrow', B * (2 * row_bytes + 8), us, '(2 x %d B + 8 B) x B=%d' % (row_bytes, B))
  res = {
      'metric': 'learner transitions/sec (sample+loss+update)', 'value': round(value, 1), 'unit': 'transitions/s',
      'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 4),
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': wl, 'minibatch': B, 'capacity': cap, 'sampler': 'mt19937 randperm prefix on the host (bit-exact; the next draw computed ahead on a helper thread, trl_randperm_prefix_ahead)' if args.sampler == 'mt19937' else
                 'philox on the device (throughput mode, not index-identical to torch.randperm)',
                 'parallelism': 'dp%d' % world, 'name': name,
                 'gemm_selection': 'none (the step calls no BLAS GEMM: fused chains)' if getattr(ctl, '_fused', None) is not None else ('blas default' if args.no_gemm_tuning else 'pytorch tunableop (measured per shape)'),
                 'learn_step': 'hip-graph replay' if args.use_graph else
                 (('eager, whole step in %d launches (csrc/trl_offpolicy.hip)%s' % (9 if not args.f32_frames else 6,
                   '; frames read in place out of the ring (learn_indexed: no gathered batch)' if in_place else '')) if atari
                  else 'eager, the 10 phases of the step in ONE launch (k_chain, csrc/trl_offpolicy.hip)'
                  if getattr(ctl, '_fused', None) is not None else 'eager')},
      'roofline': roof,
  }
  if atari and not args.f32_frames:
    res['config']['first_layer_arithmetic'] = ('uint8 x f32 as three exact bf16 partial products per weight (a uint8 is a bf16, an '
                                               'f32 is three), summed in f32 MFMA accumulators: f32 results, csrc/trl_linear_u8.hip')
  return res, dict(cap=cap, B=B, atari=atari)


def cpu_baseline_offpolicy(info):
  """Reference path on the host (examples/dqn.py:40-42): torch.randperm(n)[:B], fancy-index
  gather of float32 columns, learn() on torch-CPU."""
  from oracle import loss_oracle
  from torchrl_amd.contrib import models
  torch.set_num_threads(1)
  B, atari = info['B'], info['atari']
  torch.manual_seed(0)
  if atari:
    n = 20_000        # float32 frames as the reference stores them: 20k rows = 4.5 GB
    colsz = [28224, 1, 1, 28224, 1]
  else:
    n = info['cap']
    colsz = [3, 1, 1, 3, 1]
  cols = [torch.rand(n, w) for w in colsz]
  cols[1] = torch.randint(0, 6, (n, 1)).float() if atari else cols[1]
  cols[4] = (torch.rand(n, 1) < 0.01).float()
  if atari:
    q, tq = models.QNet(28224, 6), models.QNet(28224, 6)
    opt = torch.optim.Adam(q.parameters(), lr=1e-3)
    fn = lambda b: loss_oracle.dqn_learn(q, tq, opt, *b, gamma=0.99, double_dqn=True)
  else:
    a, ta = models.DDPGActorNet(3, 1, 256), models.DDPGActorNet(3, 1, 256)
    c, tc = models.DDPGCriticNet(3, 1, 256), models.DDPGCriticNet(3, 1, 256)
    ao, co = torch.optim.Adam(a.parameters(), lr=1e-4), torch.optim.Adam(c.parameters(), lr=1e-3)
    fn = lambda b: loss_oracle.ddpg_learn(a, c, ta, tc, ao, co, *b, gamma=0.99, tau=1e-2)
  iters, t0 = 0, time.perf_counter()
  while time.perf_counter() - t0 < 10.0:
    idx = torch.randperm(n)[:B]
    fn([x[idx] for x in cols])
    iters += 1
  dt = time.perf_counter() - t0
  return {'value': round(iters * B / dt, 1), 'unit': 'transitions/s', 'cores': 1, 'kind': 'port',
          'sample': '%d train() steps in %.1f s: torch.randperm(%d)[:%d] + fancy-index + learn() on torch-CPU, 1 thread'
                    % (iters, dt, n, B), 'host_cpus': os.cpu_count()}


def run_only(args, dev):
  """ONE kernel-timing entry, nothing else on the GPU: the process a per-configuration rocprofv3 kernel trace
  (profiles/r02_<entry>_kernel_stats.csv) and the PMC passes (tools/pmc_traffic.py) are taken from."""
  e = args.only
  if e.startswith('gae_'):
    _, temp, layout = e.split('_')
    return gae_roofline(args, dev, args.T, args.N, layout, warm=temp == 'warm', with_ceilings=False)
  if e.startswith('gather_'):
    _, kind, B = e.split('_')
    atari = kind == 'wide'
    return gather_roofline(dev, args.kernel_iters, 200_000 if atari else 1_000_000, int(B), atari=atari)
  if e in ('learn_kernel', 'learn_kernel_bf16x6'):
    ctl, _ = make_ppo_controller(args, dev, products='bf16x6' if e.endswith('bf16x6') else 'f32')
    n = args.minibatch or 1048576
    g = torch.Generator(device=dev).manual_seed(1)
    cols = (torch.randn(n, 3, device=dev, generator=g), torch.randn(n, 1, device=dev, generator=g),
            torch.randn(n, 1, device=dev, generator=g), torch.randn(n, 1, device=dev, generator=g) - 1.0,
            torch.randn(n, 1, device=dev, generator=g))
    return learn_kernel_roofline(args, ctl, cols, n)
  if e.startswith('learn_epoch'):
    # `--only learn_epoch` / `learn_epoch_256`: one k_ppo_epoch launch over 4096 consecutive minibatches of 64 (--minibatch
    # overrides) on the bench's network / the reference's 256-wide one -- the process a kernel trace of that launch is taken from
    if e.endswith('_256'):
      args.hidden = '256'
    ctl, _ = make_ppo_controller(args, dev)
    Bm, K = args.minibatch or 64, 4096
    n = Bm * K
    g = torch.Generator(device=dev).manual_seed(1)
    cols = (torch.randn(n, 3, device=dev, generator=g), torch.randn(n, 1, device=dev, generator=g),
            torch.randn(n, 1, device=dev, generator=g), torch.randn(n, 1, device=dev, generator=g) - 1.0,
            torch.randn(n, 1, device=dev, generator=g))
    row = learn_epoch_row(args, ctl, cols, n, Bm, 0.0, K)
    row.update(entry=e, kernel='k_ppo_epoch', minibatch=Bm, minibatches_per_launch=K, hidden=args.hidden)
    return row
  raise SystemExit('unknown --only entry %r' % e)


def main(argv=None):
  argv = list(sys.argv[1:] if argv is None else argv)
  args = parse(argv)
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and not args.only:
    # the driver's command shape: one process asked for N GPUs.  Become the launcher -- before anything GPU-side loads
    return launch_ranks(args, argv)
  _load()
  if args.force_gradsync:
    from torchrl_amd import config
    config.force_gradsync = True
  if args.no_graph:
    args.graph = 'off'
  # graph replay only where the step is still BLAS + autograd: float32 frames into the flat-frame QNet
  args.use_graph = args.graph == 'on' or (args.graph == 'auto' and args.workload == 'dqn_atari' and args.f32_frames)
  rank, local, world = setup_dist(args)
  dev = torch.device('cuda', local)
  if args.only:
    print(json.dumps(run_only(args, dev)))
    return 0
  cpu_base = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload != 'a2c':
    # the host baseline first, so that the GPU phase of the run is one contiguous stretch (a utilisation sampler sees it)
    cpu_base = cpu_baseline_ppo(ppo_info(args)) if args.workload == 'ppo' else None
  if args.workload == 'ppo':
    res, info = bench_ppo(args, rank, world, dev)
  elif args.workload == 'a2c':
    args.use_graph = args.graph == 'on'
    res, info = bench_a2c(args, rank, world, dev)
    args.no_cpu_baseline = True
  else:
    res, info = bench_offpolicy(args, rank, world, dev, atari=args.workload == 'dqn_atari')
  res['n_gpus'] = args.ranks_seen if world > 1 else world
  res['config']['ranks_seen'] = args.ranks_seen
  res['config']['collective'] = ('%s all-reduce of the flat gradient, one per optimizer step' % ('RCCL' if args.backend == 'nccl' else args.backend)
                                 if args.backend else 'none (one rank)')
  if rank == 0 and world == 1:
    if args.workload == 'ppo' and not args.no_extras:
      res['kernels'] = [
          gather_roofline(dev, args.kernel_iters, 1_000_000, 65536, atari=False),
          gather_roofline(dev, args.kernel_iters, 1_000_000, 128, atari=False),
          gather_roofline(dev, args.kernel_iters, 200_000, 1024, atari=True),
          gather_roofline(dev, args.kernel_iters, 200_000, 256, atari=True),
      ]
    if args.workload == 'ppo' and not args.no_extras:
      # BASELINE configs 3 and 4 (and A2C on the reference's own 256-wide net) in the same driver-run line: each at its
      # own default steps / minibatch / capacity (`bench.py --workload <name>` prints the same object on its own)
      import copy
      res['other_workloads'] = []
      for wl, hidden in (('ddpg', None), ('dqn_atari', None), ('a2c', '256')):
        a2 = copy.copy(args)
        a2.workload, a2.steps, a2.warmup, a2.minibatch, a2.capacity = wl, None, None, None, None
        a2.use_graph = False
        if hidden:
          a2.hidden = hidden
        torch.cuda.empty_cache()
        try:
          r2, _ = bench_a2c(a2, rank, world, dev) if wl == 'a2c' else bench_offpolicy(a2, rank, world, dev, atari=wl == 'dqn_atari')
          res['other_workloads'].append({k: r2[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'config', 'roofline')
                                         if k in r2})
        except Exception as e:  # noqa: BLE001  (an extra must not take the headline line down)
          res['other_workloads'].append({'workload': wl, 'error': str(e)[:300]})
      torch.cuda.empty_cache()
      try:
        res['wide_nets'] = wide_nets(dev)
      except Exception as e:  # noqa: BLE001
        res['wide_nets'] = {'error': str(e)[:300]}
    if not args.no_cpu_baseline:
      res['cpu_baseline'] = cpu_base if cpu_base is not None else cpu_baseline_offpolicy(info)
  if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
  if rank == 0:
    print(json.dumps(res), flush=True)
  return 0


if __name__ == '__main__':
  sys.exit(main())
