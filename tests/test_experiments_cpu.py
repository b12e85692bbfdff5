"""Host-side experiment plumbing (no device): built-in envs, env set, the run() loop with a
random policy (the reference's test_base_experiment.py:6-17 and test_wrappers.py assertions)."""
import numpy as np
import pytest

from torchrl_amd.envs import CartPoleEnv, PendulumEnv, ParallelEnvs, make_env, Box, Discrete
from torchrl_amd.experiments import BaseExperiment, ScalarLog, log_dict


def test_cartpole_dynamics_and_termination():
  env = CartPoleEnv(max_steps=500)
  env.seed(0)
  obs = env.reset()
  assert obs.shape == (4,) and obs.dtype == np.float32 and np.all(np.abs(obs) <= 0.05)
  steps, done = 0, False
  while not done:
    obs, rew, done, info = env.step(1)          # always push right: the pole falls quickly
    assert rew == 1.0 and isinstance(info, dict)
    steps += 1
  assert 5 <= steps < 60
  assert abs(obs[2]) > env.theta_limit or abs(obs[0]) > env.x_limit
  # same seed, same trajectory
  a, b = CartPoleEnv(), CartPoleEnv()
  a.seed(3); b.seed(3)
  assert np.array_equal(a.reset(), b.reset())
  assert np.array_equal(a.step(0)[0], b.step(0)[0])


def test_pendulum_dynamics():
  env = PendulumEnv(max_steps=200)
  env.seed(1)
  obs = env.reset()
  assert obs.shape == (3,) and abs(obs[0] ** 2 + obs[1] ** 2 - 1) < 1e-6
  total, done, n = 0.0, False, 0
  while not done:
    obs, rew, done, _ = env.step(np.array([5.0]))      # clipped to the torque limit
    assert rew <= 0 and abs(obs[2]) <= env.max_speed + 1e-6
    total += rew
    n += 1
  assert n == 200 and total < 0
  assert env.action_space.shape == (1,) and float(env.action_space.high[0]) == 2.0


def test_monitor_and_env_set():
  envs = ParallelEnvs(lambda seed=None: make_env('CartPole-v1', seed=seed), n_envs=3, base_seed=10)
  assert envs.n_procs == 3 and envs.action_space.n == 2 and envs.observation_space.shape == (4,)
  obs = envs.reset(range(3))
  assert len(obs) == 3 and not np.array_equal(obs[0], obs[1])      # rank r seeded base_seed + r
  out = envs.step([0, 1, 2], [0, 1, 0])
  assert len(out) == 3 and all(len(o) == 4 for o in out)
  info = envs.exec_remote('info', proc_list=[1])[0]
  assert info['len'] == 1 and info['return'] == 1.0
  envs.reset([1])
  assert envs.exec_remote('info', proc_list=[1])[0]['len'] == 0
  with pytest.raises(KeyError):
    make_env('Nope-v0')
  assert isinstance(envs.action_space, Discrete) and isinstance(envs.observation_space, Box)
  envs.close()


@pytest.mark.parametrize('env_id,n_envs,n_frames', [('CartPole-v1', 1, 200), ('Pendulum-v0', 4, 400)])
def test_random_policy_run_counts_frames(env_id, n_envs, n_frames):
  exp = BaseExperiment(env_id=env_id, n_envs=n_envs, n_frames=n_frames, seed=0, cuda=False)
  exp.run()
  assert exp._cur_frames == n_frames
  if env_id == 'CartPole-v1':
    assert exp.logger.counts.get('episode/return', 0) >= 1      # random CartPole episodes are short


def test_log_dict_skips_non_scalars_and_none():
  import torch
  log = ScalarLog()
  log_dict(log, dict(a=1.0, b=None, c=torch.zeros(3), d=torch.tensor(2.0)), tag='train', step=7)
  assert set(log.last) == {'train/a', 'train/d'} and log.last['train/a'] == (1.0, 7)
