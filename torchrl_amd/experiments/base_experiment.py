"""Kondo-free experiment loop (reference: torchrl/experiments/base_experiment.py:23-100).

Same run() protocol -- act, step every env, build one Transition per env, reset finished
envs, store(), and every `n_train_interval` frames train() -- and the same overridable hooks
(`build_controller`, `act`, `store`, `train`).  What kondo's `Experiment` base supplied
(`seed`, `cuda`, `logger`) is restated in `_ExperimentBase`; its Spec/CLI machinery is out of
scope.  This is host-side plumbing: the device work is all inside the controller and buffer.
"""
import functools
import random

import numpy as np
import torch

from ..controllers import Controller, RandomController
from ..envs import ParallelEnvs, make_env
from ..utils.storage import Transition


class ScalarLog:
  """Stand-in for the SummaryWriter-like logger kondo hands the reference: keeps the last value
  of every tag (device scalars are kept as they are: no host sync per step)."""

  def __init__(self, history_of=('episode/return', 'episode/len')):
    self.last = {}
    self.counts = {}
    self.history = {t: [] for t in history_of}     # host scalars only (episode statistics)

  def add_scalar(self, tag, value, global_step=None):
    if torch.is_tensor(value):
      assert value.numel() == 1, 'not a scalar'
    self.last[tag] = (value, global_step)
    self.counts[tag] = self.counts.get(tag, 0) + 1
    if tag in self.history and not torch.is_tensor(value):
      self.history[tag].append((global_step, float(value)))

  def close(self):
    pass


def log_dict(logger, info, tag, step=None):
  """Every non-None entry as `tag/key` (base_experiment.py:12-20); non-scalars are skipped."""
  for k, v in info.items():
    if v is not None:
      try:
        logger.add_scalar('%s/%s' % (tag, k), v, global_step=step)
      except AssertionError:
        pass


class _ExperimentBase:
  def __init__(self, seed=None, cuda=None, logger=None, **_unused):
    self.seed = seed
    self.cuda = torch.cuda.is_available() if cuda is None else bool(cuda)
    self.logger = logger if logger is not None else ScalarLog()
    if seed is not None:
      random.seed(seed)
      np.random.seed(seed)
      torch.manual_seed(seed)


class BaseExperiment(_ExperimentBase):
  def __init__(self, env_id=None, n_envs=1, n_frames=int(1e3), n_rand_frames=0,
               n_train_interval=100, **kwargs):
    assert env_id is not None, '"env_id" cannot be None'
    super().__init__(**kwargs)
    self.device = torch.device('cuda' if self.cuda else 'cpu')
    self.n_frames = n_frames
    self.n_rand_frames = n_rand_frames
    self.n_train_interval = n_train_interval
    self.envs = ParallelEnvs(functools.partial(make_env, env_id), n_envs=n_envs, base_seed=self.seed)
    self.controller = self.build_controller()
    self._cur_frames = 0

  def build_controller(self) -> Controller:
    return RandomController(self.envs.action_space)

  def act(self, obs_list):
    if self._cur_frames < self.n_rand_frames:
      return RandomController(self.envs.action_space).act(obs_list)
    return self.controller.act(obs_list)

  def store(self, transition_list):
    """Hook: keep the transitions of one env step."""

  def train(self) -> dict:
    """Hook: one training call; returns scalars to log."""
    return {}

  def run(self):
    procs = list(range(self.envs.n_procs))
    obs_list = self.envs.reset(procs)
    while self._cur_frames < self.n_frames:
      action_list = self.act(obs_list)
      step_list = self.envs.step(procs, action_list)
      transition_list = []
      for i, (obs, action, (next_obs, rew, done, _)) in enumerate(zip(obs_list, action_list, step_list)):
        obs_list[i] = next_obs
        self._cur_frames += 1
        transition_list.append(Transition(obs=obs, action=action, reward=rew, next_obs=next_obs, done=done))
        if done:
          log_dict(self.logger, self.envs.exec_remote('info', proc_list=[i])[0], tag='episode',
                   step=self._cur_frames)
          obs_list[i] = self.envs.reset([i])[0]
      self.store(transition_list)
      if self._cur_frames >= self.n_rand_frames and self._cur_frames % self.n_train_interval == 0:
        log_dict(self.logger, self.train(), tag='train', step=self._cur_frames)
    self.logger.close()
    self.envs.close()
