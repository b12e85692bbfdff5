"""DDPG experiment on the device replay ring (reference: examples/ddpg.py:8-42)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torchrl_amd.experiments import BaseExperiment  # noqa: E402
from torchrl_amd.contrib.controllers import DDPGController  # noqa: E402
from torchrl_amd.utils.storage import TransitionTupleDataset  # noqa: E402


class DDPGExperiment(BaseExperiment):
  def __init__(self, actor_lr=1e-4, critic_lr=1e-3, gamma=0.99, tau=1e-2, batch_size=32, buffer_size=1000,
               n_ou_reset_interval=100000, graph=False, **kwargs):
    self._controller_args = dict(actor_lr=actor_lr, critic_lr=critic_lr, gamma=gamma, tau=tau,
                                 n_reset_interval=n_ou_reset_interval, graph=graph)
    self.buffer = TransitionTupleDataset(size=buffer_size)
    self.batch_size = batch_size
    super().__init__(**kwargs)

  def store(self, transition_list):
    self.buffer.extend(transition_list)

  def build_controller(self):
    space = self.envs.action_space
    return DDPGController(self.envs.observation_space.shape[0], space.shape[0], space.low, space.high,
                          **self._controller_args, device=self.device)

  def train(self):
    if len(self.buffer) < self.batch_size:
      return {}
    return self.controller.learn(*self.buffer.sample(self.batch_size))


if __name__ == '__main__':
  exp = DDPGExperiment(env_id='Pendulum-v0', n_frames=30000, n_train_interval=1, batch_size=128,
                       buffer_size=int(1e6), seed=0)
  exp.run()
  rets = [r for _, r in exp.logger.history['episode/return']]
  k = max(1, len(rets) // 10)
  print('episodes %d  mean return first 10%%: %.1f  last 10%%: %.1f' % (len(rets), sum(rets[:k]) / k, sum(rets[-k:]) / k))
  print({k: (float(v[0]), v[1]) for k, v in exp.logger.last.items()})
