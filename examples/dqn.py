"""DQN / Double-DQN experiment on the device replay ring (reference: examples/dqn.py:8-42)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torchrl_amd.experiments import BaseExperiment  # noqa: E402
from torchrl_amd.contrib.controllers import DQNController  # noqa: E402
from torchrl_amd.utils.storage import TransitionTupleDataset  # noqa: E402


class DQNExperiment(BaseExperiment):
  def __init__(self, double_dqn=False, gamma=.99, batch_size=32, lr=1e-3, buffer_size=1000, eps_max=1.0,
               eps_min=1e-2, n_eps_anneal=100, n_update_interval=10, graph=False, **kwargs):
    self._controller_args = dict(double_dqn=double_dqn, gamma=gamma, lr=lr, eps_max=eps_max, eps_min=eps_min,
                                 n_eps_anneal=n_eps_anneal, n_update_interval=n_update_interval, graph=graph)
    self.buffer = TransitionTupleDataset(size=buffer_size)   # binds to the device at the first push
    self.batch_size = batch_size
    super().__init__(**kwargs)

  def store(self, transition_list):
    self.buffer.extend(transition_list)

  def build_controller(self):
    return DQNController(self.envs.observation_space.shape[0], self.envs.action_space.n,
                         **self._controller_args, device=self.device)

  def train(self):
    if len(self.buffer) < self.batch_size:
      return {}
    # == controller.learn(*buffer[torch.randperm(len(buffer))[:batch_size]]) (reference dqn.py:40-42): same indices, same
    # generator state afterwards (learn_indexed(..., in_place=True) would read uint8 frames where the ring holds them)
    return self.controller.learn_indexed(self.buffer, self.buffer.sample_indices(self.batch_size))


if __name__ == '__main__':
  exp = DQNExperiment(env_id='CartPole-v1', n_envs=10, n_frames=20000, n_train_interval=1, batch_size=256,
                      buffer_size=5000, n_eps_anneal=500, seed=0)
  exp.run()
  rets = [r for _, r in exp.logger.history['episode/return']]
  k = max(1, len(rets) // 10)
  print('episodes %d  mean return first 10%%: %.1f  last 10%%: %.1f' % (len(rets), sum(rets[:k]) / k, sum(rets[-k:]) / k))
  print({k: (float(v[0]), v[1]) for k, v in exp.logger.last.items()})
