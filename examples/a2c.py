"""A2C experiment on device-resident rollouts (reference: examples/a2c.py:8-53)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torchrl_amd.experiments import BaseExperiment  # noqa: E402
from torchrl_amd.contrib.controllers import A2CController  # noqa: E402
from torchrl_amd.utils.rollout import RolloutBuffer, env_major  # noqa: E402


class A2CExperiment(BaseExperiment):
  def __init__(self, gamma=0.99, rollout_steps=5, alpha=0.5, lr=3e-4, beta=1e-3, lmbda=1.0, graph=False, mask_dones=False,
               **kwargs):
    self._controller_args = dict(lr=lr, gamma=gamma, lmbda=lmbda, alpha=alpha, beta=beta, graph=graph)
    # mask_dones=True applies (1 - done) inside the rollout (standard GAE); the reference only looks at
    # the last step's done (a2c_controller.py:32), which is the default here too
    self.mask_dones = mask_dones
    kwargs['n_train_interval'] = kwargs.get('n_envs', 1) * rollout_steps
    super().__init__(**kwargs)
    self.rollout = RolloutBuffer(self.envs.n_procs, device=self.device)

  def store(self, transition_list):
    self.rollout.extend(transition_list)

  def build_controller(self):
    return A2CController(self.envs.observation_space.shape[0], self.envs.action_space.n,
                         **self._controller_args, device=self.device)

  def train(self):
    obs, action, reward, done, last_next_obs = self.rollout.tensors()
    returns, _, _ = self.controller.compute_return_batched(obs, reward, last_next_obs, done[-1], done=done,
                                                           masked=self.mask_dones, env_major_out=True)
    self.rollout.truncate()
    return self.controller.learn(env_major(obs), env_major(action), None, None, None, returns.reshape(-1, 1))


if __name__ == '__main__':
  exp = A2CExperiment(env_id='CartPole-v0', n_envs=16, n_frames=int(1e5), rollout_steps=5, seed=0)
  exp.run()
  rets = [r for _, r in exp.logger.history['episode/return']]
  k = max(1, len(rets) // 10)
  print('episodes %d  mean return first 10%%: %.1f  last 10%%: %.1f' % (len(rets), sum(rets[:k]) / k, sum(rets[-k:]) / k))
  print({k: (float(v[0]), v[1]) for k, v in exp.logger.last.items()})
