"""PPO experiment on device-resident rollouts (reference: examples/ppo.py:9-74)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torchrl_amd.experiments import BaseExperiment  # noqa: E402
from torchrl_amd.contrib.controllers import PPOController  # noqa: E402
from torchrl_amd.utils.rollout import RolloutBuffer, env_major  # noqa: E402


class PPOExperiment(BaseExperiment):
  def __init__(self, gamma=0.99, rollout_steps=10, alpha=0.5, lr=3e-4, beta=1e-3, lmbda=1.0, clip_ratio=0.2,
               max_grad_norm=1.0, batch_size=32, n_epochs=4, graph=False, mask_dones=False, **kwargs):
    self._controller_args = dict(lr=lr, gamma=gamma, lmbda=lmbda, alpha=alpha, beta=beta, clip_ratio=clip_ratio,
                                 max_grad_norm=max_grad_norm, graph=graph)
    # mask_dones=True applies (1 - done) inside the rollout (standard GAE); the reference only looks at
    # the last step's done (a2c_controller.py:32), which is the default here too
    self.mask_dones = mask_dones
    kwargs['n_train_interval'] = kwargs.get('n_envs', 1) * rollout_steps
    super().__init__(**kwargs)
    self.rollout = RolloutBuffer(self.envs.n_procs, device=self.device)
    self.batch_size = batch_size
    self.n_epochs = n_epochs

  def store(self, transition_list):
    self.rollout.extend(transition_list)

  def build_controller(self):
    return PPOController(self.envs.observation_space.shape[0], self.envs.action_space.shape[0],
                         **self._controller_args, device=self.device)

  def train(self):
    obs, action, reward, done, last_next_obs = self.rollout.tensors()
    # all envs at once; returns/advantages come back env-major like the reference's torch.cat
    returns, adv, log_probs, _ = self.controller.compute_return_batched(
        obs, action, reward, last_next_obs, done[-1], done=done, masked=self.mask_dones, env_major_out=True)
    cols = [env_major(obs), env_major(action), returns.reshape(-1, 1), env_major(log_probs), adv.reshape(-1, 1)]
    self.rollout.truncate()
    # the reference's `for epoch: for minibatch in loader: controller.learn(*minibatch)` (examples/ppo.py:70-73) as one
    # launch: learn_epoch walks the consecutive minibatches of every epoch with the weights resident on the CU (it runs
    # the loop of learn() calls itself where the networks are outside the kernel's shapes); the logged values are the
    # last minibatch's, as the reference's loop leaves them
    o, a, r, lp, ad = cols
    info = self.controller.learn_epoch([o, a, None, None, None, r, lp, ad], self.batch_size, self.n_epochs)
    return {k: v[-1] for k, v in info.items()}


if __name__ == '__main__':
  exp = PPOExperiment(env_id='Pendulum-v0', n_envs=16, n_frames=int(2e5), rollout_steps=20, lmbda=0.95,
                      batch_size=64, seed=0)
  exp.run()
  rets = [r for _, r in exp.logger.history['episode/return']]
  k = max(1, len(rets) // 10)
  print('episodes %d  mean return first 10%%: %.1f  last 10%%: %.1f' % (len(rets), sum(rets[:k]) / k, sum(rets[-k:]) / k))
  print({k: (float(v[0]), v[1]) for k, v in exp.logger.last.items()})
