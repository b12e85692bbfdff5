"""Discounted return / GAE on the CPU (TEST INFRASTRUCTURE).

Follows a2c_controller.py:38-43 and ppo_controller.py:41-46:
    gae = 0.0
    for step in reversed(range(T)):
      delta = reward[step] + gamma * values[step + 1] - values[step]
      gae = delta + gamma * lmbda * gae
      returns[step] = gae + values[step]
with float32 tensors and Python-float hyper-parameters, i.e. gamma is narrowed to float32 at
the multiply, gamma * lmbda is a double product narrowed to float32, and every operation
rounds to float32 on its own.  `advantages = returns - values` is examples/ppo.py:54.
"""
import numpy as np


def gae_numpy(reward, value, bootstrap, gamma, lmbda, done=None, masked=False):
  """reward, value [T, N] float32, bootstrap [N] -> (returns, adv) [T, N], vectorised over
  envs, sequential in t with the reference's op order (numpy does not fuse)."""
  reward = np.asarray(reward, np.float32)
  value = np.asarray(value, np.float32)
  T, N = reward.shape
  g = np.float32(gamma)
  gl = np.float32(gamma * lmbda)
  one = np.float32(1.0)
  returns = np.empty((T, N), np.float32)
  gae = np.zeros(N, np.float32)
  nxt = np.asarray(bootstrap, np.float32).reshape(N)
  for t in range(T - 1, -1, -1):
    if masked:
      nd = one - np.asarray(done[t], np.float32)
      delta = (reward[t] + (g * nxt) * nd) - value[t]
      gae = delta + (gl * nd) * gae
    else:
      delta = (reward[t] + g * nxt) - value[t]
      gae = delta + gl * gae
    returns[t] = gae + value[t]
    nxt = value[t]
  return returns, returns - value


def compute_return_loop(values_ext, reward, gamma, lmbda):
  """The reference's per-env Python loop on torch-CPU tensors, as it stands in
  ppo_controller.py:41-46 (values_ext = values with the bootstrap row appended, [T+1, 1]).
  This is the CPU baseline bench.py times: one call per env, T tiny-tensor iterations."""
  import torch
  returns = reward.new_zeros((len(reward), 1))
  gae = 0.0
  for step in reversed(range(len(reward))):
    delta = reward[step] + gamma * values_ext[step + 1] - values_ext[step]
    gae = delta + gamma * lmbda * gae
    returns[step] = gae + values_ext[step]
  return returns
