"""Built-in classic-control environments so that the example experiments run without gym
(absent from the image; the reference builds its envs with gym.make, env_utils.py:7-10).

The dynamics are the textbook ones (Barto-Sutton-Anderson cart-pole with Euler integration;
the torque-limited pendulum swing-up), written from the equations.  They are NOT checked
against gym's implementations -- physics parity with gym is unpinned -- and exist to exercise
the learner plumbing end to end (BASELINE config 1).  API: reset() -> obs, step(action) ->
(obs, reward, done, info), the 4-tuple protocol base_experiment.py:73 unpacks.
"""
import numpy as np

from .spaces import Box, Discrete


class CartPoleEnv:
  """Pole on a cart; actions {0: push left, 1: push right}; +1 per step until the pole falls
  (|theta| > 12 deg), the cart leaves the track (|x| > 2.4) or max_steps elapse."""

  def __init__(self, max_steps=500):
    self.gravity, self.masscart, self.masspole = 9.8, 1.0, 0.1
    self.length, self.force_mag, self.tau = 0.5, 10.0, 0.02
    self.theta_limit, self.x_limit = 12 * 2 * np.pi / 360, 2.4
    self.max_steps = max_steps
    high = np.array([2 * self.x_limit, np.finfo(np.float32).max, 2 * self.theta_limit, np.finfo(np.float32).max],
                    dtype=np.float32)
    self.observation_space = Box(-high, high)
    self.action_space = Discrete(2)
    self._rs = np.random.RandomState()
    self.state, self._t = None, 0

  def seed(self, seed=None):
    self._rs = np.random.RandomState(seed)
    self.action_space.seed(seed)

  def reset(self):
    self.state = self._rs.uniform(-0.05, 0.05, size=4)
    self._t = 0
    return self.state.astype(np.float32)

  def step(self, action):
    x, x_dot, th, th_dot = self.state
    force = self.force_mag if int(action) == 1 else -self.force_mag
    total_mass = self.masscart + self.masspole
    pml = self.masspole * self.length
    cos, sin = np.cos(th), np.sin(th)
    tmp = (force + pml * th_dot ** 2 * sin) / total_mass
    th_acc = (self.gravity * sin - cos * tmp) / (self.length * (4.0 / 3.0 - self.masspole * cos ** 2 / total_mass))
    x_acc = tmp - pml * th_acc * cos / total_mass
    x, x_dot = x + self.tau * x_dot, x_dot + self.tau * x_acc
    th, th_dot = th + self.tau * th_dot, th_dot + self.tau * th_acc
    self.state = np.array([x, x_dot, th, th_dot])
    self._t += 1
    done = bool(abs(x) > self.x_limit or abs(th) > self.theta_limit or self._t >= self.max_steps)
    return self.state.astype(np.float32), 1.0, done, {}

  def close(self):
    pass


class PendulumEnv:
  """Torque-limited pendulum swing-up: obs (cos th, sin th, th_dot), action torque in [-2, 2],
  reward -(th^2 + 0.1 th_dot^2 + 0.001 u^2), episodes of max_steps steps."""

  def __init__(self, max_steps=200):
    self.max_speed, self.max_torque, self.dt = 8.0, 2.0, 0.05
    self.g, self.m, self.l = 10.0, 1.0, 1.0
    self.max_steps = max_steps
    high = np.array([1.0, 1.0, self.max_speed], dtype=np.float32)
    self.observation_space = Box(-high, high)
    self.action_space = Box(-self.max_torque, self.max_torque, shape=(1,))
    self._rs = np.random.RandomState()
    self.state, self._t = None, 0

  def seed(self, seed=None):
    self._rs = np.random.RandomState(seed)
    self.action_space.seed(seed)

  def _obs(self):
    th, th_dot = self.state
    return np.array([np.cos(th), np.sin(th), th_dot], dtype=np.float32)

  def reset(self):
    self.state = self._rs.uniform([-np.pi, -1.0], [np.pi, 1.0])
    self._t = 0
    return self._obs()

  def step(self, action):
    th, th_dot = self.state
    u = float(np.clip(np.asarray(action, dtype=np.float64).reshape(-1)[0], -self.max_torque, self.max_torque))
    ang = ((th + np.pi) % (2 * np.pi)) - np.pi
    cost = ang ** 2 + 0.1 * th_dot ** 2 + 0.001 * u ** 2
    th_dot = th_dot + (3 * self.g / (2 * self.l) * np.sin(th) + 3.0 / (self.m * self.l ** 2) * u) * self.dt
    th_dot = float(np.clip(th_dot, -self.max_speed, self.max_speed))
    th = th + th_dot * self.dt
    self.state = np.array([th, th_dot])
    self._t += 1
    return self._obs(), -float(cost), bool(self._t >= self.max_steps), {}

  def close(self):
    pass


_REGISTRY = {
    'CartPole-v0': lambda: CartPoleEnv(max_steps=200),
    'CartPole-v1': lambda: CartPoleEnv(max_steps=500),
    'Pendulum-v0': lambda: PendulumEnv(max_steps=200),
    'Pendulum-v1': lambda: PendulumEnv(max_steps=200),
}


def make_env(env_id, seed=None):
  """Counterpart of make_gym_env (env_utils.py:7-10) for the built-in envs: build and seed."""
  if env_id not in _REGISTRY:
    raise KeyError('unknown env_id %r; built in: %s' % (env_id, sorted(_REGISTRY)))
  env = TransitionMonitor(_REGISTRY[env_id]())
  env.seed(seed)
  return env


class TransitionMonitor:
  """Episode return / length bookkeeping around an env (reference: wrappers.py:5-68):
  `info` = {**last step info, 'return', 'len'}, `is_done`, `obs`."""

  def __init__(self, env):
    self.env = env
    self.observation_space, self.action_space = env.observation_space, env.action_space
    self._ep_return, self._ep_len, self._ep_done, self._ep_info, self._obs = None, None, False, {}, None

  def seed(self, seed=None):
    self.env.seed(seed)

  @property
  def obs(self):
    return self._obs

  @property
  def is_done(self):
    return self._ep_done

  @property
  def info(self):
    return {**self._ep_info, 'return': self._ep_return, 'len': self._ep_len}

  def reset(self):
    self._ep_return, self._ep_len, self._ep_done, self._ep_info = 0.0, 0, False, {}
    self._obs = self.env.reset()
    return self._obs

  def step(self, action):
    next_obs, reward, self._ep_done, self._ep_info = self.env.step(action)
    self._ep_return += reward
    self._ep_len += 1
    self._obs = next_obs
    return next_obs, reward, self._ep_done, self._ep_info

  def close(self):
    self.env.close()
