"""Controller contract (reference: torchrl/controllers/controller.py:5-22).

`act(obs_list)` returns one action per observation; `learn(*batch)` returns a dict of
scalars for the logger.  Nothing here touches the device.
"""
import abc


class Controller(metaclass=abc.ABCMeta):
  @abc.abstractmethod
  def act(self, obs):
    raise NotImplementedError

  def learn(self) -> dict:
    """Default: nothing to learn, nothing to log."""
    return {}


class RandomController(Controller):
  """Samples `action_space.sample()` per observation (any object with a `sample()` method;
  the reference annotates it as gym.Space, controller.py:17)."""

  def __init__(self, action_space):
    self.action_space = action_space

  def act(self, obs):
    return [self.action_space.sample() for _ in range(len(obs))]
