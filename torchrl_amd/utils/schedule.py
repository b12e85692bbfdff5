"""Host-side scalar schedules (reference: torchrl/utils/schedule.py).

`value` returns the current value and then advances the schedule; `wrapped_value` reads
without advancing.  Only ExpDecaySchedule is on the learner path (the epsilon of
DQNController.act, dqn_controller.py:44-45, 58).
"""
import math


class Schedule:
  def step(self):
    raise NotImplementedError

  @property
  def wrapped_value(self):
    raise NotImplementedError

  @property
  def value(self):
    current = self.wrapped_value
    self.step()
    return current

  def __repr__(self):
    return str(self.wrapped_value)


class LinearSchedule(Schedule):
  """min_val -> max_val in num_steps equal increments (or max -> min with invert), clamped."""

  def __init__(self, min_val=0.0, max_val=1.0, num_steps=1000, invert=False):
    self.min_val, self.max_val, self.num_steps, self.invert = min_val, max_val, num_steps, invert
    self.incr = float(max_val - min_val) / num_steps * (-1.0 if invert else 1.0)
    self.val = max_val if invert else min_val

  @property
  def wrapped_value(self):
    return self.val

  def step(self):
    nxt = self.val + self.incr
    self.val = max(nxt, self.min_val) if self.invert else min(nxt, self.max_val)


class ExpDecaySchedule(Schedule):
  """end + (start - end) * exp(-k / num_steps) after k steps."""

  def __init__(self, start=1.0, end=1e-2, num_steps=1000):
    self.start, self.end, self.num_steps = start, end, num_steps
    self.val = start
    self.cur_step = 0

  @property
  def wrapped_value(self):
    return self.val

  def step(self):
    self.cur_step += 1
    self.val = self.end + (self.start - self.end) * math.exp(-self.cur_step / self.num_steps)
