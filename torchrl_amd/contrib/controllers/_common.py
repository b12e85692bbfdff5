"""Shared plumbing of the four controllers."""
import numpy as np
import torch

from ... import _lib
from ...optim import FlatParams, FlatAdam
from ...parallel import GradSync
from ...graphs import GraphedStep


def resolve_device(device):
  """None -> the current HIP device when one is visible (the reference's None means CPU;
  here the learner lives on the GPU), else CPU for construction / act() only."""
  if device is None:
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() \
        else torch.device('cpu')
  return torch.device(device)


def need_hip(device, what):
  if device.type != 'cuda':
    raise _lib.TrlError('%s runs HIP kernels and needs a HIP device, got %s (no CPU fallback)' % (what, device))


def obs_to_tensor(obs, device):
  """A list/array of observations -> float32 [n, ...] on the device."""
  if torch.is_tensor(obs):
    return obs.to(device=device, dtype=torch.float32)
  return torch.from_numpy(np.asarray(obs)).to(device=device, dtype=torch.float32)


class LearnRunner:
  """Runs a controller's device-side step body eagerly or as a replayed HIP graph.

  body(*tensors) -> tensor of losses; it must do device work only.  With more than one rank the
  body stops after backward and the caller reduces + steps eagerly (`split` is then True)."""

  def __init__(self, body, optims, extra_state, device, use_graph):
    self.body, self.optims = body, optims
    self.graphed = None
    if use_graph:
      if device.type != 'cuda':
        raise _lib.TrlError('graph=True needs a HIP device')
      state = lambda: [t for o in optims for t in o.state_tensors()] + list(extra_state())
      self.graphed = GraphedStep(body, state, device)

  def __call__(self, *tensors):
    if self.graphed is None:
      return self.body(*tensors)
    return self.graphed(*tensors).clone()      # the static output is overwritten by the next replay


__all__ = ['resolve_device', 'need_hip', 'obs_to_tensor', 'FlatParams', 'FlatAdam', 'GradSync', 'LearnRunner']
