"""Shared plumbing of the four controllers."""
import numpy as np
import torch

from ... import _lib
from ...optim import FlatParams, FlatAdam
from ...parallel import GradSync


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


__all__ = ['resolve_device', 'need_hip', 'obs_to_tensor', 'FlatParams', 'FlatAdam', 'GradSync']
