"""Shared helpers of the test-suite (fixture decoding, module loading)."""
import numpy as np
import torch

FIELDS = ('obs', 'action', 'reward', 'next_obs', 'done')


def state_from(g, prefix):
  """state_dict stored by make_golden.state_arrays under '<prefix>.<key>'."""
  pre = prefix + '.'
  return {k[len(pre):]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith(pre)}


def t(a, device=None):
  x = torch.from_numpy(np.array(a))
  return x if device is None else x.to(device)


def flat(module, grad=False):
  return torch.cat([(p.grad if grad else p.data).reshape(-1) for p in module.parameters()]).detach().cpu().numpy()


def sub(a, stride=16, limit=4000):
  return a if a.size <= limit else a[::stride]


def transitions_from(g, prefix):
  """list[Transition]-shaped tuples from the per-field arrays of one push."""
  cols = [np.array(g[prefix + f]) for f in FIELDS]
  k = cols[0].shape[0]
  return [tuple(c[i] for c in cols) for i in range(k)]
