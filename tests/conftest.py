import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a HIP device (run on the MI355X box)')
  _build_if_stale_on_cpu()


def _build_if_stale_on_cpu():
  """A fresh checkout has no libtrl_hip.so (built artefacts are git-ignored).  In the build container
  (no GPU, hipcc present) compile it before the CPU suite so that the C-ABI tests have something to load;
  on a GPU box nothing is ever built here: the prebuilt library travels with the tree."""
  import torch
  from torchrl_amd import _build
  if torch.cuda.is_available() or not _build.is_stale():
    return
  try:
    _build.build_library(force=False, verbose=False)
  except Exception as e:  # noqa: BLE001  (no hipcc: the tests that need the library will say so)
    sys.stderr.write('conftest: could not build libtrl_hip.so: %s\n' % e)


def pytest_collection_modifyitems(config, items):
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no HIP device')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


def load_golden(name):
  return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
  return load_golden
