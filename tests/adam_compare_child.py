"""Child process of tests/test_gpu_epoch.py::test_fast_adam_is_trl_adam_elem_bit_for_bit (diagnostic build, TRL_LIB=tune):
the learn_epoch kernel's four-wide Adam (explicit division refinements, csrc/trl_epoch.hip) against trl_adam_elem -- the
update every other kernel of the library applies, itself checked against torch.optim.Adam -- on a few million elements
spanning the magnitudes a learner sees; every bit of parameter, clipped gradient and both moments must agree."""
import ctypes as C
import sys

import torch

from torchrl_amd import _lib

assert _lib.LIB_PATH.endswith('libtrl_hip_tune.so')
lib = _lib.lib()
lib.trl_tune_adam_compare.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_double] * 4 + [C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
lib.trl_tune_adam_compare.restype = C.c_int
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n = 1 << 22
total = 0
for step, clip, lr, eps, gscale, vscale in [(1, 0.05, 3e-4, 1e-8, 1.0, 0.0), (2, 1.0, 1e-3, 1e-8, 1e-3, 1e-6), (37, 0.0, 3e-4, 1e-8, 10.0, 1.0),
                                            (1000, 0.5, 1e-2, 1e-8, 1e-6, 1e-12), (250000, 1.0, 1e-4, 1e-5, 1e-2, 1e-4),
                                            (5, 0.05, 3e-4, 1e-8, 1e-20, 1e-30)]:
  p = torch.randn(n, device=dev, generator=g)
  grad = torch.randn(n, device=dev, generator=g) * gscale * torch.exp(4 * torch.randn(n, device=dev, generator=g))
  m = torch.randn(n, device=dev, generator=g) * gscale if step > 1 else torch.zeros(n, device=dev)
  v = (torch.randn(n, device=dev, generator=g) ** 2) * vscale * torch.exp(6 * torch.randn(n, device=dev, generator=g))
  v[::97] = 0.0
  grad[::89] = 0.0
  bad = torch.zeros(1, dtype=torch.int64, device=dev)
  rc = lib.trl_tune_adam_compare(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, 0.9, 0.999, eps, step, clip,
                                 bad.data_ptr(), torch.cuda.current_stream().cuda_stream)
  assert rc == 0
  torch.cuda.synchronize()
  print('step %d clip %g: %d of %d elements differ' % (step, clip, int(bad), n))
  total += int(bad)
assert total == 0, total
print('adam compare child ok')
sys.exit(0)
