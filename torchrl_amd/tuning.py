"""GEMM selection for the small-batch learners.

The off-policy controllers learn on 32-256 rows through 128/256-wide layers (dqn_controller.py,
ddpg_controller.py with torchrl/contrib/models.py's widths).  For such shapes the BLAS default is
far from its best kernel: on MI355X a 128 x 256 x 256 product runs 34 us on hipBLASLt's default
pick (a 256 x 128 macro tile) and a DDPG learn() spends more than half of its time in ten of
them.  PyTorch-ROCm can time the candidate kernels per shape once and keep the fastest
(TunableOp); this module switches that on.  It is a process-wide torch setting, hence opt-in:
call `enable_gemm_tuning()` before the first learn() (the HIP-graph path tunes during its eager
warm-up runs, before capture).  Measured: DDPG B = 128 learner step 0.62 -> 0.32 ms.
"""
import torch


def enable_gemm_tuning(results_file=None):
  """Let PyTorch-ROCm pick each GEMM shape's fastest kernel by measurement (TunableOp).
  `results_file`: where to keep the choices between runs (default: a per-process file in the temp dir)."""
  tun = torch.cuda.tunable
  tun.enable(True)
  tun.tuning_enable(True)
  if results_file is None:      # torch writes its choices at exit: keep them out of the working directory
    import os
    import tempfile
    results_file = os.path.join(tempfile.gettempdir(), 'trl_tunableop_%d.csv' % os.getpid())
  tun.set_filename(str(results_file))


def gemm_tuning_enabled():
  return bool(torch.cuda.tunable.is_enabled())
