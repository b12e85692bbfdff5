#!/usr/bin/env python3
"""Launch each hand-written hot kernel a few times at the bench's sizes, for rocprofv3
(--kernel-trace or --pmc passes).  Usage: python3 tools/kernels_once.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import functional as F  # noqa: E402
from torchrl_amd.utils.storage import TensorTupleDataset  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
T, N = 2048, 4096
r, v, b = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.randn(N, device=dev)
out = (torch.empty(T, N, device=dev), torch.empty(T, N, device=dev))
big = torch.empty(1 << 28, device=dev)   # 1 GiB: flush the 256 MiB Infinity Cache between launches
for _ in range(reps):
  big.fill_(1.0)
  F.gae(r, v, b, 0.99, 0.95, out=out)
torch.cuda.synchronize()

rows = 200_000
ds = TensorTupleDataset(size=rows, device=dev,
                        storage_dtypes=[torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32],
                        gather_dtypes=[torch.uint8, torch.float32, torch.float32, torch.uint8, torch.float32])
for lo in range(0, rows, 8192):
  k = min(8192, rows - lo)
  ds.extend(torch.randint(0, 256, (k, 28224), device=dev, dtype=torch.uint8, generator=g),
            torch.zeros(k, 1, device=dev), torch.zeros(k, 1, device=dev),
            torch.randint(0, 256, (k, 28224), device=dev, dtype=torch.uint8, generator=g),
            torch.zeros(k, 1, device=dev))
idx = torch.randint(0, rows, (1024,), device=dev, generator=g)
o = ds._alloc_out(1024)
for _ in range(reps):
  big.fill_(1.0)
  ds._gather_index(idx, o)
torch.cuda.synchronize()

ds2 = TensorTupleDataset(size=1_000_000, device=dev)
ds2.extend(torch.rand(1_000_000, 3, device=dev), torch.rand(1_000_000, 1, device=dev), torch.rand(1_000_000, 1, device=dev),
           torch.rand(1_000_000, 3, device=dev), torch.rand(1_000_000, 1, device=dev))
idx2 = torch.randint(0, 1_000_000, (65536,), device=dev, generator=g)
o2 = ds2._alloc_out(65536)
for _ in range(reps):
  big.fill_(1.0)
  ds2._gather_index(idx2, o2)
torch.cuda.synchronize()
print('done')
