#!/usr/bin/env python3
"""Time the fused Linear(+ReLU) kernels against torch's own ops (graph replay, cold-ish)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TRL_LIB', 'tune')   # the -DTRL_TUNING diagnostic build (python -m torchrl_amd._build --tuning)
from torchrl_amd import functional as F  # noqa: E402
from tools.gae_sweep import graph_time_us  # noqa: E402

dev = 'cuda:0'
for (B, K, N) in [(65536, 64, 64), (65536, 3, 64), (65536, 64, 1), (65536, 256, 256), (65536, 128, 128), (4096, 64, 64),
                  (256, 128, 128)]:
  x = torch.randn(B, K, device=dev)
  w = (torch.randn(N, K, device=dev) / K ** 0.5).requires_grad_()
  b = torch.randn(N, device=dev).requires_grad_()
  gy = torch.randn(B, N, device=dev)
  xg = x.clone().requires_grad_()
  with torch.no_grad():
    t_f = graph_time_us([lambda: F.linear_act(x, w, b, relu=True)], 20)
    t_tf = graph_time_us([lambda: torch.relu(torch.nn.functional.linear(x, w, b))], 20)
  from torchrl_amd import _lib
  with torch.no_grad():
    y = F.linear_act(x, w, b, relu=True)
    dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(_lib.lib().trl_linear_bwd_workspace_bytes(B, K, N), dtype=torch.uint8, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    t_b = graph_time_us([lambda: _lib.check(_lib.lib().trl_linear_bwd(
        gy.data_ptr(), y.data_ptr(), x.data_ptr(), w.data_ptr(), B, K, N, 1, dx.data_ptr(), dw.data_ptr(),
        db.data_ptr(), ws.data_ptr(), st()))], 20)
    if (B, K, N) == (65536, 64, 64):
      for dbg in (1, 2, 4, 8, 16, 31):
        os.environ['TRL_MLP_DBG'] = str(dbg)
        t = graph_time_us([lambda: _lib.check(_lib.lib().trl_linear_bwd(
            gy.data_ptr(), y.data_ptr(), x.data_ptr(), w.data_ptr(), B, K, N, 1, dx.data_ptr(), dw.data_ptr(),
            db.data_ptr(), ws.data_ptr(), st()))], 20)
        print('   bwd ablation dbg=%2d: %.1f us' % (dbg, t))
      os.environ.pop('TRL_MLP_DBG')
    wd = w.detach()

    def torch_bwd():
      g = gy * (y > 0)
      return g @ wd, g.t() @ x, g.sum(0)
    t_tb = graph_time_us([torch_bwd], 20)
  mb_f = (B * K + B * N) * 4 / 1e6
  mb_b = (3 * B * N + 2 * B * K) * 4 / 1e6
  print('B=%6d K=%3d N=%3d  fwd %6.1f us (%5.0f GB/s)  torch %6.1f | bwd %6.1f us (%5.0f GB/s)  torch %6.1f'
        % (B, K, N, t_f, mb_f / t_f * 1e3, t_tf, t_b, mb_b / t_b * 1e3, t_tb), flush=True)
