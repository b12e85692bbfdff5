"""HIP-graph capture of a learner step.

A learn() call on small MLPs is a few dozen kernels of a few microseconds each, so eager
execution is bound by the host's launch rate.  `GraphedStep` captures one step body (forward,
fused loss, backward, fused update -- all on torch's capture stream, the C-ABI kernels
included since they launch on torch's current stream and neither allocate nor synchronise)
into a HIP graph per input-shape signature and replays it: per call the host then issues one
multi-tensor copy into the static inputs and one graph launch.

The body must not touch host-side state (Python counters, schedules): it runs a few times for
warm-up before capture, and the device tensors listed in `state` are restored afterwards so
the warm-up leaves no trace (step counts and weights stay in lock-step with the reference).
"""
import os
import sys

import torch


def _trace(msg):
  from . import config
  if config.graph_trace:
    print('[graph pid %d] %s' % (os.getpid(), msg), file=sys.stderr, flush=True)


class GraphedStep:
  def __init__(self, body, state, device, warmup=2):
    self.body = body
    self.state = state            # callable -> list of device tensors the body mutates
    self.device = torch.device(device)
    self.warmup = warmup
    self._cache = {}

  @staticmethod
  def _key(inputs):
    return tuple(None if t is None else (tuple(t.shape), t.dtype) for t in inputs)

  def _capture(self, inputs):
    static_in = [None if t is None else t.detach().clone() for t in inputs]
    saved = [t.clone() for t in self.state()]
    side = torch.cuda.Stream(self.device)
    side.wait_stream(torch.cuda.current_stream(self.device))
    with torch.cuda.stream(side):
      for _ in range(self.warmup):
        self.body(*static_in)
    torch.cuda.current_stream(self.device).wait_stream(side)
    _trace('warm-up issued')
    for t, s in zip(self.state(), saved):
      t.copy_(s)
    torch.cuda.synchronize(self.device)
    _trace('warm-up done, capturing')
    graph = torch.cuda.CUDAGraph()
    # thread_local: other threads of the process (collective watchdogs) may keep calling HIP
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
      outs = self.body(*static_in)
    _trace('captured')
    live_in = [t for t in static_in if t is not None]
    return live_in, graph, outs

  def __call__(self, *inputs):
    key = self._key(inputs)
    entry = self._cache.get(key)
    if entry is None:
      entry = self._cache[key] = self._capture(inputs)
    static_in, graph, outs = entry
    torch._foreach_copy_(static_in, [t for t in inputs if t is not None])
    graph.replay()
    return outs
