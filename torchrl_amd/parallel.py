"""Data-parallel learner: one process per GPU, the gradient all-reduce is the only collective.

Every loss of the four controllers is a batch mean (dqn_controller.py:78,
ddpg_controller.py:105,112, a2c_controller.py:53-57, ppo_controller.py:62-66), so with equal
per-rank batches the global gradient is the average of the per-rank gradients.  Each rank
owns a replay shard / a slice of the env axis; after backward the flat gradient buffer of a
module (optim.FlatParams) is summed with ONE all-reduce (RCCL over xGMI when the backend is
"nccl"), and the 1/world_size scale and PPO's clip_grad_value_ are folded into the Adam
kernel, i.e. applied after the reduction as single-GPU semantics require
(ppo_controller.py:71-74).  The reference has no counterpart (it is single-process).

Payloads are small (10 KB for the PPO net .. 14.5 MB for a QNet on flat Atari frames), so the
collective is latency-bound: one bucket, no overlap machinery.
"""
import torch
import torch.distributed as dist


class GradSync:
  """Sums flat gradient buffers across ranks; `scale` is what the update must multiply by."""

  def __init__(self, process_group=None):
    from . import config
    self.group = process_group
    ready = dist.is_available() and dist.is_initialized()
    # config.force_gradsync keeps the collective in the step at world_size 1, so that a one-GPU box
    # can rehearse the RCCL call sequence (stream ordering against replayed HIP graphs)
    force = ready and bool(config.force_gradsync)
    self.enabled = ready and (dist.get_world_size(process_group) > 1 or force)
    self.world_size = dist.get_world_size(process_group) if self.enabled else 1
    self.scale = 1.0 / self.world_size
    # RCCL collectives are stream-ordered: the all-reduce is enqueued behind the backward on the
    # device and the host never waits.  gloo stages device tensors through the host; an all-reduce
    # issued behind replayed HIP graphs was seen to stall there unless the compute stream is
    # drained first, so for non-RCCL backends (CPU rehearsal of the N-rank path) the drain stays.
    # config.async_allreduce = False / True overrides either way.
    backend = dist.get_backend(process_group) if self.enabled else ''
    drain = backend != 'nccl'
    if config.async_allreduce is not None:
      drain = not config.async_allreduce
    self._host_staged = self.enabled and drain

  def broadcast_params(self, *flats, src=0):
    """Make every rank start from rank `src`'s weights."""
    if self.enabled:
      for f in flats:
        dist.broadcast(f.data, src=src, group=self.group)

  def all_reduce(self, *flats):
    """Sum the gradients in place; returns the scale for the update (1 / world_size)."""
    for f in flats:
      f.collect_grads()
    if self.enabled:
      if self._host_staged and flats[0].grad.is_cuda:
        torch.cuda.current_stream(flats[0].device).synchronize()
      for f in flats:
        dist.all_reduce(f.grad, op=dist.ReduceOp.SUM, group=self.group)
    return self.scale


def shard_range(n, rank, world_size):
  """[lo, hi) of the contiguous slice of n units (envs, replay rows) that `rank` owns."""
  base, rem = divmod(n, world_size)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)
