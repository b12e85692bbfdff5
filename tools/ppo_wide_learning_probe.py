#!/usr/bin/env python3
"""PPO on the built-in Pendulum with enough envs that every learn() / compute_return call takes the
whole-step kernels at the reference's default width (256): prints the return trend and wall time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))

from ppo import PPOExperiment  # noqa: E402

if __name__ == '__main__':
  n_frames = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
  exp = PPOExperiment(env_id='Pendulum-v0', n_envs=256, n_frames=n_frames, rollout_steps=32, lmbda=0.95, lr=3e-4,
                      batch_size=1024, n_epochs=4, mask_dones=True, seed=0)
  assert exp.controller._fused is not None
  t0 = time.time()
  exp.run()
  dt = time.time() - t0
  rets = [r for _, r in exp.logger.history['episode/return']]
  k = max(1, len(rets) // 10)
  print('frames %d in %.1f s (%.0f frames/s); episodes %d; mean return first 10%%: %.1f  last 10%%: %.1f'
        % (n_frames, dt, n_frames / dt, len(rets), sum(rets[:k]) / k, sum(rets[-k:]) / k), flush=True)
