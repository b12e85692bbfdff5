"""A set of envs stepped together behind the reference's ParallelEnvs surface
(parallel_envs.py:86-116): `reset(proc_list)`, `step(proc_list, action_list)`,
`exec_remote(fn, proc_list)`, `close()`, `n_procs`, `observation_space`, `action_space`.

The reference runs one subprocess per gym env over pipes; env stepping is the slow host-side
outer loop and out of scope here, so the envs simply live in this process (env `rank` is
seeded `base_seed + rank`, parallel_envs.py:94-98)."""


class ParallelEnvs:
  def __init__(self, env_fn, n_envs=1, base_seed=0):
    self.envs = [env_fn(seed=(base_seed or 0) + rank) for rank in range(n_envs)]
    self.n_procs = n_envs
    self.observation_space = self.envs[0].observation_space
    self.action_space = self.envs[0].action_space

  def reset(self, proc_list):
    return [self.envs[i].reset() for i in proc_list]

  def step(self, proc_list, action_list):
    return [self.envs[i].step(a) for i, a in zip(proc_list, action_list)]

  def exec_remote(self, fn, proc_list=None, args=None, kwargs=None):
    out = []
    for i in (proc_list if proc_list is not None else range(self.n_procs)):
      attr = getattr(self.envs[i], fn)
      out.append(attr(*(args or ()), **(kwargs or {})) if callable(attr) else attr)
    return out

  def close(self):
    for e in self.envs:
      e.close()
