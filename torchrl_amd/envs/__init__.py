from .spaces import Box, Discrete
from .classic import CartPoleEnv, PendulumEnv, TransitionMonitor, make_env
from .parallel_envs import ParallelEnvs
