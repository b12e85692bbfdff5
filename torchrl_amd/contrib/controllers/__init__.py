from .dqn_controller import DQNController
from .ddpg_controller import DDPGController
from .a2c_controller import A2CController
from .ppo_controller import PPOController
