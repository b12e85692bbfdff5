from .schedule import Schedule, LinearSchedule, ExpDecaySchedule
from .storage import Transition, TensorTupleDataset, TransitionTupleDataset
from .sampler import randperm_prefix, Mt19937Sampler
