from .base_experiment import BaseExperiment, ScalarLog, log_dict
