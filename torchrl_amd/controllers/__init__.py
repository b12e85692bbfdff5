from .controller import Controller, RandomController
