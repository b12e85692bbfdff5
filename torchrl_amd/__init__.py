"""torchrl_amd: the MI355X-native learner hot path behind activatedgeek/torchrl's
controller / replay-buffer API.  See DESIGN.md for scope and INTEGRATION.md for the boundary.

Importing the package does not load the HIP library; the first kernel call does, and it
raises if the library has not been built (`python -m torchrl_amd._build`)."""
__version__ = '0.1.0'
