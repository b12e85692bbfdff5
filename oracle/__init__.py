"""CPU restatement of the reference's learner hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in torchrl_amd/ imports this package.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may, and only as the checker / the timed CPU baseline.

Pinning: every function here is checked in tests/test_oracle_*.py against golden vectors
that were produced by importing the reference itself (tests/golden/make_golden.py, run in
the build container; fixtures committed under tests/golden/).  The reference's own tests
hold no numeric vectors for this path, so those fixtures are the pin.

  replay_oracle   torchrl/utils/storage.py:7-72            numpy FIFO columns
  sampler_oracle  torch.randperm as called at examples/dqn.py:40   numpy MT19937 + Fisher-Yates
  gae_oracle      a2c_controller.py:29-45, ppo_controller.py:32-51 numpy [T, N] scan and the
                                                           per-env torch loop (CPU baseline)
  loss_oracle     {dqn,ddpg,a2c,ppo}_controller.py learn() torch-CPU autograd + torch.optim.Adam
"""
