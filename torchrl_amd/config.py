"""Process-wide switches of the Python layer.

Every switch is set IN CODE -- `from torchrl_amd import config; config.fused_learn = False`, or the controllers' `fused=`
argument -- and nothing in the product reads the environment to change its path (round 2's TRL_FUSED_LEARN,
TRL_FUSED_MLP, TRL_FORCE_GRADSYNC, TRL_ASYNC_ALLREDUCE, TRL_KERNEL_PUSH and TRL_INDEX_ZERO_COPY variables are gone).
The one environment variable left is TRL_LIB=tune, read once by torchrl_amd/_lib.py: it selects the -DTRL_TUNING
diagnostic build of the library for tools/ and for the tests' child processes; it never changes what the production
library computes.

  fused_learn      controllers built while it is True take the whole-step kernels (csrc/trl_learn.hip, trl_offpolicy.hip,
                   trl_epoch.hip) where their networks allow; False: the layer-by-layer / autograd path.  A controller's
                   own `fused=` argument overrides it.
  learn_products   'f32' (default, the parity mode): the whole-step PPO / A2C kernels multiply on the exact-f32 MFMA.
                   'bf16x6': where the role-split kernel applies (two hidden layers <= 64 wide, <= 4 observations, <= 2
                   actions, batches >= 32,768) its three 64 x 64 products per network run on the bf16 MFMA with both operands
                   split exactly into three bf16 parts and six of the nine partial products kept (f32 accumulators): the
                   accuracy class of an f32 dot product, not the same bits (csrc/trl_learn.hip, k_ac_learn_pair_bx).  A
                   controller's own `products=` argument overrides it.
  fused_mlp        Linear(+ReLU) stacks use the MFMA layer kernels (csrc/trl_mlp.hip) where they win; False: torch's.
  force_gradsync   keep the gradient all-reduce in the step at world size 1 (rehearses the N-rank call sequence on one GPU).
  async_allreduce  None: stream-ordered collective with RCCL, blocking with gloo; True / False forces either.
  kernel_push      small host pushes go through the push kernel reading pinned records (False: async H2D copies).
  index_zero_copy  host-drawn indices of narrow records are read by the gather kernel out of pinned memory.
  graph_trace      print a line per captured HIP graph (diagnostics).
"""

fused_learn = True
learn_products = 'f32'
fused_mlp = True
force_gradsync = False
async_allreduce = None
kernel_push = True
index_zero_copy = True
graph_trace = False
