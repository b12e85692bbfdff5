"""ctypes binding of libtrl_hip.so (declarations: include/trl_hip.h).

There is no fallback: if the library is missing, or a kernel entry point is called
without a HIP device, the call raises.  Build the library with
`python -m torchrl_amd._build` (or `__graft_entry__.build()`).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'lib', 'libtrl_hip.so')
# tools/ select the -DTRL_TUNING diagnostic build (lib/libtrl_hip_tune.so) with TRL_LIB=tune; read once, at load
if os.environ.get('TRL_LIB') == 'tune':
  LIB_PATH = os.path.join(_PKG, 'lib', 'libtrl_hip_tune.so')

TRL_MAX_COLS = 8
TRL_F32, TRL_U8 = 0, 1
TRL_GAE_REFERENCE, TRL_GAE_MASKED = 0, 1
TRL_LAYOUT_TN, TRL_LAYOUT_NT, TRL_LAYOUT_BLOCKED = 0, 1, 2
TRL_LOSS_MSE, TRL_LOSS_HUBER = 0, 1


class RingLayout(C.Structure):
  _fields_ = [('ncols', C.c_int32),
              ('col_off', C.c_int32 * TRL_MAX_COLS),
              ('col_bytes', C.c_int32 * TRL_MAX_COLS),
              ('col_dtype', C.c_int32 * TRL_MAX_COLS)]


class AdamState(C.Structure):
  """mirror of trl_adam_state"""
  _fields_ = [('step', C.c_int64), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double),
              ('eps', C.c_double), ('grad_scale', C.c_double), ('clip_value', C.c_double),
              ('derived', C.c_float * 8)]


class MlpDesc(C.Structure):
  """mirror of trl_mlp_desc"""
  _fields_ = [('depth', C.c_int32), ('dims', C.c_int32 * 4), ('w', C.c_void_p * 3), ('b', C.c_void_p * 3)]


class AdamArgs(C.Structure):
  """mirror of trl_adam_args"""
  _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p),
              ('n', C.c_int64), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double),
              ('step', C.c_int64), ('grad_scale', C.c_double), ('clip_value', C.c_double), ('dev_state', C.c_void_p)]


class RolloutCols(C.Structure):
  """mirror of trl_rollout_cols"""
  _fields_ = [('ncols', C.c_int32), ('width', C.c_int32 * TRL_MAX_COLS), ('block_envs', C.c_int32 * TRL_MAX_COLS),
              ('group_stride', C.c_int64 * TRL_MAX_COLS), ('dst', C.c_void_p * TRL_MAX_COLS)]


class TrlError(RuntimeError):
  pass


_vp, _i64, _i32, _f64 = C.c_void_p, C.c_int64, C.c_int32, C.c_double

# name -> (restype, argtypes); every symbol declared in include/trl_hip.h
PROTOTYPES = {
    'trl_version': (C.c_int, []),
    'trl_last_error': (C.c_char_p, []),
    'trl_ring_push': (C.c_int, [_vp, _i64, _i64, C.POINTER(RingLayout), _i64, C.POINTER(_vp), _i64, _vp]),
    'trl_gather_rows': (C.c_int, [_vp, _i64, _i64, C.POINTER(RingLayout), _i64, _i64, _vp, _i64,
                                  C.POINTER(_vp), C.POINTER(_i32), _vp]),
    'trl_gather_range': (C.c_int, [_vp, _i64, _i64, C.POINTER(RingLayout), _i64, _i64, _i64, _i64,
                                   C.POINTER(_vp), C.POINTER(_i32), _vp]),
    'trl_mt19937_seed': (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.c_uint64]),
    'trl_randperm_prefix': (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _i64, _i64, _vp]),
    'trl_randperm_prefix_direct': (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _i64, _i64, _vp]),
    'trl_randperm_prefix_ahead': (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _i64, _i64, _vp, _i64,
                                             C.POINTER(_i32)]),
    'trl_sample_indices': (C.c_int, [_i64, _i64, C.c_uint64, _vp, _vp, _vp]),
    'trl_gae_reverse_scan': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _i32, _i32, _vp, _vp, _vp]),
    'trl_gae_reverse_scan_blocked': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _f64, _f64, _i32, _i32, _i64,
                                               _vp, _vp, _vp]),
    'trl_loss_workspace_bytes': (_i64, []),
    'trl_td_loss': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _i32, _f64, _vp, _vp, _vp, _vp]),
    'trl_ddpg_critic_loss': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f64, _i32, _f64, _vp, _vp, _vp, _vp]),
    'trl_ppo_loss': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _f64,
                               _vp, _vp, _vp, _vp, _vp, _vp]),
    'trl_rollout_push': (C.c_int, [_vp, _i64, C.POINTER(RolloutCols), _vp]),
    'trl_act_epsilon_greedy': (C.c_int, [_vp, _i64, _i32, _f64, C.c_uint64, C.c_uint64, _vp, _vp]),
    'trl_act_normal': (C.c_int, [_vp, _vp, _i64, _i32, C.c_uint64, C.c_uint64, _vp, _vp]),
    'trl_act_categorical': (C.c_int, [_vp, _i64, _i32, C.c_uint64, C.c_uint64, _vp, _vp]),
    'trl_act_ou_noise': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _f64, _f64, _f64, _f64, _f64, C.c_uint64, C.c_uint64,
                                   _vp, _vp]),
    'trl_ppo_loss_static': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _f64,
                                      _vp, _vp, _vp, _vp, _vp]),
    'trl_normal_logp': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    'trl_a2c_loss': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp]),
    'trl_adam_step': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _i64, _f64, _f64, _vp]),
    'trl_adam_step_dev': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'trl_polyak': (C.c_int, [_vp, _vp, _i64, _f64, _vp]),
    'trl_linear_fwd': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    'trl_linear_bwd_workspace_bytes': (_i64, [_i64, _i32, _i32]),
    'trl_linear_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'trl_stage_indices': (C.c_int, [_vp, _vp, _i64, _vp]),
    'trl_linear_u8_fwd_workspace_bytes': (_i64, [_i64, _i32, _i32]),
    'trl_linear_u8_fwd': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    'trl_linear_u8_wgrad': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    'trl_mlp_fwd': (C.c_int, [C.POINTER(MlpDesc), _vp, _i64, C.POINTER(_vp), _vp]),
    'trl_mlp_bwd_workspace_bytes': (_i64, [C.POINTER(MlpDesc), _i64]),
    'trl_mlp_bwd': (C.c_int, [C.POINTER(MlpDesc), _vp, C.POINTER(_vp), _vp, _i64, _vp, C.POINTER(_vp), _vp, _vp]),
    'trl_mlp_fwd_group': (C.c_int, [C.POINTER(MlpDesc), _i32, C.POINTER(_vp), _i64, C.POINTER(_vp), _vp]),
    'trl_mlp_bwd_group': (C.c_int, [C.POINTER(MlpDesc), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i64,
                                    C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    'trl_ppo_learn_workspace_bytes': (_i64, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _i64]),
    'trl_a2c_learn_fused': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64,
                                      _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(AdamArgs), _vp, _vp]),
    'trl_ppo_forward_fused': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp,
                                        _vp]),
    'trl_ppo_learn_fused': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                      _i32, _f64, _f64, _f64, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, C.POINTER(AdamArgs),
                                      _vp, _vp]),
    'trl_ppo_learn_fused_bx6': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                          _i32, _f64, _f64, _f64, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp,
                                          C.POINTER(AdamArgs), _vp, _vp]),
    'trl_a2c_learn_fused_bx6': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _i64, _i32, _f64,
                                          _f64, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(AdamArgs), _vp, _vp]),
    'trl_ppo_forward_fused_bx6': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp,
                                            _vp]),
    'trl_bx6_product_probe': (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    'trl_ppo_learn_epoch_workspace_bytes': (_i64, [_i64, _i64, _i32]),
    'trl_ppo_learn_epoch_supported': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _i32]),
    'trl_ppo_learn_epoch': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                      _i32, _i32, _f64, _f64, _f64, _vp, C.POINTER(AdamArgs), _vp, _vp]),
    'trl_ppo_learn_tiles_workspace_bytes': (_i64, [_i64, _i64]),
    'trl_ppo_learn_tiles': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64,
                                      _f64, _f64, _vp, C.POINTER(AdamArgs), _i32, _vp, _vp]),
    'trl_a2c_learn_tiles': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _vp,
                                      C.POINTER(AdamArgs), _i32, _vp, _vp]),
    'trl_ac_forward_tiles': (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    'trl_ddpg_learn_workspace_bytes': (_i64, [C.POINTER(MlpDesc), C.POINTER(MlpDesc), _i64]),
    'trl_offpolicy_set_chain': (None, [_i32]),
    'trl_ddpg_learn_fused': (C.c_int, [C.POINTER(MlpDesc)] * 4 + [_vp] * 5 + [_i64, _f64, _i32, _f64, _vp,
                                       C.POINTER(AdamArgs), C.POINTER(AdamArgs), _vp, _vp, _f64, _vp, _vp]),
    'trl_dqn_learn_workspace_bytes': (_i64, [C.POINTER(MlpDesc), _i64]),
    'trl_dqn_learn_fused': (C.c_int, [C.POINTER(MlpDesc)] * 2 + [_vp] * 5 + [_i64, _f64, _i32, _i32, _f64, _vp,
                                      C.POINTER(AdamArgs), _vp, _vp]),
    'trl_dqn_learn_u8_workspace_bytes': (_i64, [C.POINTER(MlpDesc), _i64]),
    'trl_dqn_learn_u8_ring_workspace_bytes': (_i64, [C.POINTER(MlpDesc), _i64]),
    'trl_dqn_learn_fused_u8_ring': (C.c_int, [C.POINTER(MlpDesc)] * 2 + [_vp, _i64, _i64, C.POINTER(RingLayout), _i64, _i64, _vp,
                                              _i64, _f64, _i32, _i32, _f64, _vp, C.POINTER(AdamArgs), _vp, _vp]),
    'trl_dqn_learn_fused_u8': (C.c_int, [C.POINTER(MlpDesc)] * 2 + [_vp] * 5 + [_i64, _f64, _i32, _i32, _f64, _vp,
                                      C.POINTER(AdamArgs), _vp, _vp]),
}

_lib = None


def lib():
  """The loaded library; raises with a build hint if it is not there."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise TrlError('%s not found: build it with `python -m torchrl_amd._build` '
                     '(there is no CPU fallback)' % LIB_PATH)
    # torch first: its bundled HIP runtime must be the one in the process.  Loaded the other way
    # round, libtrl_hip.so pulls in the system libamdhip64 and torch then finds no device.
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
      fn = getattr(handle, name)   # AttributeError if the symbol is missing
      fn.restype = res
      fn.argtypes = args
    _lib = handle
  return _lib


def check(rc):
  if rc != 0:
    raise TrlError('libtrl_hip error %d: %s' % (rc, lib().trl_last_error().decode(errors='replace')))


def stream_ptr(device=None):
  """hipStream_t of torch's current stream on `device` as an integer."""
  import torch
  return torch.cuda.current_stream(device).cuda_stream


def require_device(t, what):
  """Kernel entry points take device memory only."""
  if not t.is_cuda:
    raise TrlError('%s must live on a HIP device (got %s); torchrl_amd has no CPU path' % (what, t.device))
  return t


def ptr(t):
  return None if t is None else t.data_ptr()
