"""Build libtrl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m torchrl_amd._build            # rebuild if sources are newer than the library
    python -m torchrl_amd._build --force
    python -m torchrl_amd._build --tuning   # lib/libtrl_hip_tune.so: -DTRL_TUNING diagnostic build (tools/ only)

The library is NOT built at import time or inside tests: `__graft_entry__.build()` (or the
command above) builds it here, and the built .so travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB_DIR = os.path.join(PKG, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libtrl_hip.so')
# Diagnostic build: the same sources with -DTRL_TUNING, which compiles in the tuning / ablation switches
# that tools/*_sweep.py drive through environment variables (TRL_GAE_TUNE, TRL_GATHER_*, TRL_MLP_DBG,
# TRL_U8_FWD_WGS).  The production library above contains none of them and never reads the environment.
TUNE_LIB_PATH = os.path.join(LIB_DIR, 'libtrl_hip_tune.so')
SOURCES = ['trl_host.cpp', 'trl_replay.hip', 'trl_gae.hip', 'trl_loss.hip', 'trl_mlp.hip', 'trl_learn.hip',
           'trl_offpolicy.hip', 'trl_linear_u8.hip', 'trl_act.hip', 'trl_epoch.hip']
HEADERS = [os.path.join(CSRC, 'trl_common.h'), os.path.join(ROOT, 'include', 'trl_hip.h')]

# -ffp-contract=off: the parity bar is the reference's op order, one f32 rounding per op.
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off',
         '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def _hipcc():
  for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found: cannot build libtrl_hip.so')


def is_stale(tuning=False):
  lib = TUNE_LIB_PATH if tuning else LIB_PATH
  if not os.path.exists(lib):
    return True
  t = os.path.getmtime(lib)
  deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
  return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, tuning=False):
  lib_path = TUNE_LIB_PATH if tuning else LIB_PATH
  if not force and not is_stale(tuning):
    return lib_path
  os.makedirs(LIB_DIR, exist_ok=True)
  hipcc = _hipcc()
  objs = []
  procs = []
  for s in SOURCES:
    obj = os.path.join(LIB_DIR, os.path.splitext(s)[0] + ('.tune.o' if tuning else '.o'))
    if s.endswith('.cpp'):   # host-only translation unit (no device pass: it uses target_clones)
      cmd = [shutil.which('g++') or 'g++', '-O3', '-std=c++17', '-fPIC', '-Wall', '-I',
             os.path.join(ROOT, 'include'), '-c', os.path.join(CSRC, s), '-o', obj]
    else:
      extra = ['-DTRL_TUNING'] if tuning else []
      if tuning and os.environ.get('TRL_LEARN_STAMP') == '1':     # in-kernel clock stamps (tools/learn_stamps.py)
        extra.append('-DTRL_LEARN_STAMP')
      cmd = [hipcc] + FLAGS + extra + ['-x', 'hip', '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-c',
                               os.path.join(CSRC, s), '-o', obj]
    if verbose:
      print(' '.join(cmd), flush=True)
    procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs.append(obj)
  for s, p in procs:
    out, _ = p.communicate()
    if p.returncode != 0:
      raise RuntimeError('hipcc failed on %s:\n%s' % (s, out.decode(errors='replace')))
    if verbose and out.strip():
      print(out.decode(errors='replace'))
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path] + objs
  if verbose:
    print(' '.join(cmd), flush=True)
  subprocess.check_call(cmd)
  return lib_path


if __name__ == '__main__':
  print(build_library(force='--force' in sys.argv, tuning='--tuning' in sys.argv))
