"""Build libtrl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m torchrl_amd._build            # rebuild if sources are newer than the library
    python -m torchrl_amd._build --force

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
SOURCES = ['trl_host.cpp', 'trl_replay.hip', 'trl_gae.hip', 'trl_loss.hip', 'trl_mlp.hip', 'trl_learn.hip',
           'trl_offpolicy.hip', 'trl_linear_u8.hip']
HEADERS = [os.path.join(CSRC, 'trl_common.h'), os.path.join(ROOT, 'include', 'trl_hip.h')]

# -ffp-contract=off: the parity bar is the reference's op order, one f32 rounding per op.
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off',
         '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def _hipcc():
  for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found: cannot build libtrl_hip.so')


def is_stale():
  if not os.path.exists(LIB_PATH):
    return True
  t = os.path.getmtime(LIB_PATH)
  deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
  return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
  if not force and not is_stale():
    return LIB_PATH
  os.makedirs(LIB_DIR, exist_ok=True)
  hipcc = _hipcc()
  objs = []
  procs = []
  for s in SOURCES:
    obj = os.path.join(LIB_DIR, os.path.splitext(s)[0] + '.o')
    if s.endswith('.cpp'):   # host-only translation unit (no device pass: it uses target_clones)
      cmd = [shutil.which('g++') or 'g++', '-O3', '-std=c++17', '-fPIC', '-Wall', '-I',
             os.path.join(ROOT, 'include'), '-c', os.path.join(CSRC, s), '-o', obj]
    else:
      extra = ['-DTRL_LEARN_STAMP'] if os.environ.get('TRL_LEARN_STAMP') == '1' else []   # diagnostic build
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
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
  if verbose:
    print(' '.join(cmd), flush=True)
  subprocess.check_call(cmd)
  return LIB_PATH


if __name__ == '__main__':
  print(build_library(force='--force' in sys.argv))
