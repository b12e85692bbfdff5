"""bench.py with the off-policy steps issued one launch per phase (trl_offpolicy_set_chain(0)): the A/B partner of the
default one-launch chain.  Same arguments as bench.py."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchrl_amd import _lib  # noqa: E402

_lib.lib().trl_offpolicy_set_chain(0)
sys.argv[0] = os.path.join(ROOT, 'bench.py')
runpy.run_path(sys.argv[0], run_name='__main__')
