import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope='session', autouse=True)
def _native_build():
  """Every session -- CPU or GPU -- compiles the HIP library and the oracle's C restatement from source before the first test
  (`__graft_entry__.build()`: hipcc --offload-arch=gfx950, in-tree; a no-op when the binaries are newer than every source).  A GPU
  box that received a stale or foreign libchimera_hip.so with the snapshot therefore never tests it."""
  import __graft_entry__ as entry
  # on a GPU box (a fresh snapshot: file times say nothing) the library is ALWAYS rebuilt from the sources that travelled with it
  entry.build(force=os.path.exists('/dev/kfd') and not os.environ.get('CHIMERA_NO_REBUILD'))
  yield
