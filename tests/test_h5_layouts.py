"""HDF5 branch of the loaders (CHIMERA/utils/io.py:7-66, data.py:61-216, 395-404, catalog.py:96-103): the reference's products are
HDF5 and this image's main interpreter has no h5py, so the branch is exercised by a second interpreter that has it
(/opt/conda/bin/python3.9 here).  The script writes each layout with h5py directly and reads it back through the package."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANDIDATES = [sys.executable, '/opt/conda/bin/python3.9', '/opt/conda/bin/python3', '/opt/conda/bin/python']


def _python_with_h5py():
  for exe in CANDIDATES:
    if os.path.exists(exe):
      try:
        if subprocess.call([exe, '-c', 'import h5py, numpy'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120) == 0:
          return exe
      except Exception:                                        # noqa: BLE001
        pass
  return None


def test_hdf5_layouts_round_trip_through_the_loaders(tmp_path):
  exe = _python_with_h5py()
  if exe is None:
    pytest.skip("no interpreter with h5py on this machine")
  r = subprocess.run([exe, os.path.join(ROOT, 'tests', 'tools', 'h5_layouts.py'), str(tmp_path)], capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stdout + r.stderr
  assert 'h5 layouts ok' in r.stdout
