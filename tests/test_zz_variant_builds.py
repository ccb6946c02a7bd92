"""GPU tests that compile VARIANTS of the library with hipcc and run them in processes of their own.  The file sorts last on purpose: under
`pytest -x` a toolchain hiccup on the GPU box must not sit in front of a parity test (every parity test of the release library has run by the
time these start)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(name, flags):
  subprocess.check_call(['bash', os.path.join(ROOT, 'scripts', 'build_variant.sh'), name] + flags, cwd=ROOT)
  return os.path.join(ROOT, 'chimera_amd', 'lib', 'variants', f'libchimera_hip_{name}.so')


def test_diagnostic_build_compares_the_code_paths_the_release_library_does_not_expose():
  """-DCHM_DIAG: the general kernels against the production ones, the two full-mode kernels against each other, and the decades case WITHOUT
  the dense redo (wrong by > 1e-3: the case does exercise the limit of the prefix-sum form) -- tests/tools/diag_checks.py."""
  lib = _build('diag', ['-DCHM_DIAG', '-DCHM_WITH_FUSED'])      # (with the fused event kernel: diag_checks.py compares it too)
  env = dict(os.environ, CHIMERA_LIB=lib, CHIMERA_NO_REBUILD='1')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'diag_checks.py')], cwd=ROOT, env=env, capture_output=True, text=True,
                     timeout=900)
  assert p.returncode == 0, p.stdout + p.stderr
  for name in ('full_chain_against_general', 'dense_redo_is_what_saves_the_decades_case', 'generic_kernels_against_the_fast_ones'):
    assert f'ok {name}' in p.stdout, p.stdout + p.stderr


def test_k_tables_runs_with_a_private_segment_at_1024_threads():
  """[r3] Round 2 held k_tables to zero scratch after builds that spilled two registers at 1024 threads per block died with
  HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  Builds of the kernel that DO use scratch (64 B per lane, forced) -- the short-table variant
  at 1024 threads beside 33 KB static + dynamic LDS, the long-table variant at 512 and at 1024 threads -- run and reproduce the default
  build's tables and likelihoods (scripts/probe_tables_scratch.py, each variant in its own process): the private segment at that block
  size is not the cause."""
  for name, flags in (('ts512', ['-DCHM_TABLES_FORCE_SCRATCH=6']), ('ts1024', ['-DCHM_TABLES_LONG_NT=1024', '-DCHM_TABLES_FORCE_SCRATCH=6'])):
    _build(name, flags)
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'probe_tables_scratch.py')], cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert p.returncode == 0, p.stdout + p.stderr
  assert 'ts512: rc 0' in p.stdout and 'ts1024: rc 0' in p.stdout, p.stdout


def test_fused_event_kernel_variant_build():
  """[r5] The fused per-(event, draw) kernel (chm_fused.h) left the release library -- it is parity-green and moves 5.8x fewer bytes, but is slower
  than the separate kernels at every call size measured (profiles/r04/ab_fused_event_kernel.txt).  It stays buildable and tested: the
  -DCHM_WITH_FUSED variant runs the round-4 tests of the kernel (tests/tools/fused_kernel_checks.py: against the oracle, against the separate kernels,
  bit-reproducible, the exact route for hostile distances / unsorted tables) in a process of its own."""
  lib = _build('fused', ['-DCHM_WITH_FUSED'])
  env = dict(os.environ, CHIMERA_LIB=lib, CHIMERA_NO_REBUILD='1')
  p = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'tools', 'fused_kernel_checks.py'), '-q', '-m', 'gpu', '-x', '-p', 'no:cacheprovider'],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
  assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1500:]
  assert ' passed' in p.stdout and 'failed' not in p.stdout, p.stdout[-1500:]

