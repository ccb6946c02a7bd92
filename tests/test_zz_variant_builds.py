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



def test_probe_build_replays_the_production_bodies_and_leaves_the_results_alone():
  """[r6] -DCHM_PROBE: scripts/gw_loop_probe.hip / sample_body_probe.hip replay the production bodies of k_kde_marg_sub2 and k_samples_fast on a
  cache-resident workload (the measured ceilings behind bench.py's roofline.frac_of_sustained).  A short run: both bodies report a rate, and the
  evaluation made AFTER the probes (same handle, same workspaces) gives the value a run without probes gives."""
  import json
  import re
  # the probe kernels must BE the production code: same registers, nothing spilled to scratch (the first form of the probes looped the bodies inside
  # a block and carried 12-48 spilled vector registers through the replayed stream -- the ceilings it gave were 17-29 % too low)
  b = subprocess.run(['bash', os.path.join(ROOT, 'scripts', 'build_variant.sh'), 'probe', '-DCHM_PROBE', '-Rpass-analysis=kernel-resource-usage'], cwd=ROOT,
                     capture_output=True, text=True, timeout=900)
  assert b.returncode == 0, b.stderr[-2000:]
  lib = os.path.join(ROOT, 'chimera_amd', 'lib', 'variants', 'libchimera_hip_probe.so')
  res = {}
  for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?VGPRs Spill: (\d+)', b.stderr, re.S):
    res[m.group(1)] = tuple(int(x) for x in m.group(2, 3, 4))
  probe = {k: v for k, v in res.items() if 'k_probe_gw' in k or 'k_probe_samples' in k}
  assert len(probe) == 2, sorted(res)[:5]
  prod = {'gw': next(v for k, v in res.items() if 'k_kde_marg_sub2ILi32ELi4ELi200ELb0' in k), 'sf': next(v for k, v in res.items() if 'k_samples_fastILi2ELb0ELb0' in k)}
  for k, v in probe.items():
    assert v[1] == 0 and v[2] == 0, (k, v)
    assert v[0] <= prod['gw' if 'k_probe_gw' in k else 'sf'][0] + 2, (k, v, prod)
  env = dict(os.environ, CHIMERA_LIB=lib, CHIMERA_NO_REBUILD='1')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'run_probes.py'), '--events', '2', '--draws', '2', '--seconds', '0.3'], cwd=ROOT, env=env,
                     capture_output=True, text=True, timeout=600)
  assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
  out = json.loads(p.stdout.strip().split('\n')[-1])
  gw = next(v for k, v in out.items() if k.startswith('k_kde_marg_sub2'))
  sf = next(v for k, v in out.items() if k.startswith('k_samples_fast'))
  assert gw['units_per_s'] > 1e7 and sf['units_per_s'] > 1e9, (gw['units_per_s'], sf['units_per_s'])
  # the same workload on the release library, no probes
  code = ("import sys, numpy as np; sys.path.insert(0, %r); import chimera_amd as CH; from chimera_amd import synth; "
          "from chimera_amd.catalog import dVdz_completeness, pixelated_catalog; cfg, ev, inj = synth.make_config('C3', E=2, I=4000); "
          "pe = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside'); "
          "th = CH.data.theta_pe_det(**{k: ev[k] for k in pe}); gc = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels']); "
          "pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gc, scale_free=True); "
          "sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), N_inj=inj['N_inj'], N_eff=5.); "
          "like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', kernel='epan', bw_method=None, cut_grid=2, binning=True, num_bins=200); "
          "print(repr(float(like(H0=70.))))") % ROOT
  q = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=dict(os.environ, CHIMERA_NO_REBUILD='1'), capture_output=True, text=True, timeout=600)
  assert q.returncode == 0, q.stderr[-2000:]
  assert float(q.stdout.strip().split('\n')[-1]) == out['log_hyper_after'], (q.stdout, out['log_hyper_after'])
