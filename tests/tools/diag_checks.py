"""Checks that need the DIAGNOSTIC build of the library (-DCHM_DIAG: other kernels for the same quantity, switched-off safeguards).  Run by
tests/test_zz_variant_builds.py in a process of its own with CHIMERA_LIB pointing at that build; prints one `ok <name>` line per check."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chimera_amd import _lib
from tests import helpers as H
from tests.test_gpu_parity import _decades_case, RTOL_L


def full_chain_against_general():
  """kind_p_gw3d='full': the sample-stationary kernel (k_full_kde_chain) against the general one (k_full_kde) on the same inputs."""
  for case, kw, like_kw in (('wide_kernels', {}, {}), ('narrow_kernels', {}, dict(bw_method=0.12)),
                            ('many_samples', dict(S=4500, E=2, P=2, Z=600), {})):
    k = dict(E=3, S=700, P=3, Z=900, I=1500, seed=41)
    k.update(kw)
    cfg, ev, inj = H.small_config(ragged=True, **k)
    like, _, _ = H.build_product(ev, inj, kind='full', like_kw=like_kw)
    pop = [like.population.update(H0=69.)]
    res = like._eval(pop, want=('log_like_evs',))
    assert like.full_general_pixels(1) == 0
    like.set_option('diag_full_chain', 0)
    ref = like._eval(pop, want=('log_like_evs',))
    assert like.full_general_pixels(1) == 0                  # (nothing flagged: the chain kernel did not run)
    H.assert_loglike_close(res['log_like_evs'][0], ref['log_like_evs'][0], rtol=1e-11, atol=1e-11)
    like.close()
  print('ok full_chain_against_general')


def dense_redo_is_what_saves_the_decades_case():
  cfg, ev, inj = _decades_case()
  like_o, _, _ = H.build_oracle(ev, inj)
  with np.errstate(all='ignore'):
    ro = like_o.compute_all(H0=70.)
  like, _, _ = H.build_product(ev, inj)
  H.assert_loglike_close(like.compute_all(H0=70.)[0], ro[0], rtol=RTOL_L, atol=1e-9)
  like.set_option('diag_no_dense_node', 1)
  rq = like.compute_all(H0=70.)
  assert np.max(np.abs(rq[0] - ro[0])) > 1e-3                 # the prefix differences alone lose these events
  like.set_option('fused', 2)                                  # ... in the fused event kernel as well
  rq = like.compute_all(H0=70.)
  assert np.max(np.abs(rq[0] - ro[0])) > 1e-3
  like.set_option('diag_no_dense_node', 0)
  H.assert_loglike_close(like.compute_all(H0=70.)[0], ro[0], rtol=RTOL_L, atol=1e-9)
  print('ok dense_redo_is_what_saves_the_decades_case')


def generic_kernels_against_the_fast_ones():
  """The general sample stage, selection kernel, marginalized kernel and whole-grid per-z factors against the production forms."""
  cfg, ev, inj = H.small_config(E=7, S=384, P=5, Z=72, I=3000, seed=5, ragged=True)
  like, _, _ = H.build_product(ev, inj)
  lams = [dict(H0=float(h)) for h in (61., 70., 77.5)]
  pops = [like.population.update(**l) for l in lams]
  base = like._eval(pops, want=('log_like_evs',))
  for name in ('diag_samples_generic', 'diag_selection_generic', 'diag_marg_generic', 'diag_zf_full', 'diag_no_grid_prep', 'diag_no_zf_sel'):
    like.set_option(name, 1)
    r = like._eval(pops, want=('log_like_evs',))
    like.set_option(name, 0)
    H.assert_loglike_close(r['log_like_evs'], base['log_like_evs'], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r['N_exp'], base['N_exp'], rtol=1e-11)
  print('ok generic_kernels_against_the_fast_ones')


if __name__ == '__main__':
  assert _lib.lib().chm_diag_build() == 1, 'not the diagnostic build'
  full_chain_against_general()
  dense_redo_is_what_saves_the_decades_case()
  generic_kernels_against_the_fast_ones()
