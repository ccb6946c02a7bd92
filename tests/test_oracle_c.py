"""The plain-C restatement (oracle/chimera_oracle_c.c) against the NumPy one (oracle/chimera_oracle.py): two independent
restatements of the same reference lines must agree to rounding on the same inputs.  CPU only."""
import numpy as np
import pytest
from oracle import chimera_oracle as O
from oracle import oracle_c as OC
from tests import helpers as H

RT = 2e-11          # the two differ in summation order (np.sum is pairwise) and in libm's pow vs NumPy's loops


@pytest.mark.parametrize('models', [dict(), dict(mass='tpl'), dict(mass='bpl'), dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.6, n=1.9)),
                                    dict(rate='power_law'), dict(rate='trunc_madau_dickinson', rate_kw=dict(zmax=2.5)),
                                    dict(cosmo_kw=dict(Ok0=0.05)), dict(cosmo_kw=dict(w0=-0.9, wa=0.2))])
def test_tables_match_numpy_oracle(models):
  cfg, ev, inj = H.small_config(E=2, S=32, P=2, Z=16, I=50)
  like, pop, sel = H.build_oracle(ev, inj, models=models)
  t = OC.tables(pop)
  np.testing.assert_allclose(t['zt'], pop.cosmo.z_grid_interp, rtol=1e-13)
  np.testing.assert_allclose(t['It'], pop.cosmo.integral_invE_interp, rtol=1e-13)
  np.testing.assert_allclose(t['dLt'], O.dL_at_z(pop.cosmo, pop.cosmo.z_grid_interp), rtol=1e-13)
  np.testing.assert_allclose(t['m_grid'], pop.mass.m_grid, rtol=1e-13)
  np.testing.assert_allclose(t['cdf_m2'], pop.mass.cdf_m2_conditioned, rtol=1e-12, atol=1e-16 * pop.mass.cdf_m2_conditioned[-1])
  np.testing.assert_allclose(t['norm_p_m1'], pop.mass.norm_p_m1, rtol=1e-13)
  np.testing.assert_allclose(t['fR'], pop.gal_cat.completeness.fR(pop.cosmo), rtol=1e-12)


@pytest.mark.parametrize('models,like_kw,lam', [
  (dict(), dict(), dict(H0=67.)),
  (dict(mass='tpl'), dict(), dict(H0=74., alpha=2.2)),
  (dict(mass='bpl'), dict(num_bins=50), dict(H0=70.)),
  (dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.6, n=1.9)), dict(), dict(Xi0=1.3)),
  (dict(rate='power_law'), dict(bw_method='silverman'), dict(gamma=2.2)),
  (dict(), dict(binning=False), dict(H0=70.)),
  (dict(), dict(cut_grid=None, bw_method=0.3), dict(H0=72.)),
  (dict(), dict(pe_neff=1e9), dict(H0=70.)),               # every event fails the n_eff guard -> L_i = 0
  # masses whose grid end nodes differ in the last bit between NumPy's SIMD pow/log10 and the C library (oracle pins them to libm)
  (dict(mass='tpl'), dict(), dict(m_low=3.760945123513589, m_high=80.5068404471645)),
  (dict(mass='tpl'), dict(), dict(m_low=4.090710327355209, m_high=97.51325226101044)),
])
def test_marginalized_path_matches_numpy_oracle(models, like_kw, lam):
  cfg, ev, inj = H.small_config(E=5, S=192, P=4, Z=48, I=1500, seed=11, ragged=True)
  like, pop, sel = H.build_oracle(ev, inj, models=models, like_kw=like_kw)
  ro = like.compute_all(**lam)
  rc = OC.compute_all(like, lam, nthreads=2)
  H.assert_loglike_close(rc[0], ro[0], rtol=RT, atol=1e-11)
  popu = pop.update(**lam)
  np.testing.assert_allclose(OC.numlike_marg(like, popu, nthreads=1), like.compute_numlike_evs(popu), rtol=RT, atol=1e-300)
  np.testing.assert_allclose(OC.n_exp(sel, popu)[0], sel.N_exp(popu), rtol=1e-12)
  if np.isfinite(ro[3]):
    np.testing.assert_allclose(rc[3], ro[3], rtol=0, atol=1e-9)


def test_selection_guard_and_thread_count_do_not_change_the_result():
  cfg, ev, inj = H.small_config(E=3, S=64, P=2, Z=24, I=4000, seed=5)
  like, pop, sel = H.build_oracle(ev, inj)
  a = OC.n_exp(sel, pop, nthreads=1)
  b = OC.n_exp(sel, pop, nthreads=3)
  np.testing.assert_allclose(a, b, rtol=1e-13)
  sel_strict = O.selection_function(sel.theta_inj_det, sel.N_inj, N_eff=1e12)
  assert OC.n_exp(sel_strict, pop)[0] == 0.0 and sel_strict.N_exp(pop) == 0.0        # n_eff < N_eff -> 0
  sel_none = O.selection_function(sel.theta_inj_det, sel.N_inj, N_eff=None)
  np.testing.assert_allclose(OC.n_exp(sel_none, pop)[0], sel_none.N_exp(pop), rtol=1e-12)
  assert OC.max_threads() >= 1


@pytest.mark.parametrize('pixelated,like_kw,models,lam', [
  (False, dict(), dict(), dict(H0=67.)),
  (False, dict(kernel='gauss'), dict(mass='tpl'), dict(H0=74., alpha=2.2)),
  (False, dict(binning=False, cut_grid=None), dict(), dict(H0=70.)),
  (True, dict(), dict(), dict(H0=69.)),
  (True, dict(kernel='gauss', bw_method='silverman', num_bins=40), dict(mass='bpl', cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.4, n=1.7)), dict(H0=72.)),
  (True, dict(cut_grid=None, bw_method=0.3), dict(rate='power_law'), dict(gamma=2.0)),
])
def test_1d_and_approximate_modes_match_numpy_oracle(pixelated, like_kw, models, lam):
  cfg, ev, inj = H.small_config(E=5, S=160, P=3, Z=40, I=1200, seed=13, ragged=True, pixelated=pixelated)
  like, pop, sel = H.build_oracle(ev, inj, pixelated=pixelated, kind='approximate' if pixelated else None, models=models, like_kw=like_kw)
  ro = like.compute_all(**lam)
  rc = OC.compute_all(like, lam, nthreads=2)
  H.assert_loglike_close(rc[0], ro[0], rtol=RT, atol=1e-11)
  popu = pop.update(**lam)
  np.testing.assert_allclose(OC.numlike(like, popu, nthreads=1), like.compute_numlike_evs(popu), rtol=RT, atol=1e-300)
  if np.isfinite(ro[3]):
    np.testing.assert_allclose(rc[3], ro[3], rtol=0, atol=1e-9)


@pytest.mark.parametrize('like_kw,models,lam', [
  (dict(), dict(), dict(H0=68.)),
  (dict(bw_method='silverman', cut_grid=1.5), dict(mass='tpl'), dict(H0=73., alpha=2.3)),
  (dict(bw_method=0.4), dict(mass='bpl', cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.5, n=1.8)), dict(H0=71.)),
  (dict(pe_neff=1e9), dict(), dict(H0=70.)),               # every event skipped (`if n_effs[ev] < pe_neff: continue`) -> L_i = 0
])
def test_full_mode_matches_numpy_oracle(like_kw, models, lam):
  """kind_p_gw3d='full': the C restatement of likelihood.py:211-260 + utils/math.py:154-229 against the NumPy one."""
  cfg, ev, inj = H.small_config(E=4, S=160, P=3, Z=40, I=1200, seed=17, ragged=True)
  like, pop, sel = H.build_oracle(ev, inj, kind='full', models=models, like_kw=like_kw)
  ro = like.compute_all(**lam)
  rc = OC.compute_all(like, lam, nthreads=2)
  H.assert_loglike_close(rc[0], ro[0], rtol=RT, atol=1e-11)
  popu = pop.update(**lam)
  np.testing.assert_allclose(OC.numlike_full(like, popu, nthreads=1), like.compute_numlike_evs(popu), rtol=RT, atol=1e-300)
  if np.isfinite(ro[3]):
    np.testing.assert_allclose(rc[3], ro[3], rtol=0, atol=1e-9)


def test_random_hostile_configurations_c_against_numpy():
  """[r4] The generator of scripts/fuzz_parity.py (random shapes, modes, models, KDE options; 60 % of the configurations with one hostile input out of
  24 kinds, 30 % with a hyper-parameter at the edge of a prior box) with the C restatement in the product's place: the two restatements of the
  reference algorithm must agree on every one -- also on the inputs on which the round-4 GPU campaign found the product wrong (NaN primary mass with
  the truncated power law, negative priors, NaNs in the catalogue term, vanishing weight sums).  7 500 configurations were run that way in round 4
  (profiles/r04/fuzz_parity.txt); 80 of them here."""
  import os
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, os.path.join(root, 'scripts'))
  import fuzz_parity as F

  class CLike:
    def __init__(self, like_o): self.o = like_o
    def compute_all(self, **lam): return OC.compute_all(self.o, lam, nthreads=2)
    def __call__(self, **lam): return 0.                    # (the bit-for-bit checks of the generator concern the device: neutralised)
    def batch(self, lams): return np.zeros(len(lams))
    def set_option(self, *a): raise AssertionError('no fused kernel here')
    def close(self): pass
  cap = {}
  orig_o, orig_p, shares = H.build_oracle, H.build_product, (F.HOSTILE_SHARE, F.EXTREME_SHARE)

  def bo(ev, inj, **kw):
    r = orig_o(ev, inj, **kw)
    cap['o'] = r[0]
    return r
  H.build_oracle, H.build_product = bo, (lambda ev, inj, **kw: (CLike(cap['o']), None, CLike(cap['o'])))
  F.HOSTILE_SHARE, F.EXTREME_SHARE = 0.6, 0.3
  try:
    bad = []
    for i in range(80):
      ok, desc, _ = F.one(np.random.default_rng(77000 + 700000 + i))
      if not ok and 'no fused kernel here' not in desc:
        bad.append(desc[:800])
    assert not bad, '\n'.join(bad)
  finally:
    H.build_oracle, H.build_product = orig_o, orig_p
    F.HOSTILE_SHARE, F.EXTREME_SHARE = shares
