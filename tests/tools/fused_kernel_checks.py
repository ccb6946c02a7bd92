"""[r5: NOT collected by the suite -- the fused kernel left the release library; tests/test_zz_variant_builds.py builds the -DCHM_WITH_FUSED variant
and runs this file in a process of its own with CHIMERA_LIB pointing at it.]
GPU tests of the fused event kernel (chimera_amd/csrc/chm_fused.h, CHM_OPT_FUSED): samples -> statistics -> per-pixel histograms -> per-z
factors -> KDE + integrand of one (event, draw) in one block, z and w never leaving the CU.  It is off by default (slower than the separate
kernels, profiles/r04/ab_fused_event_kernel.txt), so these tests switch it on: against the oracle to the stated tolerance (per-event log L_i
rtol 1e-9, log_hyper atol 1e-7 sqrt(E)), against the separate kernels to rounding (the sums that carry no order in the reference -- a bin's
weights, the event's sum w -- are formed in another order), run to run bit for bit (the fused histograms are summed in one fixed order), and on
the inputs that take its exact route (non-finite / non-positive distances, draws whose distance table is not sorted)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

RTOL_L = 1e-9


def _both(like, lams):
  out = {}
  for mode in (0, 2):
    like.set_option('fused', mode)
    out[mode] = like._eval(like._params_array(lams), want=('log_like_evs',))
  like.set_option('fused', 0)
  return out[0], out[2]


@pytest.mark.parametrize('name,kw,models,like_kw', [
  ('ragged', dict(E=6, S=256, P=4, Z=64, seed=7), None, None),
  ('odd number of pixels', dict(E=5, S=384, P=5, Z=48, seed=3), None, None),
  ('one pixel', dict(E=3, S=128, P=1, Z=32, seed=5), None, None),
  ('fewer samples than a tile', dict(E=4, S=100, P=3, Z=40, seed=11), None, None),
  ('32 pixels', dict(E=4, S=2048, P=32, Z=200, seed=2, ragged=False), None, None),
  ('segments longer than a wave range', dict(E=3, S=6000, P=16, Z=100, seed=9), None, None),
  ('bpl', dict(E=6, S=512, P=6, Z=64, seed=1), dict(mass='bpl'), None),
  ('tpl + mg_flrw', dict(E=6, S=512, P=6, Z=64, seed=4), dict(mass='tpl', cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.6, n=1.5)), None),
  ('silverman', dict(E=5, S=512, P=8, Z=64, seed=8), None, dict(bw_method='silverman')),
  ('scalar bandwidth, 4 sigma cut', dict(E=5, S=512, P=4, Z=96, seed=6), None, dict(bw_method=0.3, cut_grid=4.)),
])
def test_fused_kernel_against_the_oracle_and_the_separate_kernels(name, kw, models, like_kw):
  kw = dict(dict(I=1500, ragged=True), **kw)
  cfg, ev, inj = H.small_config(**kw)
  like_o, _, _ = H.build_oracle(ev, inj, models=models, like_kw=like_kw)
  like_p, _, _ = H.build_product(ev, inj, models=models, like_kw=like_kw)
  lams = [dict(H0=h) for h in (60., 70., 85.)]
  sep, fus = _both(like_p, lams)
  ref = np.array([like_o.compute_all(**l)[0] for l in lams])
  refh = np.array([like_o.compute_all(**l)[3] for l in lams])
  H.assert_loglike_close(fus['log_like_evs'], ref, rtol=RTOL_L, atol=1e-9)
  np.testing.assert_allclose(fus['log_hyper'], refh, rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
  H.assert_loglike_close(fus['log_like_evs'], sep['log_like_evs'], rtol=0, atol=1e-11)         # rounding of re-ordered sums only
  # the scalar call takes the few-draw instantiation (non-temporal streaming): the same bits as the batched one; and run to run the same bits
  like_p.set_option('fused', 1)
  sc = np.array([like_p(**l) for l in lams])
  np.testing.assert_array_equal(sc, fus['log_hyper'])
  like_p.set_option('fused', 2)
  again = like_p._eval(like_p._params_array(lams), want=('log_like_evs',))
  np.testing.assert_array_equal(again['log_like_evs'], fus['log_like_evs'])
  like_p.close()


def test_fused_kernel_takes_the_exact_route_for_hostile_distances_and_unsorted_tables():
  """Events with a NaN, zero or negative distance cannot take min z = z(min dL): the block writes z to the workspace first and reduces
  min / max from it (NaN-propagating as jnp.min / jnp.max); so does every event of a draw whose distance table is not monotonic (an unphysical
  closed universe: the reference's scan search on the unsorted table, NaN tails)."""
  cfg, ev, inj = H.small_config(E=8, S=384, P=4, Z=64, I=3000, seed=13, ragged=True)
  ev2 = dict(ev)
  dL = ev['dL'].copy(); m1 = ev['m1det'].copy()
  dL[0, 5] = np.nan; dL[1, 17] = 0.; dL[2, 3] = -1.5; dL[3, 100] = np.inf
  m1[4, 7] = np.nan; m1[5, :40] = 1e4
  ev2['dL'], ev2['m1det'] = dL, m1
  like_o, _, _ = H.build_oracle(ev2, inj)
  like_p, _, _ = H.build_product(ev2, inj)
  like_p.set_option('fused', 2)
  for lam in (dict(H0=70.), dict(H0=61., alpha=2.9)):
    with np.errstate(all='ignore'):
      ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
    H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
  like_p.close()
  lam = {'H0': 20.475122477377834, 'Om0': 0.04178692963304655, 'Ok0': -0.27540350807880004, 'Xi0': 0.28514090098152856,
         'n': 2.349145907232978, 'gamma': 1.54, 'kappa': 4.18, 'zp': 4.63, 'm_low': 6.16, 'm_high': 163.4, 'beta': -2.9,
         'alpha': 2.86, 'lambda_peak': 0.91, 'mu_g': 42.2, 'sigma_g': 6.9, 'delta_m': 6.7}
  cfg, ev, inj = H.small_config(E=16, S=512, P=5, Z=120, I=8000, seed=77, ragged=True)
  models = dict(mass='plp', cosmo='mg_flrw')
  like_p, _, _ = H.build_product(ev, inj, models=models)
  like_o, _, _ = H.build_oracle(ev, inj, models=models)
  like_p.set_option('fused', 2)
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
  assert np.any(np.isneginf(ro[0])) and np.any(np.isfinite(ro[0]))
  H.assert_loglike_close(rp[0], ro[0], rtol=RTOL_L, atol=1e-9)
  # a batch that mixes the unphysical draw with ordinary ones: the route is chosen per (event, draw)
  lams = [dict(H0=70.), lam, dict(H0=64.)]
  with np.errstate(all='ignore'):
    got = like_p._eval([like_p.population.update(**l) for l in lams], want=('log_like_evs',))['log_like_evs']
    ref = np.array([like_o.compute_all(**l)[0] for l in lams])
  H.assert_loglike_close(got, ref, rtol=RTOL_L, atol=1e-9)
  like_p.close()


def test_fused_kernel_with_events_that_fail_the_n_eff_guard_and_padded_pixels():
  """pe_neff above every event's effective sample size: every L_i = 0 -> -1.797e308 (likelihood.py:199); ragged pixel counts: the
  padded pixels contribute exact zeros."""
  cfg, ev, inj = H.small_config(E=5, S=256, P=6, Z=48, I=2000, seed=21, ragged=True)
  like_o, _, _ = H.build_oracle(ev, inj, like_kw=dict(pe_neff=1e9))
  like_p, _, _ = H.build_product(ev, inj, like_kw=dict(pe_neff=1e9))
  like_p.set_option('fused', 2)
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(H0=70.), like_p.compute_all(H0=70.)
  assert np.all(rp[0] == -np.finfo(np.float64).max) and np.all(ro[0] == rp[0])
  like_p.close()


@pytest.mark.timeout(900)
def test_fused_kernel_at_the_headline_shape_against_the_c_oracle():
  """C3's shape (32 pixels x 1000 z-bins x 4096 samples per event) on 150 of its events: every event of three draws against the plain-C
  restatement, batched (128-draw instantiation) and as scalar calls (few-draw instantiation)."""
  import os
  from chimera_amd import synth
  from oracle import oracle_c as OC
  cfg, ev, inj = synth.make_config('C3', E=150, I=20_000)
  assert (cfg['P'], cfg['Z'], cfg['S']) == (32, 1000, 4096)
  like_p, _, _ = H.build_product(ev, inj)
  like_o, _, _ = H.build_oracle(ev, inj)
  lams = [dict(H0=67.), dict(H0=88., lambda_peak=0.08, gamma=2.0), dict(H0=58., alpha=2.8, mu_g=31.)]
  like_p.set_option('fused', 2)
  many = like_p._eval([like_p.population.update(**l) for l in lams] * 4, want=('log_like_evs',))      # 12 draws: the many-draw instantiation
  like_p.set_option('fused', 1)
  for i, lam in enumerate(lams):
    rp = like_p.compute_all(**lam)
    rc = OC.compute_all(like_o, lam, nthreads=min(16, os.cpu_count() or 1))
    H.assert_loglike_close(rp[0], rc[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rp[3], rc[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
    np.testing.assert_array_equal(many['log_like_evs'][i], rp[0])
    np.testing.assert_array_equal(many['log_like_evs'][i + 6], rp[0])
  like_p.close()


def test_fused_kernel_event_beyond_the_distance_table():
  """[r4, found by the fuzz run] An event whose every distance lies beyond the last node of the draw's dL table (z clamps to z_max): the fused event
  kernel's per-event slice of the node records began behind the record the clamp reads."""
  cfg, ev, inj = H.small_config(E=3, S=1024, P=21, Z=58, I=1500, seed=12)
  ev = dict(ev); ev['dL'] = ev['dL'].copy(); ev['dL'][0] *= 40.
  lo, _, _ = H.build_oracle(ev, inj, like_kw=dict(cut_grid=2.0))
  lp, _, sp = H.build_product(ev, inj, like_kw=dict(cut_grid=2.0))
  with np.errstate(all='ignore'):
    ro, rp = lo.compute_all(H0=60.2, Om0=0.22), lp.compute_all(H0=60.2, Om0=0.22)
    lp.set_option('fused', 2)
    rf = lp.compute_all(H0=60.2, Om0=0.22)
  assert H.neginf_class(ro[0][0])
  H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
  H.assert_loglike_close(rf[0], rp[0], rtol=1e-12, atol=1e-12)
  lp.close(); sp.close()
