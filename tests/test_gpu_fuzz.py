"""GPU: a short run of scripts/fuzz_parity.py (the randomized parity generator with hostile inputs) + the regressions its long runs found in round 4."""
import os
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))


@pytest.mark.parametrize('share,seed0,many', [(0.15, 0, 0), (1.0, 20000, 0), (0.5, 600000, 4)])
def test_randomized_configurations_and_hostile_inputs(share, seed0, many):
  """150 random configurations per parameter set (shapes up to 4096 samples x 32 pixels x 400 grid points, every mode / model / KDE option; with
  `share` of them carrying one hostile input: NaN / inf / 0 / negative masses, distances or priors, distances beyond or below the table, events
  without weight, zero spread, hostile injections): compute_all against the NumPy oracle, scalar call == draws of a two- and a ten-draw batch to the bit, the
  fused event kernel against the separate kernels.  (Round 4 ran 30 000 of them: profiles/r04/fuzz_parity.txt.)"""
  import fuzz_parity as F
  F.HOSTILE_SHARE = share
  F.CHECK_PGW = True                                        # [r5] also the p_gw arrays of the API (p_gw3d / p_gw1d): NaN / 0 pattern and values against the oracle's
  bad = []
  for i in range(150 if not many else 60):                   # (many: every 4th configuration has 500+ small events -- the event-group path of ten-draw batches)
    ok, desc, _ = F.one(np.random.default_rng(77000 + seed0 + i), many_events=bool(many) and (seed0 + i) % many == many - 1)
    if not ok:
      bad.append(f"seed {seed0 + i}: {desc[:1200]}")
  assert not bad, '\n'.join(bad)


@pytest.mark.parametrize('seed,inf_share', [(6001464, 0.03), (6009213, 0.03), (7001164, 0.1)])
def test_findings_of_the_round_5_campaigns(seed, inf_share):
  """[r5] The three configurations the round's campaigns found (30 % hostile, 30 % extreme draws, every 40th with 500+ events; profiles/r05/fuzz_final_binary.txt):
  6001464 -- gamma = +inf in the 1-D mode with a Gaussian kernel and no cut_grid: p_gw1d vanishes nowhere, L_i = +inf (fixed: k_integrate_1d forms the
  reference's products over the whole grid for such a draw);  6009213 and 7001164 -- full mode, plp with lambda_peak = 1, sigma_g = 0.5: an event whose whole
  weight sits on one sample (1 - sum W^2 = 1e-10 / 4e-12), a covariance of rank one divided by ~0 -- the checker leaves such an event out, keyed on the
  conditioning number of the ORACLE's weights (never on the difference)."""
  import fuzz_parity as F
  keep = (F.HOSTILE_SHARE, F.EXTREME_SHARE, F.INF_RATE_SHARE, F.CHECK_PGW)
  F.HOSTILE_SHARE, F.EXTREME_SHARE, F.INF_RATE_SHARE, F.CHECK_PGW = 0.3, 0.3, inf_share, True
  try:
    ok, desc, checks = F.one(np.random.default_rng(77000 + seed), many_events=seed % 40 == 39)
  finally:
    F.HOSTILE_SHARE, F.EXTREME_SHARE, F.INF_RATE_SHARE, F.CHECK_PGW = keep
  assert ok, desc[:1500]
  assert 'oracle' in checks and 'p_gw' in checks


@pytest.mark.parametrize('mass', ['tpl', 'plp', 'bpl'])
@pytest.mark.parametrize('kind', ['marginalized', None])
def test_nan_primary_mass_by_mass_model(mass, kind):
  """[r4, found by the fuzz run] A NaN primary mass: the smoothing window of bpl / plp turns p_m1 NaN (the event's sums, L_i, log L_i follow); the
  truncated power law has no window and every factor is a masked 0 (mass.py:240-245, 334-341) -- the sample simply carries no weight.  The fast
  sample stage made it NaN for every model."""
  pix = kind is not None
  cfg, ev, inj = H.small_config(E=3, S=260, P=4, Z=64, I=1500, seed=5, pixelated=pix)
  ev = dict(ev); ev['m1det'] = ev['m1det'].copy(); ev['m1det'][1, 17] = np.nan
  lo, _, _ = H.build_oracle(ev, inj, pixelated=pix, kind=kind, models=dict(mass=mass))
  lp, _, sp = H.build_product(ev, inj, pixelated=pix, kind=kind, models=dict(mass=mass))
  with np.errstate(all='ignore'):
    ro, rp = lo.compute_all(H0=70.), lp.compute_all(H0=70.)
  assert bool(H.neginf_class(ro[0][1])) == (mass != 'tpl')
  H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
  if mass == 'tpl':
    np.testing.assert_allclose(rp[3], ro[3], rtol=1e-12, atol=1e-7)
  lp.close(); sp.close()


@pytest.mark.parametrize('like_kw', [dict(), dict(cut_grid=None, num_bins=64), dict(binning=False)])
@pytest.mark.parametrize('kind', ['marginalized', 'approximate'])
def test_negative_pe_prior_takes_the_dense_sums(kind, like_kw):
  """[r4, found by the fuzz run] A negative pe_prior gives a negative sample weight, which the reference's arithmetic takes as it comes.  The
  prefix-sum forms of the binned Epanechnikov KDE (standard GW kernel, general kernel, 1-D kernel) clamp at 0 and bound their rounding on the
  assumption of weights >= 0: a handle that saw a negative prior at upload uses the dense sums (LikeDev.neg_w) -- 1e-3 off before."""
  cfg, ev, inj = H.small_config(E=4, S=1024, P=8, Z=132, I=1500, seed=8)
  ev = dict(ev); ev['pe_prior'] = np.array(ev['pe_prior'], dtype=np.float64, copy=True)
  ev['pe_prior'][2, 100:140] = -ev['pe_prior'][2, 100:140]
  lo, _, _ = H.build_oracle(ev, inj, kind=kind, like_kw=like_kw)
  lp, _, sp = H.build_product(ev, inj, kind=kind, like_kw=like_kw)
  with np.errstate(all='ignore'):
    ro, rp = lo.compute_all(H0=68.), lp.compute_all(H0=68.)
  H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
  lp.close(); sp.close()


@pytest.mark.parametrize('kind', ['marginalized', 'approximate', 'full'])
@pytest.mark.parametrize('field', ['p_cat', 'z_grids'])
def test_nan_in_the_catalogue_term_or_the_grid_of_an_event(kind, field):
  """[r4, found by the fuzz run] A NaN in p_cat (a live pixel) or in the z grid of an event: the reference's integrand is 0 * NaN = NaN on that grid
  point also where p_gw is zero, trapz sums it, L_i = NaN -> log L_i = -inf for every draw.  The kernels skip the grid points outside the KDE's
  support (a NaN out there was not seen); the flag is formed at upload (chm_like::d_ev_bad) and applied where the per-event values are reduced."""
  cfg, ev, inj = H.small_config(E=4, S=300, P=5, Z=60, I=1500, seed=21)
  ev = dict(ev)
  ev[field] = np.array(ev[field], dtype=np.float64, copy=True)
  if field == 'p_cat':
    ev['p_cat'][2, 0, 0] = np.nan; ev['p_cat'][2, 0, -1] = np.nan            # the two ends of the grid: outside the KDE's support
  else:
    ev['z_grids'][2, -1] = np.nan
  lo, _, _ = H.build_oracle(ev, inj, kind=kind)
  lp, _, sp = H.build_product(ev, inj, kind=kind)
  with np.errstate(all='ignore'):
    ro, rp = lo.compute_all(H0=70.), lp.compute_all(H0=70.)
    assert np.isnan(lo.compute_numlike_evs(lo.population.update(H0=70.))[2]) and np.isnan(lp.compute_numlike_evs(lp.population.update(H0=70.))[2])
    batch = lp.batch([dict(H0=70.), dict(H0=64.)])
  assert ro[0][2] == -np.inf and rp[0][2] == -np.inf
  H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
  assert np.all(np.isneginf(batch)) and ro[3] == -np.inf and lp(H0=70.) == -np.inf
  lp.close(); sp.close()


def test_full_mode_with_vanishing_weight_sums():
  """[r4, found by the fuzz run] lambda_peak = 1, sigma_g = 0.5: the mass model is a 0.5 M_sun wide Gaussian, every sample of an event lies tens of
  widths away, sum(w) ~ 1e-200.  The reference's n_eff = sum(w)^2 / sum(w^2) is 0 / 0 = NaN there, `NaN < pe_neff` is False and the 3-D KDE goes
  ahead with weights it normalises first (log L_i ~ -470 ... -700).  The device formed sum(W^2) as sum(w^2) / sum(w)^2 from the un-normalised partial
  sums -- both underflow -- and returned NaN; k_full_prep / k_full_kde now sum (w / sum w)^2 over the samples below sum(w) = 1e-140."""
  import fuzz_parity as F
  F.HOSTILE_SHARE, F.EXTREME_SHARE = 0.6, 0.3
  try:
    for seed in (402278, 402329, 403751):
      ok, desc, _ = F.one(np.random.default_rng(77000 + seed))
      assert 'lambda_peak\': 1.0' in desc and 'kind=full' in desc
      assert ok, desc[:1500]
  finally:
    F.HOSTILE_SHARE, F.EXTREME_SHARE = 0.15, 0.2
