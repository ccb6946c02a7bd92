"""BASELINE.json configurations at their FULL sizes on the GPU, every event against the plain-C restatement
(oracle/chimera_oracle_c.c, OpenMP; itself cross-checked against the NumPy oracle in tests/test_oracle_c.py):

  C3  1000 events x 32 pixels x 1000 z-bins x 4096 samples, PLP + Madau-Dickinson, flat-LCDM  (the headline configuration)
  C5  10 000 events x 32 pixels x 1000 z-bins x 4096 samples, modified GW propagation (Xi0, n), H0 varied as well
  a9  kind_p_gw3d='full' (3-D Gaussian KDE) at 50 events x 32 pixels x 1000 z-bins x 4096 samples

(C1, C2, C4 at full size: tests/test_gpu_parity.py::test_baseline_configurations_at_full_size_against_the_c_oracle.)
Stated fp64 tolerance: per-event log L_i rtol 1e-9 (+1e-9 abs), log N_exp rtol 1e-10, log_hyper atol 1e-7 sqrt(E).
"""
import gc
import os
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _nthreads():
  return min(16, os.cpu_count() or 1)


def _check(like_p, like_o, lams, E):
  from oracle import oracle_c as OC
  batch = like_p.batch(lams)
  for i, lam in enumerate(lams):
    rp = like_p.compute_all(**lam)
    rc = OC.compute_all(like_o, lam, nthreads=_nthreads())
    assert np.mean(np.isfinite(rc[0])) > 0.9                # some events have L_i = 0 for a draw (-1.797e308 class): compared by class
    H.assert_loglike_close(rp[0], rc[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rp[2], rc[2], rtol=1e-10)
    if np.isfinite(rc[3]):
      np.testing.assert_allclose(rp[3], rc[3], rtol=0, atol=1e-7 * np.sqrt(E))
    else:
      assert rp[3] == rc[3] or (rp[3] <= -1e300 and rc[3] <= -1e300)
    assert batch[i] == rp[3] or (np.isnan(batch[i]) and np.isnan(rp[3]))                                   # the batched call is the scalar call, bit for bit


@pytest.mark.timeout(900)
def test_c3_full_size_every_event_against_the_c_oracle():
  from chimera_amd import synth
  cfg, ev, inj = synth.make_config('C3')
  assert (cfg['E'], cfg['P'], cfg['Z'], cfg['S']) == (1000, 32, 1000, 4096)
  like_p, _, _ = H.build_product(ev, inj)
  like_o, _, _ = H.build_oracle(ev, inj)
  _check(like_p, like_o, [dict(H0=67.), dict(H0=88., lambda_peak=0.08, gamma=2.0), dict(H0=58., alpha=2.8, mu_g=31.)], cfg['E'])
  like_p.close()
  del like_p, like_o, ev, inj
  gc.collect()


@pytest.mark.timeout(1500)
def test_c5_full_size_every_event_against_the_c_oracle():
  """BASELINE.json configs[4]: 10 000 events, mg_flrw with (Xi0, n, H0) varied per draw, on ONE GPU (4.2 GB of inputs)."""
  from chimera_amd import synth
  cfg, ev, inj = synth.make_config('C5')
  assert (cfg['E'], cfg['P'], cfg['Z'], cfg['S']) == (10000, 32, 1000, 4096)
  models = dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.8, n=1.9))
  like_p, _, _ = H.build_product(ev, inj, models=models)
  like_o, _, _ = H.build_oracle(ev, inj, models=models)
  _check(like_p, like_o, [dict(H0=72., Xi0=2.4, n=1.5), dict(H0=61., Xi0=0.7, n=2.6)], cfg['E'])
  like_p.close()
  del like_p, like_o, ev, inj
  gc.collect()


@pytest.mark.timeout(900)
def test_full_mode_at_baseline_shape_against_the_c_oracle():
  """a9 (p_gw3dfull + numba_gkde_nd) at 50 events x 32 pixels x 1000 z-bins x 4096 samples: ~2e9 sample x query pairs per
  evaluation in the C restatement's plain double loop (one exp per pair) against k_full_kde's chunked recurrence."""
  from chimera_amd import synth
  from oracle import oracle_c as OC
  cfg, ev, inj = synth.make_config('C3', E=50, I=20_000)
  assert (cfg['P'], cfg['Z'], cfg['S']) == (32, 1000, 4096)
  like_p, _, _ = H.build_product(ev, inj, kind='full')
  like_o, _, _ = H.build_oracle(ev, inj, kind='full')
  for lam in (dict(H0=67.), dict(H0=84., alpha=3.0, gamma=2.2)):
    rp = like_p.compute_all(**lam)
    rc = OC.compute_all(like_o, lam, nthreads=_nthreads())
    assert np.all(np.isfinite(rc[0]))
    H.assert_loglike_close(rp[0], rc[0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rp[3], rc[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
  like_p.close()


@pytest.mark.timeout(600)
def test_streaming_and_plain_workspace_stores_give_the_same_bits():
  """[r6] A many-draw launch whose (z, w) workspaces exceed the 256 MB of the memory-side cache stores them with the streaming hint
  (LikeDev.zw_stream, k_samples_fast); smaller launches store plainly.  40 events x 4096 samples x 128 draws = 335 MB in one call against the same
  draws in two calls of 64 (168 MB each) and against scalar calls: every value bit for bit, and the C restatement within the stated tolerance."""
  from chimera_amd import synth
  from oracle import oracle_c as OC
  cfg, ev, inj = synth.make_config('C3', E=40, I=20_000)
  like_p, _, _ = H.build_product(ev, inj)
  like_p.set_option('groups', 1)                              # one event group: the whole call is one launch of the sample stage
  rng = np.random.default_rng(606)
  draws = dict(H0=rng.uniform(55., 90., 128), gamma=rng.uniform(1.5, 3.5, 128), lambda_peak=rng.uniform(0.01, 0.1, 128))
  assert 40 * 4096 * 128 * 16 > 256 << 20 > 40 * 4096 * 64 * 16
  whole = like_p.batch(draws)
  halves = np.concatenate([like_p.batch({k: v[:64] for k, v in draws.items()}), like_p.batch({k: v[64:] for k, v in draws.items()})])
  assert np.all(np.isfinite(whole))
  assert np.array_equal(whole, halves)
  for i in (0, 63, 64, 127):
    assert like_p(**{k: float(v[i]) for k, v in draws.items()}) == whole[i]
  like_o, _, _ = H.build_oracle(ev, inj)
  for i in (5, 100):
    rc = OC.compute_all(like_o, {k: float(v[i]) for k, v in draws.items()}, nthreads=_nthreads())
    np.testing.assert_allclose(whole[i], rc[3], rtol=0, atol=1e-7 * np.sqrt(cfg['E']))
  like_p.close()
  del like_p, like_o, ev, inj
  gc.collect()
