"""One-off: extreme hyper-parameters in the 1d / approximate / full modes (and marginalized without binning / cut), HIP vs NumPy oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bad = tot = 0
for kind, like_kw in [(None, {}), (None, dict(kernel='gauss', binning=False)), ('approximate', {}), ('approximate', dict(kernel='gauss', cut_grid=None)),
                      ('full', {}), ('marginalized', dict(binning=False)), ('marginalized', dict(cut_grid=None, num_bins=31))]:
  pixelated = kind is not None
  cfg, ev, inj = H.small_config(E=6, S=300, P=3, Z=50, I=3000, seed=int(rng.integers(1, 10**6)), ragged=True, pixelated=pixelated)
  for mass, cosmo in [('plp', 'mg_flrw'), ('bpl', 'flrw'), ('tpl', 'flrw')]:
    like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw, models=dict(mass=mass, cosmo=cosmo))
    like_o, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw, models=dict(mass=mass, cosmo=cosmo))
    for _ in range(n):
      lam = dict(H0=rng.uniform(20., 200.), Om0=rng.uniform(0.02, 0.98), gamma=rng.uniform(-2., 8.), kappa=rng.uniform(0., 8.), zp=rng.uniform(0.2, 5.),
                 m_low=rng.uniform(1.5, 9.), m_high=rng.uniform(40., 200.), beta=rng.uniform(-3., 6.))
      if rng.random() < 0.3: lam.update(w0=rng.uniform(-2., -0.3), wa=rng.uniform(-1., 1.))
      if rng.random() < 0.3: lam.update(Ok0=rng.uniform(-0.3, 0.3))
      if cosmo == 'mg_flrw': lam.update(Xi0=rng.uniform(0.1, 8.), n=rng.uniform(0., 6.))
      if mass == 'plp': lam.update(alpha=rng.uniform(-1., 9.), lambda_peak=rng.uniform(0., 1.), mu_g=rng.uniform(10., 70.), sigma_g=rng.uniform(0.2, 15.), delta_m=rng.uniform(0.05, 15.))
      elif mass == 'bpl': lam.update(alpha_1=rng.uniform(-2., 6.), alpha_2=rng.uniform(-1., 12.), break_fraction=rng.uniform(0.01, 0.99), delta_m=rng.uniform(0.05, 15.))
      else: lam.update(alpha=rng.uniform(-1., 9.))
      lam = {k: float(v) for k, v in lam.items()}
      tot += 1
      try:
        with np.errstate(all='ignore'):
          ro = like_o.compute_all(**lam)
      except np.linalg.LinAlgError:            # the oracle's Cholesky raises where the device returns NaN (degenerate covariance)
        continue
      with np.errstate(all='ignore'):
        rp = like_p.compute_all(**lam)
      # events whose whole integrand sits below the rounding noise of the prefix-sum KDE (1e-13 of its peak): the oracle has an exact
      # zero (-1.8e308), the device a value of order exp(-40) -- a documented limit, counted separately
      noise = (ro[0] < -1e300) & (rp[0] > -1e300) & (rp[0] < -30.)
      if noise.any():
        nnoise = globals().get('nnoise', 0) + int(noise.sum()); globals()['nnoise'] = nnoise
        rp = list(rp); rp[0] = np.where(noise, ro[0], rp[0])
      try:
        H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
        np.testing.assert_array_equal(np.isneginf(rp[0]), np.isneginf(ro[0]))
        if np.isfinite(ro[2]): np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)
      except AssertionError as e:
        bad += 1
        print(f"MISMATCH kind={kind} like_kw={like_kw} {mass}/{cosmo} lam={lam}\n   {str(e)[:500]}", flush=True)
    like_p.close()
print('done;', bad, 'mismatches of', tot, '; noise-level events:', globals().get('nnoise', 0))
