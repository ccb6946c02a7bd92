"""One-off: hyper-parameters drawn from wide / extreme ranges (flat priors far beyond the sensible ones), HIP vs the C oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H
from oracle import oracle_c as OC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
cfg, ev, inj = H.small_config(E=16, S=512, P=5, Z=120, I=8000, seed=77, ragged=True)
bad = 0
nfinite = 0
for mass, cosmo in [('plp', 'flrw'), ('bpl', 'flrw'), ('tpl', 'mg_flrw'), ('plp', 'mg_flrw')]:
  like_p, _, _ = H.build_product(ev, inj, models=dict(mass=mass, cosmo=cosmo))
  like_o, _, _ = H.build_oracle(ev, inj, models=dict(mass=mass, cosmo=cosmo))
  lams = []
  for _ in range(n):
    lam = dict(H0=rng.uniform(20., 200.), Om0=rng.uniform(0.02, 0.98), gamma=rng.uniform(-2., 8.), kappa=rng.uniform(0., 8.), zp=rng.uniform(0.2, 5.),
               m_low=rng.uniform(1.5, 9.), m_high=rng.uniform(40., 200.), beta=rng.uniform(-3., 6.))
    if rng.random() < 0.3:
      lam.update(w0=rng.uniform(-2., -0.3), wa=rng.uniform(-1., 1.))
    if rng.random() < 0.3:
      lam.update(Ok0=rng.uniform(-0.3, 0.3))
    if cosmo == 'mg_flrw':
      lam.update(Xi0=rng.uniform(0.1, 8.), n=rng.uniform(0., 6.))
    if mass == 'plp':
      lam.update(alpha=rng.uniform(-1., 9.), lambda_peak=rng.choice([0., 1., rng.uniform(0., 1.)]), mu_g=rng.uniform(10., 70.), sigma_g=rng.uniform(0.2, 15.),
                 delta_m=rng.choice([0.01, rng.uniform(0.05, 15.)]))
    elif mass == 'bpl':
      lam.update(alpha_1=rng.uniform(-2., 6.), alpha_2=rng.uniform(-1., 12.), break_fraction=rng.uniform(0.01, 0.99), delta_m=rng.uniform(0.05, 15.))
    else:
      lam.update(alpha=rng.uniform(-1., 9.))
    lams.append({k: float(v) for k, v in lam.items()})
  got = like_p.batch(lams)
  for i, lam in enumerate(lams):
    with np.errstate(all='ignore'):
      rc = OC.compute_all(like_o, lam, nthreads=8)
    g, r = got[i], rc[3]
    nfinite += int(np.isfinite(r) and r > -1e300)
    ok = (abs(g - r) <= 1e-7 * 4 + 1e-9 * abs(r)) if (np.isfinite(r) and np.isfinite(g)) else ((np.isnan(g) and np.isnan(r)) or g == r or (g < -1e300 and r < -1e300))
    if not ok:
      bad += 1
      with np.errstate(all='ignore'):
        ev_p = like_p.compute_all(**lam)[0]
      d = ev_p - rc[0]
      print(f"MISMATCH {mass}/{cosmo}: hip {g!r} c {r!r}  max|dlogL_i| {np.nanmax(np.abs(np.where(np.isfinite(d), d, 0.))):.3e}  lam={lam}", flush=True)
print('done;', bad, 'mismatches of', 4 * n, '(', nfinite, 'with a finite log-hyperlikelihood )')
