"""One-off: extreme hyper-parameter draws evaluated as ONE batch must equal the same draws evaluated one call at a time, bit for bit
(per-draw flags -- unsorted / NaN-tailed distance tables, poisoned grids -- must not leak between the draws of a call)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
bad = tot = 0
for kind, like_kw in [(None, {}), ('approximate', {}), ('full', {}), ('marginalized', {}), ('marginalized', dict(cut_grid=None, num_bins=31)),
                      ('marginalized', dict(binning=False))]:
  pixelated = kind is not None
  cfg, ev, inj = H.small_config(E=7, S=300, P=3, Z=50, I=3000, seed=int(rng.integers(1, 10**6)), ragged=True, pixelated=pixelated)
  for mass, cosmo in [('plp', 'mg_flrw'), ('bpl', 'flrw')]:
    like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw, models=dict(mass=mass, cosmo=cosmo))
    lams = []
    for _ in range(n):
      lam = dict(H0=rng.uniform(20., 200.), Om0=rng.uniform(0.02, 0.98), gamma=rng.uniform(-2., 8.), kappa=rng.uniform(0., 8.), zp=rng.uniform(0.2, 5.),
                 m_low=rng.uniform(1.5, 9.), m_high=rng.uniform(40., 200.), beta=rng.uniform(-3., 6.), w0=rng.uniform(-2., -0.3), wa=rng.uniform(-1., 1.),
                 Ok0=rng.uniform(-0.3, 0.3) if rng.random() < 0.5 else 0.)
      if cosmo == 'mg_flrw': lam.update(Xi0=rng.uniform(0.1, 8.), n=rng.uniform(0., 6.))
      if mass == 'plp': lam.update(alpha=rng.uniform(-1., 9.), lambda_peak=rng.uniform(0., 1.), mu_g=rng.uniform(10., 70.), sigma_g=rng.uniform(0.2, 15.), delta_m=rng.uniform(0.05, 15.))
      else: lam.update(alpha_1=rng.uniform(-2., 6.), alpha_2=rng.uniform(-1., 12.), break_fraction=rng.uniform(0.01, 0.99), delta_m=rng.uniform(0.05, 15.))
      lams.append({k: float(v) for k, v in lam.items()})
    with np.errstate(all='ignore'):
      b = like_p.batch(lams)
      s = np.array([like_p(**l) for l in lams])
    tot += n
    neq = ~((b == s) | (np.isnan(b) & np.isnan(s)))
    if neq.any():
      bad += int(neq.sum())
      i = int(np.argmax(neq))
      print(f"MISMATCH kind={kind} {like_kw} {mass}/{cosmo}: {int(neq.sum())} draws differ, e.g. draw {i}: batch {b[i]!r} single {s[i]!r} lam={lams[i]}", flush=True)
    print(kind, like_kw, mass, cosmo, 'finite', int(np.isfinite(b).sum()), '-inf', int(np.isneginf(b).sum()), 'nan', int(np.isnan(b).sum()), flush=True)
    like_p.close()
print('done;', bad, 'mismatches of', tot)
