"""One-off: hostile DATA (not parameters): non-uniform event grids, distances outside the table, masses outside the population,
zero / huge priors, zero draw probabilities, -100 rows, empty pixels -- HIP vs the NumPy oracle, all modes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
bad = 0
for it in range(n):
  pixelated = rng.random() < 0.8
  kind = str(rng.choice(['marginalized', 'marginalized', 'approximate', 'full'])) if pixelated else None
  E, S, P, Z = int(rng.integers(2, 6)), int(rng.integers(64, 500)), int(rng.integers(1, 6)), int(rng.integers(16, 80))
  cfg, ev, inj = H.small_config(E=E, S=S, P=P, Z=Z, I=int(rng.integers(300, 2000)), seed=int(rng.integers(1, 10**6)), ragged=bool(rng.random() < 0.5), pixelated=pixelated)
  ev = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in ev.items()}
  inj = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in inj.items()}
  what = []
  if rng.random() < 0.5:                       # non-uniform, still increasing event grids
    zg = ev['z_grids']
    w = rng.uniform(0.2, 1.8, size=zg.shape); w[:, 0] = 0.
    t = np.cumsum(w, axis=1); t /= t[:, -1:]
    ev['z_grids'] = zg[:, :1] + t * (zg[:, -1:] - zg[:, :1]); what.append('nonuniform-grid')
  if rng.random() < 0.4:
    i = rng.integers(0, E); ev['dL'][i, :7] = [1e-12, 1e-6, 5e2, 1e5, 0., 3e1, 2e-3]; what.append('dL-out-of-table')
  if rng.random() < 0.4:
    i = rng.integers(0, E); ev['m1det'][i, :20] *= 40.; ev['m2det'][i, 20:40] *= 0.01; what.append('masses-out-of-range')
  if rng.random() < 0.3:
    i = rng.integers(0, E); ev['pe_prior'][i, :3] = [0., 1e-300, 1e300]; what.append('prior-extremes')
  if rng.random() < 0.3:
    inj['p_draw'][:3] = [0., 1e-300, 1e300]; what.append('p_draw-extremes')
  if pixelated and rng.random() < 0.3:
    i = rng.integers(0, E); ev['p_cat'][i, 0, : Z // 2] = 0.; what.append('p_cat-zeros')
  if pixelated and rng.random() < 0.3:
    i = rng.integers(0, E); ev['pixels_pe_opt_nside'][i, :] = ev['pixels_opt_nsides'][i, 0]; what.append('all-samples-in-one-pixel')
  like_kw = {}
  if kind != 'full':
    like_kw = dict(cut_grid=[None, 2.0][int(rng.integers(0, 2))], binning=bool(rng.random() < 0.7), num_bins=int(rng.choice([5, 40, 200])))
  lam = dict(H0=float(rng.uniform(50., 100.)))
  try:
    like_o, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw)
    like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw)
    with np.errstate(all='ignore'):
      ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
    H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
    if np.isfinite(ro[2]):
      np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)
    like_p.close()
  except AssertionError as e:
    bad += 1
    print(f"MISMATCH it={it} kind={kind} shape=({E},{S},{P},{Z}) like_kw={like_kw} what={what}\n   {str(e)[:600]}", flush=True)
print('done;', bad, 'mismatches of', n)
