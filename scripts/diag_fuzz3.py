import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers as H
from oracle import chimera_oracle as O
# replay scripts/fuzz_data.py up to iteration 9
rng = np.random.default_rng(31337)
for it in range(10):
  pixelated = rng.random() < 0.8
  kind = str(rng.choice(['marginalized', 'marginalized', 'approximate', 'full'])) if pixelated else None
  E, S, P, Z = int(rng.integers(2, 6)), int(rng.integers(64, 500)), int(rng.integers(1, 6)), int(rng.integers(16, 80))
  cfg, ev, inj = H.small_config(E=E, S=S, P=P, Z=Z, I=int(rng.integers(300, 2000)), seed=int(rng.integers(1, 10**6)), ragged=bool(rng.random() < 0.5), pixelated=pixelated)
  ev = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in ev.items()}
  inj = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in inj.items()}
  what = []
  if rng.random() < 0.5:
    zg = ev['z_grids']
    w = rng.uniform(0.2, 1.8, size=zg.shape); w[:, 0] = 0.
    t = np.cumsum(w, axis=1); t /= t[:, -1:]
    ev['z_grids'] = zg[:, :1] + t * (zg[:, -1:] - zg[:, :1]); what.append('nonuniform-grid')
  if rng.random() < 0.4:
    i = rng.integers(0, E); ev['dL'][i, :7] = [1e-12, 1e-6, 5e2, 1e5, 0., 3e1, 2e-3]; what.append(('dL-out-of-table', int(i)))
  if rng.random() < 0.4:
    i = rng.integers(0, E); ev['m1det'][i, :20] *= 40.; ev['m2det'][i, 20:40] *= 0.01; what.append(('masses', int(i)))
  if rng.random() < 0.3:
    i = rng.integers(0, E); ev['pe_prior'][i, :3] = [0., 1e-300, 1e300]; what.append(('prior', int(i)))
  if rng.random() < 0.3:
    inj['p_draw'][:3] = [0., 1e-300, 1e300]; what.append('p_draw-extremes')
  if pixelated and rng.random() < 0.3:
    i = rng.integers(0, E); ev['p_cat'][i, 0, : Z // 2] = 0.; what.append(('pcat0', int(i)))
  if pixelated and rng.random() < 0.3:
    i = rng.integers(0, E); ev['pixels_pe_opt_nside'][i, :] = ev['pixels_opt_nsides'][i, 0]; what.append(('onepix', int(i)))
  like_kw = {}
  if kind != 'full':
    like_kw = dict(cut_grid=[None, 2.0][int(rng.integers(0, 2))], binning=bool(rng.random() < 0.7), num_bins=int(rng.choice([5, 40, 200])))
  lam = dict(H0=float(rng.uniform(50., 100.)))
print(it, kind, what, lam)
like_o, pop_o, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw)
like_p, pop_p, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw)
with np.errstate(all='ignore'):
  ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
  print(ro[0], rp[0])
  ou, pu = pop_o.update(**lam), pop_p.update(**lam)
  th, w = O.get_theta_src_and_weights(ou, like_o.theta_gw_det)
  e = 2
  print('z stats', th.z[e].min(), th.z[e].max(), th.z[e].std(), 'w sum', w[e].sum(), 'neff', w[e].sum()**2 / (w[e]**2).sum(), 'nan w', np.isnan(w[e]).sum())
  pg_o = like_o.p_gw3d(ou)[e]; pg_p = like_p.p_gw3dfull(pu)[e]
  print('p_gw oracle: nan', np.isnan(pg_o).sum(), 'max', np.nanmax(pg_o), ' hip: nan', np.isnan(pg_p).sum(), 'max', np.nanmax(pg_p))
  print('maxabs diff', np.nanmax(np.abs(pg_o - pg_p)))
  print('numlike', like_o.compute_numlike_evs(ou), like_p.compute_numlike_evs(pu))
  print('neff_pixels', ev['neff_pixels'], 'zgrid e', ev['z_grids'][e][[0, -1]])
