import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers as H
from oracle import chimera_oracle as O
lam = {'H0': 24.290499963367672, 'Om0': 0.2152939768200043, 'gamma': 6.607447559072025, 'kappa': 4.939111736190646, 'zp': 0.4185365142451188, 'm_low': 2.918631138596546, 'm_high': 48.95454419413673, 'beta': 3.2648892657674606, 'Xi0': 2.552936597758529, 'n': 3.9237766598722676, 'alpha': 4.873948163521552, 'lambda_peak': 0.25224023681923646, 'mu_g': 35.04449729619295, 'sigma_g': 8.023910139768896, 'delta_m': 6.910295241314639}
cfg, ev, inj = H.small_config(E=6, S=300, P=3, Z=50, I=3000, seed=666080, ragged=True, pixelated=True)
models = dict(mass='plp', cosmo='mg_flrw')
like_o, pop_o, _ = H.build_oracle(ev, inj, kind='approximate', models=models)
like_p, pop_p, _ = H.build_product(ev, inj, kind='approximate', models=models)
ou, pu = pop_o.update(**lam), pop_p.update(**lam)
e = 3
with np.errstate(all='ignore'):
  a, b = like_o.p_gw1d(ou)[e], like_p.p_gw1d(pu)[e]
  print('oracle nonzero idx', np.flatnonzero(a), a[np.flatnonzero(a)])
  print('hip    nonzero idx', np.flatnonzero(b), b[np.flatnonzero(b)])
  zg = like_o.z_grids[e]
  print('zgrid around', zg[:4])
  pc = ev['p_cat'][e]
  print('p_cat at those idx', pc[:, np.flatnonzero(b)])
  th, w = O.get_theta_src_and_weights(ou, like_o.theta_gw_det)
  z = th.z[e]; sd = z.std(); print('zmin', z.min(), 'zmax', z.max(), 'sd', sd, 'lb', z.min() - 2 * sd, 'ub', z.max() + 2 * sd)
  print('L oracle', like_o.compute_numlike_evs(ou)[e], 'hip', like_p.compute_numlike_evs(pu)[e])
