import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H
from oracle import chimera_oracle as O, oracle_c as OC
import chimera_amd as CH
n = 150
rng = np.random.default_rng(8888)
found = None
for kind, like_kw in [(None, {}), (None, dict(kernel='gauss', binning=False)), ('approximate', {}), ('approximate', dict(kernel='gauss', cut_grid=None)),
                      ('full', {}), ('marginalized', dict(binning=False)), ('marginalized', dict(cut_grid=None, num_bins=31))]:
  pixelated = kind is not None
  seed = int(rng.integers(1, 10**6))
  for mass, cosmo in [('plp', 'mg_flrw'), ('bpl', 'flrw'), ('tpl', 'flrw')]:
    for _ in range(n):
      lam = dict(H0=rng.uniform(20., 200.), Om0=rng.uniform(0.02, 0.98), gamma=rng.uniform(-2., 8.), kappa=rng.uniform(0., 8.), zp=rng.uniform(0.2, 5.),
                 m_low=rng.uniform(1.5, 9.), m_high=rng.uniform(40., 200.), beta=rng.uniform(-3., 6.))
      if rng.random() < 0.3: lam.update(w0=rng.uniform(-2., -0.3), wa=rng.uniform(-1., 1.))
      if rng.random() < 0.3: lam.update(Ok0=rng.uniform(-0.3, 0.3))
      if cosmo == 'mg_flrw': lam.update(Xi0=rng.uniform(0.1, 8.), n=rng.uniform(0., 6.))
      if mass == 'plp': lam.update(alpha=rng.uniform(-1., 9.), lambda_peak=rng.uniform(0., 1.), mu_g=rng.uniform(10., 70.), sigma_g=rng.uniform(0.2, 15.), delta_m=rng.uniform(0.05, 15.))
      elif mass == 'bpl': lam.update(alpha_1=rng.uniform(-2., 6.), alpha_2=rng.uniform(-1., 12.), break_fraction=rng.uniform(0.01, 0.99), delta_m=rng.uniform(0.05, 15.))
      else: lam.update(alpha=rng.uniform(-1., 9.))
      if abs(lam['H0'] - float(os.environ.get('DIAG_H0', '172.90880151995614'))) < 1e-9:
        found = (kind, like_kw, seed, mass, cosmo, {k: float(v) for k, v in lam.items()})
kind, like_kw, seed, mass, cosmo, lam = found
print(found)
pix = kind is not None
cfg, ev, inj = H.small_config(E=6, S=300, P=3, Z=50, I=3000, seed=seed, ragged=True, pixelated=pix)
like_p, pop_p, sel_p = H.build_product(ev, inj, pixelated=pix, kind=kind, like_kw=like_kw, models=dict(mass=mass, cosmo=cosmo))
like_o, pop_o, sel_o = H.build_oracle(ev, inj, pixelated=pix, kind=kind, like_kw=like_kw, models=dict(mass=mass, cosmo=cosmo))
pu, ou = pop_p.update(**lam), pop_o.update(**lam)
with np.errstate(all='ignore'):
  print('N_exp hip', sel_p.N_exp(pu), 'np', sel_o.N_exp(ou), 'c', OC.n_exp(sel_o, ou)[0])
  dl = inj['dL']
  zo, zp = O.z_from_dGW(ou.cosmo, dl), CH.cosmo.z_from_dGW(pu.cosmo, dl)
  bad = ~np.isclose(zo, zp, rtol=1e-12, atol=0, equal_nan=True)
  print('z mismatches', bad.sum(), 'of', dl.size, zo[bad][:5], zp[bad][:5], dl[bad][:5])
  dlt = O.dL_at_z(ou.cosmo, ou.cosmo.z_grid_interp)
  print('dLt sorted', np.all(np.diff(dlt) >= 0), 'nan', np.isnan(dlt).sum(), 'first nan idx', np.argmax(np.isnan(dlt)) if np.isnan(dlt).any() else None, 'max dlt', np.nanmax(dlt), 'argmax', np.nanargmax(dlt))
  dN = sel_o.dN(ou)
  print('dN nan', np.isnan(dN).sum(), 'inf', np.isinf(dN).sum(), 'nansum', np.nansum(dN))
  zg = ou.cosmo.z_grid_interp
  print('table z_max', zg[-1], 'It nan', np.isnan(ou.cosmo.integral_invE_interp).sum())
with np.errstate(all='ignore'):
  cp, co = pu.cosmo, ou.cosmo
  K = int(os.environ.get('DIAG_K', '1426'))
  print('device It', cp.integral_invE_interp[K:K+8])
  print('oracle It', co.integral_invE_interp[K:K+8])
  from chimera_amd.population._base import make_params, model_tables
  t = model_tables(make_params(cosmo=cp))
  print('device dLt', t['dL_interp'][K:K+8])
  print('oracle dLt', dlt[K:K+8])
  print('E(z) device', CH.cosmo.E_at_z(cp, co.z_grid_interp[K:K+8]), 'oracle', O.E_at_z(co, co.z_grid_interp[K:K+8]))
