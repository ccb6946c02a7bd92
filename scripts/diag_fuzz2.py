import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers as H
from oracle import oracle_c as OC, chimera_oracle as O
lam={'H0': 20.475122477377834, 'Om0': 0.04178692963304655, 'gamma': 1.541044927483358, 'kappa': 4.184002999149263, 'zp': 4.628008531223384, 'm_low': 6.157857522308403, 'm_high': 163.37876011732743, 'beta': -2.9138266003982998, 'Ok0': -0.27540350807880004, 'Xi0': 0.28514090098152856, 'n': 2.349145907232978, 'alpha': 2.864398050174438, 'lambda_peak': 0.9103270676574132, 'mu_g': 42.242161006238454, 'sigma_g': 6.919694405137084, 'delta_m': 6.71820961358286}
cfg, ev, inj = H.small_config(E=16, S=512, P=5, Z=120, I=8000, seed=77, ragged=True)
models=dict(mass='plp', cosmo='mg_flrw')
like_p, pop_p, sel_p = H.build_product(ev, inj, models=models)
like_o, pop_o, sel_o = H.build_oracle(ev, inj, models=models)
with np.errstate(all='ignore'):
  rp = like_p.compute_all(**lam); ro = like_o.compute_all(**lam); rc = OC.compute_all(like_o, lam, nthreads=4)
  print('hip', rp[1:], '\nnp ', ro[1:], '\nc  ', rc[1:])
  pu, ou = pop_p.update(**lam), pop_o.update(**lam)
  print('N_exp hip', sel_p.N_exp(pu), 'np', sel_o.N_exp(ou), 'c', OC.n_exp(sel_o, ou))
  dN = sel_o.dN(ou)
  print('dN: nan', np.isnan(dN).sum(), 'inf', np.isinf(dN).sum(), 'sum', np.nansum(dN), 'sum sq', np.sum(dN**2))
  t = OC.tables(ou)
  dlt = O.dL_at_z(ou.cosmo, ou.cosmo.z_grid_interp)
  print('dLt monotonic:', np.all(np.diff(dlt) >= 0), 'nan in dLt', np.isnan(dlt).sum(), 'max z', ou.cosmo.z_grid_interp[-1])
  print('c dLt vs np', np.nanmax(np.abs(t['dLt'] - dlt)))
with np.errstate(all='ignore'):
  print('hip logL', np.array2string(rp[0], precision=6))
  print('np  logL', np.array2string(ro[0], precision=6))
  th, w = O.get_theta_src_and_weights(ou, like_o.theta_gw_det)
  for e in np.flatnonzero(~np.isclose(rp[0], ro[0], rtol=1e-6, atol=1e-6) | (np.isinf(ro[0]) != np.isinf(rp[0]))):
    z = th.z[e]
    print('event', e, 'nan z', np.isnan(z).sum(), 'nan w', np.isnan(w[e]).sum(), 'inf w', np.isinf(w[e]).sum(), 'zmin/max', np.nanmin(z), np.nanmax(z), 'sum w', np.nansum(w[e]),
          'dL range', like_o.theta_gw_det.dL[e].min(), like_o.theta_gw_det.dL[e].max())
    numl = like_o.compute_numlike_evs(ou)[e]
    print('   oracle L_i', numl, ' hip L_i', like_p.compute_numlike_evs(pu)[e])
    pg = like_o.p_gw3d(ou)[e]; pgp = like_p.p_gw3dmarg(pu)[e]
    print('   oracle p_gw nan count', np.isnan(pg).sum(), 'hip', np.isnan(pgp).sum(), 'oracle max', np.nanmax(pg), 'hip max', np.nanmax(pgp))
    zg = like_o.z_grids[e]
    jac = O.ddLdz_at_z(ou.cosmo, zg) * (1 + zg)**2
    print('   jacobian nan', np.isnan(jac).sum(), 'zgrid range', zg[0], zg[-1], ' p_cbc nan', np.isnan(O.p_cbc(ou, like_o.z_grids)[e]).sum())
