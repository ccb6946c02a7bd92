"""Diagnostics for one hyper-parameter draw: tables, per-event likelihoods and N_exp of the HIP path vs the C and NumPy oracles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers as H
from oracle import oracle_c as OC, chimera_oracle as O
import chimera_amd as CH

lam = {'H0': 78.70065408351988, 'Om0': 0.25296675808021296, 'gamma': 3.3335476960053523, 'kappa': 4.896156387495601, 'zp': 1.6636800224745663,
       'm_low': 4.090710327355209, 'm_high': 97.51325226101044, 'beta': 1.4316645633120282, 'alpha': 3.4238928899556047}
cfg, ev, inj = H.small_config(E=24, S=1024, P=6, Z=200, I=20000, seed=23, ragged=True)
models = dict(mass='tpl', cosmo='flrw')
like_p, pop_p, sel_p = H.build_product(ev, inj, models=models)
like_o, pop_o, sel_o = H.build_oracle(ev, inj, models=models)
rp = like_p.compute_all(**lam)
rc = OC.compute_all(like_o, lam, nthreads=8)
ro = like_o.compute_all(**lam)
print('log_hyper hip/c/numpy', rp[3], rc[3], ro[3])
print('logNexp   hip/c/numpy', rp[2], rc[2], ro[2])
d = rp[0] - rc[0]
print('per-event diff', np.array2string(d, precision=3))
pu, ou = pop_p.update(**lam), pop_o.update(**lam)
mp, mo = pu.mass, ou.mass
print('m_grid last hip', repr(mp.m_grid[-1]), 'oracle', repr(mo.m_grid[-1]), 'm_high', repr(lam['m_high']))
print('m_grid first hip', repr(mp.m_grid[0]), 'oracle', repr(mo.m_grid[0]), 'm_low', repr(lam['m_low']))
print('max rel diff m_grid', np.max(np.abs(mp.m_grid / mo.m_grid - 1)), 'cdf', np.max(np.abs(mp.cdf_m2_conditioned[1:] / mo.cdf_m2_conditioned[1:] - 1)),
      'norm', mp.norm_p_m1 / mo.norm_p_m1 - 1)
co_p, co_o = pu.cosmo, ou.cosmo
print('zt rel', np.max(np.abs(co_p.z_grid_interp[1:] / co_o.z_grid_interp[1:] - 1)), 'It rel', np.max(np.abs(co_p.integral_invE_interp[1:] / co_o.integral_invE_interp[1:] - 1)))
m1 = np.linspace(3., 110., 2001); m2 = m1 * 0.7
print('p_m1m2 max rel', np.nanmax(np.abs(CH.mass.p_m1m2(mp, m1, m2) / np.where(O.p_m1m2(mo, m1, m2) == 0, np.nan, O.p_m1m2(mo, m1, m2)) - 1)))
from chimera_amd.population._base import make_params, model_eval
from chimera_amd import _lib
pf = model_eval(make_params(mass=mp), _lib.F_PM1M2_FUSED, m1, m2)
ref = O.p_m1m2(mo, m1, m2)
print('fused max rel', np.nanmax(np.abs(pf / np.where(ref == 0, np.nan, ref) - 1)), 'zeros agree', np.array_equal(pf == 0, ref == 0))
