import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chimera_amd as CH
from oracle import chimera_oracle as O
co, cp = O.flrw(H0=70., Om0=0.25, z_max=5.), CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.)
z = np.linspace(0.01, 3., 200)
def rel(a, b):
    a, b = np.asarray(a), np.asarray(b); m = np.isfinite(b) & (b != 0)
    return np.max(np.abs(a[m]/b[m]-1))
for fn in ('E_at_z','dCr_at_z','dL_at_z','ddLdz_at_z','dVcdz_at_z','Vc_at_z'):
    print(fn, rel(getattr(CH.cosmo, fn)(cp, z), getattr(O, fn)(co, z)))
print('It table', rel(cp.integral_invE_interp[1:], co.integral_invE_interp[1:]))
mo, mp = O.plp(), CH.mass.plp()
print('mgrid', rel(mp.m_grid, mo.m_grid), 'cdf', rel(mp.cdf_m2_conditioned[5:], mo.cdf_m2_conditioned[5:]), 'norm', mp.norm_p_m1/mo.norm_p_m1-1)
m1 = np.linspace(6., 80., 400); m2 = m1*np.linspace(0.3,0.99,400)
print('primary', rel(CH.mass.primary_mass_pdf_notnorm(mp, m1), O.primary_mass_pdf_notnorm(mo, m1)))
print('secondary', rel(CH.mass.secondary_mass_conditioned_pdf_notnorm(mp, m2, m1), O.secondary_mass_conditioned_pdf_notnorm(mo, m2, m1)))
print('p_m1m2', rel(CH.mass.p_m1m2(mp, m1, m2), O.p_m1m2(mo, m1, m2)))
print('smoothing', rel(CH.mass.smoothing(np.linspace(5.2,9.8,100), 4.8, 5.1), O.smoothing(np.linspace(5.2,9.8,100), 4.8, 5.1)))
print('rate', rel(CH.rate.merger_rate(CH.rate.madau_dickinson(), z), O.merger_rate(O.madau_dickinson(), z)))
