import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H
from oracle import chimera_oracle as O
cfg, ev, inj = H.small_config(E=6, S=256, P=4, Z=64, I=3000, seed=11, ragged=True)
for kind, pix in (('marginalized', True), ('approximate', True), (None, False)):
    if pix:
        lo, po, so = H.build_oracle(ev, inj, kind=kind); lp, pp, sp = H.build_product(ev, inj, kind=kind)
    else:
        c2, ev2, inj2 = H.small_config(E=5, S=300, Z=80, I=3000, seed=12, pixelated=False)
        lo, po, so = H.build_oracle(ev2, inj2, pixelated=False); lp, pp, sp = H.build_product(ev2, inj2, pixelated=False)
    ro, rp = lo.compute_all(H0=70.), lp.compute_all(H0=70.)
    print(kind, 'log L_i abs diff', np.abs(rp[0]-ro[0]).max(), 'logNexp diff', abs(rp[2]-ro[2]), 'log_hyper diff', abs(rp[3]-ro[3]))
    pop_o, pop_p = lo.population.update(H0=70.), lp.population.update(H0=70.)
    go = lo.p_gw3d(pop_o) if lo.pixelated else lo.p_gw1d(pop_o); gp = lp.p_gw3d(pop_p) if lp.pixelated else lp.p_gw1d(pop_p)
    if lo.pixelated:
        valid = np.arange(go.shape[1])[None,:] < np.asarray(lo.neff_pixels)[:,None]; go, gp = go[valid], gp[valid]
    m = np.isfinite(go) & (np.abs(go) > 1e-6*np.abs(go).max())
    print('   p_gw max rel diff', np.max(np.abs(gp[m]-go[m])/np.abs(go[m])))
import chimera_amd as CH
co, cp = O.flrw(H0=70., Om0=0.25, z_max=5.), CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.)
d = np.linspace(0.1, 8., 50)
print('z_from_dGW rel', np.max(np.abs(CH.cosmo.z_from_dGW(cp, d)/O.z_from_dGW(co, d)-1)))
m1 = np.linspace(6., 80., 40); m2 = m1*0.7
print('p_m1m2 rel', np.max(np.abs(CH.mass.p_m1m2(CH.mass.plp(), m1, m2)/O.p_m1m2(O.plp(), m1, m2)-1)))
