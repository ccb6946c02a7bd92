"""Replays tests/tools/fuzz_extreme_modes.py's random stream up to the draw whose H0 is given (DIAG_H0) and prints, for that draw,
the per-event likelihoods and the GW kernel p_gw of both paths.  Usage: DIAG_H0=<value> python tests/tools/diag_modes.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H
from oracle import chimera_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
targets = [float(x) for x in os.environ["DIAG_H0"].split(",")]
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
rng = np.random.default_rng(SEED)
found = []
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
      if any(abs(lam['H0'] - t) < 1e-9 for t in targets):
        found.append((kind, like_kw, seed, mass, cosmo, {k: float(v) for k, v in lam.items()}))

np.set_printoptions(precision=12, linewidth=200)
for kind, like_kw, seed, mass, cosmo, lam in found:
  print('=' * 30, kind, like_kw, seed, mass, cosmo, lam)
  pix = kind is not None
  cfg, ev, inj = H.small_config(E=6, S=300, P=3, Z=50, I=3000, seed=seed, ragged=True, pixelated=pix)
  like_p, pop_p, _ = H.build_product(ev, inj, pixelated=pix, kind=kind, like_kw=like_kw, models=dict(mass=mass, cosmo=cosmo))
  like_o, pop_o, _ = H.build_oracle(ev, inj, pixelated=pix, kind=kind, like_kw=like_kw, models=dict(mass=mass, cosmo=cosmo))
  with np.errstate(all='ignore'):
    ro, rp = like_o.compute_all(**lam), like_p.compute_all(**lam)
    print('log L_i oracle', ro[0]); print('log L_i hip   ', rp[0])
    bad = np.where(~np.isclose(ro[0], rp[0], rtol=1e-9, atol=1e-9))[0]
    pu, ou = pop_p.update(**lam), pop_o.update(**lam)
    fn = {None: 'p_gw1d', 'approximate': 'p_gw3dapprox', 'marginalized': 'p_gw3dmarg', 'full': 'p_gw3dfull'}[kind]
    go, gp = getattr(like_o, fn)(ou), getattr(like_p, fn)(pu)
    go, gp = np.asarray(go), np.asarray(gp)
    print('p_gw shapes', go.shape, gp.shape)
    for e in bad:
      a, b = go[e], gp[e]
      d = ~np.isclose(a, b, rtol=1e-9, atol=1e-9 * np.nanmax(np.abs(a)))
      print('event', e, 'p_gw differing entries', int(d.sum()), 'of', d.size, 'peak', np.nanmax(np.abs(a)))
      idx = np.argwhere(d)[:12]
      for i in idx:
        print('   ', tuple(i), 'oracle', a[tuple(i)], 'hip', b[tuple(i)])
      nz_o, nz_p = np.argwhere(a != 0), np.argwhere(b != 0)
      print('   nonzero oracle', len(nz_o), 'hip', len(nz_p))
      # the event's samples in source frame
      th, wts = O.get_theta_src_and_weights(ou, like_o.theta_gw_det)
      z, w = th.z[e], wts[e]
      print('   z range', np.nanmin(z), np.nanmax(z), 'std', np.std(z), 'w: nonzero', int((w > 0).sum()), 'max', np.nanmax(w), 'min>0', w[w > 0].min() if (w > 0).any() else None,
            'nan', int(np.isnan(w).sum()), 'n_eff', w.sum()**2 / (w**2).sum(), 'denormal w', int(((w > 0) & (w < 2.3e-308)).sum()))
      print('   z_grid', like_o.z_grids[e][[0, -1]], 'nonzero-w z', np.sort(z[w > 0])[:6], '...', np.sort(z[w > 0])[-3:])
  like_p.close()
