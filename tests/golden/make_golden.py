#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz.

The reference itself cannot be imported here (jax, equinox, plum, numba absent), so these vectors are produced by
the oracle (oracle/chimera_oracle.py) on tiny seeded inputs: they pin the oracle against silent drift (CPU test) and
give the HIP path fixed expected outputs (GPU test).  Each file stores the full inputs and the expected outputs of
``hyperlikelihood.compute_all`` for three hyper-parameter draws, plus a p_gw slice.

  python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests import helpers as H

LAMBDAS = [dict(H0=70.), dict(H0=58., alpha=3.1, gamma=2.2), dict(H0=88., mu_g=31., lambda_peak=0.06)]

CASES = {
  # name: (pixelated, kind, models, like_kw)
  'marg_flrw_plp': (True, 'marginalized', {}, {}),
  'approx_flrw_plp': (True, 'approximate', {}, {}),
  'full_flrw_plp': (True, 'full', {}, {}),
  'marg_mg_plp': (True, 'marginalized', dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.8, n=1.9)), {}),
  'marg_flrw_tpl': (True, 'marginalized', dict(mass='tpl'), {}),
  'marg_flrw_bpl': (True, 'marginalized', dict(mass='bpl'), {}),
  'approx_gauss_nobin': (True, 'approximate', {}, dict(kernel='gauss', binning=False)),
  'onedim_flrw_plp': (False, None, {}, {}),
  'onedim_mg_tpl_pl': (False, None, dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=0.7, n=1.2), mass='tpl', rate='power_law'), {}),
}

ARRAY_KEYS = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf',
              'pixels_pe_opt_nside', 'neff_pixels', 'z_grids', 'p_cat')


def make_case(name):
  pixelated, kind, models, like_kw = CASES[name]
  if pixelated:
    cfg, ev, inj = H.small_config(E=4, S=64, P=3, Z=32, I=256, seed=101, ragged=True)
  else:
    cfg, ev, inj = H.small_config(E=4, S=96, Z=40, I=256, seed=102, pixelated=False)
  like, pop, sel = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind or 'marginalized', models=models, like_kw=like_kw)
  out = {}
  for k in ARRAY_KEYS:
    if k in ev:
      out['ev_' + k] = ev[k]
  for k in ('m1det', 'm2det', 'dL', 'p_draw'):
    out['inj_' + k] = inj[k]
  out['N_inj'] = np.float64(inj['N_inj'])
  res = [like.compute_all(**lam) for lam in LAMBDAS]
  out['log_like_evs'] = np.array([r[0] for r in res])
  out['log_num'] = np.array([r[1] for r in res])
  out['log_Nexp'] = np.array([r[2] for r in res])
  out['log_hyper'] = np.array([r[3] for r in res])
  p0 = like.population.update(**LAMBDAS[0])
  out['p_gw'] = like.p_gw3d(p0) if pixelated else like.p_gw1d(p0)
  return out


if __name__ == '__main__':
  for name in CASES:
    d = make_case(name)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print(name, 'log_hyper =', d['log_hyper'])
