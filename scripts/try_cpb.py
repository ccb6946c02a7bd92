#!/usr/bin/env python3
"""Chunks per block of the fast sample stage at the per-rank share of an 8-GPU C3 run (-DCHM_DIAG build: CHM_OPT_DIAG_SAMP_CPB): step time of 128-draw calls.
  CHIMERA_LIB=chimera_amd/lib/variants/libchimera_hip_diag.so python3 scripts/try_cpb.py [events] [inj]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import chimera_amd as CH
from chimera_amd import synth, _lib
from chimera_amd.catalog import dVdz_completeness, pixelated_catalog

E = int(sys.argv[1]) if len(sys.argv) > 1 else 125
I = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
cfg, ev, inj = synth.make_config('C3', E=E, I=I)
pe = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')
th = CH.data.theta_pe_det(**{k: ev[k] for k in pe})
gc = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gc, scale_free=True)
sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), N_inj=inj['N_inj'], N_eff=5.)
like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', kernel='epan', bw_method=None, cut_grid=2, binning=True, num_bins=200)
H0s = np.linspace(55., 95., 4099)
draws = [dict(H0=H0s[(k * 128 + np.arange(128)) % 4099].copy()) for k in range(210)]
L = _lib.lib()
ref = None
for rep in range(2):
  for cpb in (0, 4, 3, 2, 1):
    like.set_option('diag_samp_cpb', cpb)
    for k in range(10):
      v = like.batch(draws[k])
    _lib.check(L.chm_device_synchronize(0))
    t0 = time.perf_counter()
    for k in range(200):
      v = like.batch(draws[10 + k])
    _lib.check(L.chm_device_synchronize(0))
    dt = (time.perf_counter() - t0) / 200
    ref = v[-1] if ref is None else ref
    print(f"E={E} cpb={cpb} (0 = automatic) rep{rep + 1}: {1e3 * dt:.4f} ms per 128-draw step, last {v[-1]!r} {'(bit-identical)' if v[-1] == ref else '(DIFFERS)'}", flush=True)
