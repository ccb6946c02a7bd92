#!/usr/bin/env python3
"""Where the host time of the scalar call like(**lambda) goes (run on the GPU box): CHM_HOST_PROF=1 prints the C side (before launch /
hipGraphLaunch / hipStreamSynchronize) at exit; cProfile lists the Python side."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('CHM_HOST_PROF', '1')
import numpy as np
import chimera_amd as CH
from chimera_amd import synth
from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
cfg, ev, inj = synth.make_config('C3')
th = CH.data.theta_pe_det(**{k: ev[k] for k in ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')})
gc = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gc, scale_free=True)
sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), N_inj=inj['N_inj'], N_eff=5.)
like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized')
hs = np.linspace(60., 80., 2000)
for h in hs[:20]:
  like(H0=float(h))
t0 = time.perf_counter()
for h in hs:
  like(H0=float(h))
dt = (time.perf_counter() - t0) / hs.size
print('like(H0=...) mean %.1f us per call' % (dt * 1e6))
pr = cProfile.Profile()
pr.enable()
for h in hs:
  like(H0=float(h))
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
