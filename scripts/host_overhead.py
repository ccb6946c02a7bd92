"""Where the host time of a bench step goes: lambdas, parameter packing, chm_eval (ctypes), last_timing.  Run on a GPU box:
   python scripts/host_overhead.py [--events 125 --inj 12500 --nbatch 128]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get('HO_TORCH'):
  import torch; torch.cuda.device_count()
import gc as pygc
if os.environ.get('HO_NOGC'):
  pygc.disable()
ap = argparse.ArgumentParser()
ap.add_argument('--events', type=int, default=125); ap.add_argument('--inj', type=int, default=12500)
ap.add_argument('--nbatch', type=int, default=128); ap.add_argument('--steps', type=int, default=100)
a = ap.parse_args()
import chimera_amd as CH
from chimera_amd import synth
from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
cfg, ev, inj = synth.make_config('C3', E=a.events, I=a.inj)
pe = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')
th = CH.data.theta_pe_det(**{k: ev[k] for k in pe})
gc = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gc)
sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), N_inj=inj['N_inj'], N_eff=5.)
like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', cut_grid=2, num_bins=200)
nb = a.nbatch
H0s = np.linspace(55., 95., 4099)
lambdas = lambda s: [dict(H0=float(H0s[(s * nb + j) % len(H0s)])) for j in range(nb)]
for w in range(5): like.batch(lambdas(w))
if os.environ.get('HO_FREEZE'):
  pygc.collect(); pygc.freeze()
import threading; print('threads', threading.active_count(), 'gc', pygc.isenabled(), pygc.get_count(), "objects", len(pygc.get_objects()))
T = np.zeros(5); gpu = 0.
t_all = time.perf_counter()
for k in range(a.steps):
  t0 = time.perf_counter(); lams = lambdas(5 + k)
  t1 = time.perf_counter(); arr = like._params_array(lams)
  t2 = time.perf_counter(); r = like._eval(arr)
  t3 = time.perf_counter(); ms = like.last_timing()
  t4 = time.perf_counter()
  T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, 0.]; gpu += ms[0]
t_all = time.perf_counter() - t_all
n = a.steps
print('per step [us]: lambdas %.1f  params_array %.1f  _eval(ctypes+wait) %.1f  last_timing %.1f | total %.1f | GPU-event eval %.1f' %
      (tuple(1e6 * x / n for x in (T[0], T[1], T[2], T[3], t_all)) + (1e3 * gpu / n,)))
