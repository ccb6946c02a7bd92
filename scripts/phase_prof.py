"""Phase timing of k_kde_marg_sub on a CHM_PHASE_PROF build (scripts/build_variant.sh prof -DCHM_PHASE_PROF):
   CHIMERA_LIB=chimera_amd/lib/variants/libchimera_hip_prof.so python scripts/phase_prof.py"""
import ctypes as C, os, sys
os.environ.setdefault("CHM_KDE_ONE_ITEM", "1")       # the phase marks live in the one-item form of the GW kernel (k_kde_marg_sub)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import chimera_amd as CH
from chimera_amd import synth, _lib
from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
cfg, ev, inj = synth.make_config('C3', E=int(os.environ.get('PP_E', 1000)), I=10000)
pe = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')
th = CH.data.theta_pe_det(**{k: ev[k] for k in pe})
gcat = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gcat)
like = CH.hyperlikelihood(th, ev['z_grids'], pop, None, kind_p_gw3d='marginalized', cut_grid=2, num_bins=200)
nb = 128
H0s = np.linspace(55., 95., 4099)
lam = lambda s: [dict(H0=float(H0s[(s * nb + j) % len(H0s)])) for j in range(nb)]
L = _lib.lib()
out = (C.c_double * 8)()
for w in range(3): like.batch(lam(w))
L.chm_debug_phase(out); L.chm_debug_phase_samples(out)
n = 10
for k in range(n): like.batch(lam(3 + k))
L.chm_debug_phase(out)
v = np.array(out[:]); waves = v[7]
names = ['0 evstat+guard', '1 samples+max z', '2 histogram', '3 prefix sums', '4 constants', '5 grid loop']
tot = v[:6].sum()
print('sampled waves', int(waves), ' cycles per wave: total %.0f' % (tot / waves))
for i, nm in enumerate(names):
  print('  %-18s %8.0f cycles  %5.1f %%' % (nm, v[i] / waves, 100 * v[i] / tot))
print('GW kernel ms (HIP events, last call):', like.last_timing()[3])
L.chm_debug_phase_samples(out)
v = np.array(out[:]); blocks = v[7]
names = ['0 table staging', '1 z_ref + bracket', '2 sample loop', '3 block reduction']
tot = v[:4].sum()
print('k_samples: sampled blocks', int(blocks), ' cycles per block (two chunks): total %.0f' % (tot / blocks))
for i, nm in enumerate(names):
  print('  %-18s %8.0f cycles  %5.1f %%' % (nm, v[i] / blocks, 100 * v[i] / tot))
