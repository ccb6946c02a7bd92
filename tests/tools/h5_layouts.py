"""HDF5 layouts of the reference's data products, written and read back through this package's loaders.  Run by
tests/test_h5_layouts.py under an interpreter that has h5py (this image: /opt/conda/bin/python3.9; the main interpreter has none).

Layouts (from the reference's loaders, nothing else is known about the Zenodo files):
  GW posterior samples   group 'posteriors' with (Nevents, Nsamples) datasets dL, m1det, m2det, phi, theta      CHIMERA/data.py:107-148
  injections             flat datasets m1src, m2src, z, dL, SNR_net, log_p_draw_nospin                          CHIMERA/data.py:150-216
  galaxy catalogue       flat datasets ra_gal, dec_gal (deg), z_cgal                                            CHIMERA/data.py:66-105
  catalogue cache        attrs max_npixels, neff_pixels; datasets p_cat, N_gal, P_compl                         CHIMERA/catalog/catalog.py:96-103
  pixelated GW catalogue datasets of theta_pe_det + group pixels_pe_all_nsides                                   CHIMERA/data.py:61-64,366-404
usage: h5_layouts.py <dir>    (exit code 0 = every round trip exact)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import h5py
import numpy as np

from chimera_amd import data as D
from chimera_amd.utils import io
from chimera_amd.catalog import dVdz_completeness, pixelated_catalog


def main(d):
  rng = np.random.default_rng(5)
  E, S, P, Z = 4, 50, 3, 12
  # --- GW posteriors
  f_pe = os.path.join(d, 'pe.h5')
  pe = {k: rng.uniform(0.1, 2., (E, S)) for k in ('dL', 'm1det', 'm2det', 'phi', 'theta')}
  with h5py.File(f_pe, 'w') as f:
    g = f.create_group('posteriors')
    for k, v in pe.items():
      g.create_dataset(k, data=v)
  th = D.load_gw_pe_samples(f_pe)
  for k in ('dL', 'm1det', 'm2det'):
    assert np.array_equal(getattr(th, k), pe[k]), k
  assert np.array_equal(th.ra, pe['phi']) and np.array_equal(th.dec, 0.5 * np.pi - pe['theta'])
  sub = D.load_gw_pe_samples(f_pe, nevents=[0, 2], nsamples=[1, 3, 5])
  assert np.array_equal(sub.dL, pe['dL'][[0, 2]][:, [1, 3, 5]])
  try:
    D.load_gw_pe_samples(f_pe, parameters=['dL', 'chi_eff'])
    raise SystemExit("missing key not reported")
  except ValueError:
    pass
  # --- injections
  f_inj = os.path.join(d, 'inj.h5')
  n = 400
  z = rng.uniform(0.05, 1., n)
  m1 = rng.uniform(10., 60., n); m2 = m1 * rng.uniform(0.2, 1., n)
  inj = dict(m1src=m1, m2src=m2, z=z, dL=rng.uniform(0.2, 6., n), SNR_net=rng.uniform(4., 30., n), log_p_draw_nospin=rng.normal(-8., 1., n))
  with h5py.File(f_inj, 'w') as f:
    for k, v in inj.items():
      f.create_dataset(k, data=v)
  ti = D.load_injection_data(f_inj, snr_cut=11.)
  keep = inj['SNR_net'] > 11.
  assert np.array_equal(ti.m1det, (m1 * (1 + z))[keep]) and np.array_equal(ti.dL, inj['dL'][keep])
  assert np.array_equal(ti.p_draw, np.exp(inj['log_p_draw_nospin'][keep]))
  # --- galaxy catalogue
  f_gal = os.path.join(d, 'gal.h5')
  gal = dict(ra_gal=rng.uniform(0., 360., 30), dec_gal=rng.uniform(-60., 60., 30), z_cgal=rng.uniform(0.01, 1., 30))
  with h5py.File(f_gal, 'w') as f:
    for k, v in gal.items():
      f.create_dataset(k, data=v)
  gc = D.load_galaxy_catalog(f_gal)
  assert np.array_equal(gc['ra'], np.deg2rad(gal['ra_gal'])) and np.array_equal(gc['z'], gal['z_cgal'])
  # --- catalogue cache: written in the reference's layout by hand, read by pixelated_catalog(gal_cat_file=...); and written by save()
  f_gc = os.path.join(d, 'galcat_test.h5')
  p_cat = rng.uniform(0., 3., (E, P, Z)); p_cat[1, 2] = -100.
  P_compl = (rng.random((E, 1, Z)) > 0.5).astype(float)
  with h5py.File(f_gc, 'w') as f:
    f.attrs['max_npixels'] = P
    f.attrs['neff_pixels'] = np.array([3, 2, 3, 3])
    f.create_dataset('p_cat', data=p_cat); f.create_dataset('N_gal', data=np.arange(E)); f.create_dataset('P_compl', data=P_compl)
  cat = pixelated_catalog(dVdz_completeness(), gal_cat_file=f_gc)
  assert cat.max_npixels == P and np.array_equal(cat.neff_pixels, [3, 2, 3, 3])
  assert np.array_equal(cat.p_cat, p_cat) and np.array_equal(cat.P_compl, P_compl) and np.array_equal(cat.N_gal, np.arange(E))
  f_gc2 = os.path.join(d, 'galcat_saved.h5')
  cat.save(f_gc2)
  with h5py.File(f_gc2, 'r') as f:
    assert int(f.attrs['max_npixels']) == P and np.array_equal(f.attrs['neff_pixels'], [3, 2, 3, 3])
    assert sorted(f.keys()) == ['N_gal', 'P_compl', 'p_cat'] and np.array_equal(f['p_cat'][:], p_cat)
  cat2 = pixelated_catalog(dVdz_completeness(), gal_cat_file=f_gc2)
  assert np.array_equal(cat2.p_cat, p_cat) and cat2.max_npixels == P
  # --- pixelated GW catalogue (datasets + the per-nside group)
  f_pix = os.path.join(d, 'pix.h5')
  full = D.theta_pe_det(dL=pe['dL'], m1det=pe['m1det'], m2det=pe['m2det'], ra=pe['phi'], dec=0.5 * np.pi - pe['theta'],
                        pixels_pe_all_nsides={'nside_8': rng.integers(0, 768, (E, S)), 'nside_16': rng.integers(0, 3072, (E, S))},
                        opt_nsides=np.array([8, 16, 8, 8]), pixels_opt_nsides=rng.integers(0, 768, (E, P)),
                        ra_pix=rng.uniform(0, 6, (E, P)), dec_pix=rng.uniform(-1, 1, (E, P)), gw_loc2d_pdf=rng.uniform(0, 1, (E, P)),
                        pixels_pe_opt_nside=rng.integers(0, 768, (E, S)))
  dsets = [k for k in D.theta_pe_pixelated_datasets if getattr(full, k) is not None]
  io.save_set(full, f_pix, datasets=dsets, groups=D.theta_pe_pixelated_groups)
  back = D.load_pixelated_gw_catalog(f_pix)
  for k in dsets:
    assert np.array_equal(getattr(back, k), getattr(full, k)), k
  assert sorted(back.pixels_pe_all_nsides) == ['nside_16', 'nside_8']
  assert np.array_equal(back.pixels_pe_all_nsides['nside_16'], full.pixels_pe_all_nsides['nside_16'])
  # --- load_data_h5 on a group
  assert sorted(io.load_data_h5(f_pe, group_h5='posteriors')) == sorted(pe)
  print("h5 layouts ok")


if __name__ == '__main__':
  main(sys.argv[1])
