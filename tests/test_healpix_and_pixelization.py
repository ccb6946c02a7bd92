"""HEALPix RING / NESTED indexing (own implementation; healpy is absent), file loaders and the GW-catalogue pixelisation
(reference: CHIMERA/utils/angles.py, CHIMERA/utils/io.py, CHIMERA/data.py:107-404)."""
import numpy as np
import pytest

import chimera_amd as CH
from chimera_amd import data as D
from chimera_amd.utils import angles as A
from oracle import chimera_oracle as O


# ----------------------------------------------------------------------------------------------------------
# HEALPix pins (not against healpy: "parity unpinned"; these are properties of the published RING scheme)
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('nside', [1, 2, 4, 8, 32, 128])
def test_healpix_round_trip_and_ranges(nside):
  npix = A.nside2npix(nside)
  assert npix == 12 * nside**2
  p = np.arange(npix)
  th, ph = A.pix2ang(nside, p)
  np.testing.assert_array_equal(A.ang2pix(nside, th, ph), p)
  assert np.all((th > 0) & (th < np.pi)) and np.all((ph >= 0) & (ph < 2 * np.pi))
  # iso-latitude rings: 4 nside - 1 distinct colatitudes, ring sizes 4, 8, ..., 4 nside, ..., 8, 4
  z = np.round(np.cos(th), 12)
  uz, counts = np.unique(z, return_counts=True)
  assert len(uz) == 4 * nside - 1
  assert counts.min() == 4 and counts.max() == 4 * nside
  np.testing.assert_allclose(np.sort(uz), -np.sort(-uz)[::-1] * 1 if False else np.sort(uz))      # (symmetry checked below)
  np.testing.assert_allclose(np.sort(uz), np.sort(-uz), atol=1e-12)


def test_healpix_known_values_and_equal_area():
  th, ph = A.pix2ang(1, [0, 4, 8])                       # z = 2/3, 0, -2/3
  np.testing.assert_allclose(np.cos(th), [2 / 3, 0., -2 / 3], atol=1e-15)
  np.testing.assert_allclose(ph, [np.pi / 4, 0., np.pi / 4])
  th, ph = A.pix2ang(2, [0, 3, 4, 47])
  np.testing.assert_allclose(np.cos(th), [1 - 1 / 12, 1 - 1 / 12, 2 / 3, -(1 - 1 / 12)], atol=1e-15)
  np.testing.assert_allclose(ph, [np.pi / 4, 7 * np.pi / 4, np.pi / 8, 7 * np.pi / 4])
  # poles and wrap-around
  assert A.ang2pix(4, 1e-9, 0.1) == 0 and A.ang2pix(4, np.pi - 1e-9, 6.2) == A.nside2npix(4) - 1
  assert A.ang2pix(8, 1.0, 0.3) == A.ang2pix(8, 1.0, 0.3 + 2 * np.pi) == A.ang2pix(8, 1.0, 0.3 - 2 * np.pi)
  # equal areas: uniform points fill all pixels evenly
  rng = np.random.default_rng(0)
  n = 600_000
  z, phi = rng.uniform(-1, 1, n), rng.uniform(0, 2 * np.pi, n)
  c = np.bincount(A.ang2pix(8, np.arccos(z), phi), minlength=A.nside2npix(8))
  assert np.abs(c - n / 768).max() < 5.5 * np.sqrt(n / 768)
  # a point lies within ~ one pixel scale of its pixel centre
  pp = A.ang2pix(16, np.arccos(z[:5000]), phi[:5000])
  tc, pc = A.pix2ang(16, pp)
  sep = A.angular_separation_from_LOS(phi[:5000], np.pi / 2 - np.arccos(z[:5000]), pc, np.pi / 2 - tc)
  assert sep.max() < 1.1 * np.sqrt(4 * np.pi / A.nside2npix(16))
  with pytest.raises(ValueError):
    A.ang2pix(6, 1., 1., nest=True)                        # NESTED needs nside = 2^k


# RING -> NESTED at nside = 2, the table of the HEALPix scheme (what healpy.ring2nest(2, arange(48)) returns)
RING2NEST_NSIDE2 = [3, 7, 11, 15, 2, 1, 6, 5, 10, 9, 14, 13, 19, 0, 23, 4, 27, 8, 31, 12, 17, 22, 21, 26, 25, 30, 29, 18, 16, 35, 20, 39, 24, 43,
                    28, 47, 34, 33, 38, 37, 42, 41, 46, 45, 32, 36, 40, 44]


def test_healpix_nested_ordering():
  """NESTED indexing (CHIMERA/data.py:266,288,325 and utils/angles.py:32-85 take ``nest``): the published nside = 2 table, bijection
  with RING, base pixels (nside = 1: NESTED = RING = face number), and the hierarchy that defines the scheme -- the children of
  pixel p at 2 nside are 4p .. 4p + 3 and their centres lie inside p."""
  np.testing.assert_array_equal(A.ring2nest(2, np.arange(48)), RING2NEST_NSIDE2)
  np.testing.assert_array_equal(A.nest2ring(2, RING2NEST_NSIDE2), np.arange(48))
  np.testing.assert_array_equal(A.ring2nest(1, np.arange(12)), np.arange(12))
  for nside in (1, 2, 4, 16, 64, 512):
    npix = A.nside2npix(nside)
    r = np.arange(npix) if nside <= 64 else np.random.default_rng(1).integers(0, npix, 200_000)
    n = A.ring2nest(nside, r)
    np.testing.assert_array_equal(A.nest2ring(nside, n), r)
    if nside <= 64:
      assert np.array_equal(np.sort(n), np.arange(npix))
    th, ph = A.pix2ang(nside, n, nest=True)
    thr, phr = A.pix2ang(nside, r)
    np.testing.assert_array_equal(th, thr); np.testing.assert_array_equal(ph, phr)
    np.testing.assert_array_equal(A.ang2pix(nside, th, ph, nest=True), n)
  for nside in (1, 2, 8, 32):
    ch = np.arange(A.nside2npix(2 * nside))
    th, ph = A.pix2ang(2 * nside, ch, nest=True)
    np.testing.assert_array_equal(A.ang2pix(nside, th, ph, nest=True), ch // 4)
  # the reference's helpers pass `nest` through
  ra, dec = np.array([0.3, 2.0, 5.1]), np.array([-0.4, 0.9, 0.1])
  pn = A.find_pix_RAdec(ra, dec, 16, nest=True)
  np.testing.assert_array_equal(pn, A.ring2nest(16, A.find_pix_RAdec(ra, dec, 16)))
  r2, d2 = A.find_ra_dec(pn, 16, nest=True)
  r3, d3 = A.find_ra_dec(A.find_pix_RAdec(ra, dec, 16), 16)
  np.testing.assert_array_equal(r2, r3); np.testing.assert_array_equal(d2, d3)
  np.testing.assert_array_equal(A.convert_pixelization(np.array([pn]), [16], 4, nest_in=True, nest_out=True)[0], pn // 16)


def test_angle_helpers():
  ra, dec = np.array([0.3, 2.0]), np.array([-0.4, 0.9])
  th, ph = A.th_phi_from_ra_dec(ra, dec)
  r2, d2 = A.ra_dec_from_th_phi(th, ph)
  np.testing.assert_allclose([r2, d2], [ra, dec])
  np.testing.assert_array_equal(A.find_pix_RAdec(ra, dec, 16), A.ang2pix(16, th, ph))
  r3, d3 = A.find_ra_dec(A.find_pix_RAdec(ra, dec, 64), 64)
  assert np.all(A.angular_separation_from_LOS(ra, dec, r3, d3) < 0.02)
  assert A.angular_separation_from_LOS(0., 0., np.pi / 2, 0.) == pytest.approx(np.pi / 2)
  groups = A.healpixelize(4, np.array([0.1, 0.1001, 3.0]), np.array([0.2, 0.2001, -1.0]))
  assert sorted(len(v) for v in groups.values()) == [1, 2]
  # galactic -> equatorial (angles.py:93-110): the north galactic pole, and a point on the galactic equator
  ra_p, dec_p = A.gal_to_eq(np.radians(122.93192), np.radians(90.))
  assert np.degrees(ra_p) == pytest.approx(192.859508) and np.degrees(dec_p) == pytest.approx(27.128336)
  _, dec_e = A.gal_to_eq(np.radians(122.93192), 0.)
  assert np.degrees(dec_e) == pytest.approx(90. - 27.128336)
  # re-indexing pixel centres at another resolution (angles.py:163-190): refine then coarsen is the identity, rows keep their nside
  pix = np.vstack([np.arange(48), np.arange(100, 148)])
  fine = A.convert_pixelization(pix, [2, 8], 64)
  assert fine.shape == pix.shape
  np.testing.assert_array_equal(A.convert_pixelization(fine[:1], [64], 2), pix[:1])
  np.testing.assert_array_equal(A.convert_pixelization(fine[1:], [64], 8), pix[1:])
  with pytest.raises(AssertionError):
    A.convert_pixelization(pix, [2], 64)


# ----------------------------------------------------------------------------------------------------------
# sky-confidence pixels, loaders
# ----------------------------------------------------------------------------------------------------------
def test_sky_conf_pixels():
  # 10 samples: pixel 5 holds 6, pixel 7 holds 3, pixel 2 holds 1 (nside = 1 -> 12 pixels)
  hpx = np.array([5] * 6 + [7] * 3 + [2])
  np.testing.assert_array_equal(D.compute_sky_conf_event(hpx, 0.5, 1), [5])
  np.testing.assert_array_equal(D.compute_sky_conf_event(hpx, 0.8, 1), [5, 7])
  np.testing.assert_array_equal(D.compute_sky_conf_event(hpx, 0.95, 1), [2, 5, 7])
  # every event at once (sky_conf_pixels) against the per-event rule of data.py:239-260 spelt out on the full map of 12 nside^2 pixels
  rng = np.random.default_rng(4)
  for nside in (1, 2, 8):
    npix = 12 * nside * nside
    centre = rng.integers(0, npix, size=(7, 1))
    pix = (centre + rng.geometric(0.3, size=(7, 500)) - 1) % npix
    for level in (0.5, 0.9, 0.99):
      got = D.sky_conf_pixels(pix, level, nside)
      for e in range(7):
        p = np.bincount(pix[e], minlength=npix) / pix.shape[1]
        srt = np.sort(p)[::-1]
        thr = srt[np.searchsorted(np.cumsum(srt), level)]
        np.testing.assert_array_equal(got[e], np.flatnonzero(p >= thr))
  # ragged rows -> padded array
  rows = [np.array([3, 1, 2]), np.array([7]), np.array([5, 6])]
  np.testing.assert_array_equal(D._pad_arr_list(rows, -100), [[3, 1, 2], [7, -100, -100], [5, 6, -100]])
  assert D._pad_arr_list([np.array([1.5]), np.array([2.5, 3.5])], -100.).dtype == np.float64


def test_loaders_round_trip(tmp_path):
  rng = np.random.default_rng(1)
  E, S = 5, 40
  pe = {f'posteriors/{k}': rng.random((E, S)) + 1 for k in ('dL', 'm1det', 'm2det', 'phi', 'theta')}
  f = str(tmp_path / 'pe.npz'); np.savez(f, **pe)
  th = D.load_gw_pe_samples(f, nevents=[0, 2, 4], nsamples=None)
  assert th.dL.shape == (3, S) and np.all(th.pe_prior == 1.)
  np.testing.assert_allclose(th.dec, np.pi / 2 - pe['posteriors/theta'][[0, 2, 4]])
  np.testing.assert_allclose(th.ra, pe['posteriors/phi'][[0, 2, 4]])
  th2 = D.load_gw_pe_samples(f, parameters=['dL', 'm1det', 'm2det'], nevents=2, nsamples=10)
  assert th2.dL.shape == (2, 10)
  with pytest.raises(ValueError):
    D.load_gw_pe_samples(f, parameters=['dL', 'nope'])
  n = 200
  m1 = rng.uniform(10, 50, n); m2 = m1 * rng.uniform(0.2, 1., n); z = rng.uniform(0.1, 1., n)
  inj = dict(m1src=m1, m2src=m2, z=z, dL=rng.uniform(0.5, 5., n), SNR_net=rng.uniform(5, 40, n), log_p_draw_nospin=rng.normal(-5, 1, n))
  fi = str(tmp_path / 'inj.npz'); np.savez(fi, **inj)
  ti = D.load_injection_data(fi, snr_cut=20)
  keep = inj['SNR_net'] > 20
  assert ti.dL.shape == (keep.sum(),)
  np.testing.assert_allclose(ti.m1det, (m1 * (1 + z))[keep])
  np.testing.assert_allclose(ti.p_draw, np.exp(inj['log_p_draw_nospin'][keep]))
  # theta container <-> file
  from chimera_amd.utils.io import save_set, load_set
  fs = str(tmp_path / 'th.npz')
  th3 = th.update(pixels_pe_all_nsides={'nside_8': np.arange(6).reshape(3, 2)})
  save_set(th3, fs, datasets=['dL', 'm1det'], groups=['pixels_pe_all_nsides'])
  back = load_set(D.theta_pe_det(), fs, datasets=['dL', 'm1det'], groups=['pixels_pe_all_nsides'])
  np.testing.assert_array_equal(back.dL, th.dL)
  np.testing.assert_array_equal(back.pixels_pe_all_nsides['nside_8'], np.arange(6).reshape(3, 2))


# ----------------------------------------------------------------------------------------------------------
# pixelisation end to end (GPU: the 2-D KDE at the pixel centres and the downstream likelihood)
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_pixelize_gw_catalog_and_downstream_likelihood(tmp_path):
  from chimera_amd import synth
  from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
  cfg, ev, inj = synth.make_config('C2', E=6, S=512, P=4, Z=64, I=2000)
  # widen the sky posteriors so that several HEALPix pixels are hit
  rng = np.random.default_rng(3)
  ra = np.mod(ev['ra'] + 0.15 * rng.standard_normal(ev['ra'].shape), 2 * np.pi)
  dec = np.clip(ev['dec'] + 0.1 * rng.standard_normal(ev['dec'].shape), -1.5, 1.5)
  th = CH.data.theta_pe_det(m1det=ev['m1det'], m2det=ev['m2det'], dL=ev['dL'], ra=ra, dec=dec, pe_prior=ev['pe_prior'])
  pix = D.pixelize_gw_catalog(th, nside_list=[32, 16, 8, 4], mean_npixels_event=8, sky_conf=0.9, prefix=str(tmp_path / 'cat'))
  E = cfg['E']
  P = pix.pixels_opt_nsides.shape[1]
  assert pix.opt_nsides.shape == (E,) and set(pix.opt_nsides) <= {32, 16, 8, 4}
  for e in range(E):
    good = pix.pixels_opt_nsides[e][pix.pixels_opt_nsides[e] != -100]
    n = len(good)
    assert np.all(np.isin(pix.pixels_pe_opt_nside[e], good))                       # every sample sits in an event pixel
    np.testing.assert_array_equal(good, D.compute_sky_conf_event(A.find_pix_RAdec(ra[e], dec[e], pix.opt_nsides[e]), 0.9, pix.opt_nsides[e]))
    r, d = A.find_ra_dec(good, pix.opt_nsides[e])
    np.testing.assert_allclose(pix.ra_pix[e, :n], r); np.testing.assert_allclose(pix.dec_pix[e, :n], d)
    assert np.all(pix.ra_pix[e, n:] == -100.) and np.all(pix.gw_loc2d_pdf[e, n:] == -100.)
    ref = O.gkde_nd(np.array([ra[e], dec[e]]), np.array([r, d]))                  # jax_gkde_nd == gaussian_kde (math.py:96)
    np.testing.assert_allclose(pix.gw_loc2d_pdf[e, :n], ref, rtol=1e-10)
  # file round trip
  import glob
  back = D.load_pixelated_gw_catalog(glob.glob(str(tmp_path / 'cat_pixelated_*.npz'))[0])
  np.testing.assert_array_equal(back.pixels_opt_nsides, pix.pixels_opt_nsides)
  np.testing.assert_array_equal(back.pixels_pe_all_nsides['nside_8'], pix.pixels_pe_all_nsides['nside_8'])
  # downstream: catalogue term from a galaxy sample indexed with the same HEALPix code, then the likelihood
  cosmo = CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.)
  zg = CH.compute_z_grids(cosmo, pix, cosmo_prior={'H0': [20, 200]}, z_int_res=64)
  ng = 4000
  gal = dict(z=rng.uniform(0.08, 1.2, ng))
  gra, gdec = rng.uniform(0, 2 * np.pi, ng), np.arcsin(rng.uniform(-1, 1, ng))
  for ns in np.unique(pix.opt_nsides):
    gal[f'pix{ns}'] = A.find_pix_RAdec(gra, gdec, ns)
  gc = pixelated_catalog(dVdz_completeness(), cosmo=cosmo, z_grids=zg, data_gal=gal, data_gw_pixelated=pix, z_err=0.01)
  assert gc.p_cat.shape == (E, P, 64) and gc.max_npixels == P
  pop = CH.population(cosmo, CH.mass.plp(), CH.rate.madau_dickinson(), gal_cat=gc)
  sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), inj['N_inj'])
  for kind in ('marginalized', 'approximate', 'full'):
    like = CH.hyperlikelihood(pix, zg, pop, sel, kind_p_gw3d=kind)
    r = like.compute_all(H0=70.)
    # same inputs through the oracle
    tho = O.theta_pe_det(m1det=pix.m1det, m2det=pix.m2det, dL=pix.dL, ra=pix.ra, dec=pix.dec, pe_prior=pix.pe_prior,
                         pixels_opt_nsides=pix.pixels_opt_nsides, ra_pix=pix.ra_pix, dec_pix=pix.dec_pix,
                         gw_loc2d_pdf=pix.gw_loc2d_pdf, pixels_pe_opt_nside=pix.pixels_pe_opt_nside)
    gco = O.pixelated_catalog(O.dVdz_completeness(), gc.p_cat, zg, gc.neff_pixels)
    popo = O.population(O.flrw(H0=70., Om0=0.25, z_max=5.), O.plp(), O.madau_dickinson(), gal_cat=gco)
    selo = O.selection_function(O.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), inj['N_inj'])
    ro = O.hyperlikelihood(tho, zg, popo, selo, kind_p_gw3d=kind).compute_all(H0=70.)
    from tests import helpers as H
    H.assert_loglike_close(r[0], ro[0], rtol=1e-9, atol=1e-9)
