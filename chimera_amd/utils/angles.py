"""Angles and HEALPix indexing (reference: CHIMERA/utils/angles.py).

The reference delegates to ``healpy`` (absent here).  ``ang2pix`` / ``pix2ang`` below are an own NumPy implementation of the
published HEALPix RING scheme (Gorski et al. 2005, ApJ 622, 759, eqs. 2-9 and the ring-index construction of the HEALPix
``ang2pix_ring`` / ``pix2ang_ring`` routines); NESTED ordering is not provided.  Pinned by round trips over all pixels,
equal-area counts and hand-checked values (tests/test_healpix_and_pixelization.py) -- not against healpy itself.
"""
import numpy as np


def th_phi_from_ra_dec(ra, dec):
  """angles.py:8-18."""
  return 0.5 * np.pi - np.asarray(dec), np.asarray(ra)


def ra_dec_from_th_phi(theta, phi):
  """angles.py:20-29."""
  return np.asarray(phi), 0.5 * np.pi - np.asarray(theta)


def nside2npix(nside):
  return 12 * int(nside) * int(nside)


def _no_nest(nest):
  if nest:
    raise NotImplementedError("NESTED HEALPix ordering is not implemented (RING only)")


def ang2pix(nside, theta, phi, nest=False):
  """HEALPix RING pixel index of (theta, phi) [rad]."""
  _no_nest(nest)
  nside = int(nside)
  theta = np.asarray(theta, dtype=np.float64)
  phi = np.asarray(phi, dtype=np.float64)
  z = np.cos(theta)
  za = np.abs(z)
  tt = np.mod(phi, 2 * np.pi) * (2. / np.pi)               # in [0, 4)
  tt = np.where(tt >= 4., 0., tt)
  nl4 = 4 * nside
  ncap = 2 * nside * (nside - 1)
  npix = 12 * nside * nside
  # equatorial region
  temp1 = nside * (0.5 + tt)
  temp2 = nside * z * 0.75
  jp = np.floor(temp1 - temp2).astype(np.int64)            # index of the ascending edge line
  jm = np.floor(temp1 + temp2).astype(np.int64)            # index of the descending edge line
  ir = nside + 1 + jp - jm                                 # ring number counted from z = 2/3, in {1, 2 nside + 1}
  kshift = 1 - (ir & 1)
  ip = (jp + jm - nside + kshift + 1) // 2
  ip = np.mod(ip, nl4)
  pix_eq = ncap + (ir - 1) * nl4 + ip
  # polar caps
  tp = tt - np.floor(tt)
  tmp = nside * np.sqrt(3. * (1. - za))
  jp2 = np.floor(tp * tmp).astype(np.int64)
  jm2 = np.floor((1. - tp) * tmp).astype(np.int64)
  irp = jp2 + jm2 + 1                                      # ring number counted from the closest pole
  ipp = np.floor(tt * irp).astype(np.int64)
  ipp = np.mod(ipp, 4 * irp)
  pix_n = 2 * irp * (irp - 1) + ipp
  pix_s = npix - 2 * irp * (irp + 1) + ipp
  return np.where(za <= 2. / 3., pix_eq, np.where(z > 0, pix_n, pix_s))


def pix2ang(nside, pix, nest=False):
  """(theta, phi) [rad] of the centre of HEALPix RING pixel ``pix``."""
  _no_nest(nest)
  nside = int(nside)
  pix = np.asarray(pix, dtype=np.int64)
  nl4 = 4 * nside
  ncap = 2 * nside * (nside - 1)
  npix = 12 * nside * nside
  fact2 = 4. / npix
  fact1 = (2 * nside) * fact2
  # north cap
  iring_n = (1 + np.floor(np.sqrt(1. + 2. * np.maximum(pix, 0))).astype(np.int64)) >> 1
  iring_n = np.where(2 * iring_n * (iring_n - 1) > pix, iring_n - 1, iring_n)       # guard the float sqrt
  iring_n = np.where(2 * iring_n * (iring_n + 1) <= pix, iring_n + 1, iring_n)
  iring_n = np.maximum(iring_n, 1)
  iphi_n = (pix + 1) - 2 * iring_n * (iring_n - 1)
  z_n = 1. - (iring_n * iring_n) * fact2
  phi_n = (iphi_n - 0.5) * (np.pi / 2) / iring_n
  # equatorial
  ipe = pix - ncap
  iring_e = ipe // nl4 + nside
  iphi_e = np.mod(ipe, nl4) + 1
  fodd = np.where(((iring_e + nside) & 1) == 1, 1.0, 0.5)
  z_e = (2 * nside - iring_e) * fact1
  phi_e = (iphi_e - fodd) * (np.pi / 2) / nside
  # south cap
  ips = npix - pix
  iring_s = (1 + np.floor(np.sqrt(np.maximum(2. * ips - 1., 0.))).astype(np.int64)) >> 1
  iring_s = np.where(2 * iring_s * (iring_s - 1) >= ips, iring_s - 1, iring_s)
  iring_s = np.where(2 * iring_s * (iring_s + 1) < ips, iring_s + 1, iring_s)
  iring_s = np.maximum(iring_s, 1)
  iphi_s = 4 * iring_s + 1 - (ips - 2 * iring_s * (iring_s - 1))
  z_s = -1. + (iring_s * iring_s) * fact2
  phi_s = (iphi_s - 0.5) * (np.pi / 2) / iring_s
  north, south = pix < ncap, pix >= npix - ncap
  z = np.where(north, z_n, np.where(south, z_s, z_e))
  phi = np.where(north, phi_n, np.where(south, phi_s, phi_e))
  return np.arccos(z), phi


def find_pix_RAdec(ra, dec, nside, nest=False):
  """angles.py:32-46."""
  theta, phi = th_phi_from_ra_dec(ra, dec)
  return ang2pix(nside, theta, phi, nest=nest)


def find_pix(theta, phi, nside, nest=False):
  """angles.py:48-60."""
  return ang2pix(nside, theta, phi, nest=nest)


def find_theta_phi(pix, nside, nest=False):
  """angles.py:61-71."""
  return pix2ang(nside, pix, nest=nest)


def find_ra_dec(pix, nside, nest=False):
  """angles.py:73-85."""
  theta, phi = find_theta_phi(pix, nside, nest=nest)
  return ra_dec_from_th_phi(theta, phi)


def hav(theta):
  """angles.py:87-88."""
  return (np.sin(np.asarray(theta) / 2))**2


def haversine(phi, theta, phi0, theta0):
  """angles.py:90-91."""
  return np.arccos(1 - 2 * (hav(theta - theta0) + hav(phi - phi0) * np.sin(theta) * np.sin(theta0)))


def healpixelize(nside, ra, dec, nest=False):
  """angles.py:111-142: dict pixel -> indices of the objects inside it."""
  healpix = find_pix_RAdec(ra, dec, nside, nest=nest)
  order = np.argsort(healpix, kind='stable')
  uniq, start = np.unique(healpix[order], return_index=True)
  return {int(k): v for k, v in zip(uniq, np.split(order, start[1:]))}


def angular_separation_from_LOS(ra, dec, ra_los, dec_los):
  """angles.py:144-160."""
  cos_angle = np.sin(dec) * np.sin(dec_los) + np.cos(dec) * np.cos(dec_los) * np.cos(ra - ra_los)
  return np.arccos(cos_angle)


def gal_to_eq(l, b):
  """angles.py:93-110: equatorial (RA, dec) [rad] from galactic (l, b) [rad], the reference's arctan form."""
  l, b = np.asarray(l, dtype=np.float64), np.asarray(b, dtype=np.float64)
  l_NCP, del_NGP, alpha_NGP = np.radians(122.93192), np.radians(27.128336), np.radians(192.859508)
  RA = np.arctan((np.cos(b) * np.sin(l_NCP - l)) / (np.cos(del_NGP) * np.sin(b) - np.sin(del_NGP) * np.cos(b) * np.cos(l_NCP - l))) + alpha_NGP
  dec = np.arcsin(np.sin(del_NGP) * np.sin(b) + np.cos(del_NGP) * np.cos(b) * np.cos(l_NCP - l))
  return RA, dec


def convert_pixelization(pixels, nside_in, nside_out, nest_in=False, nest_out=False):
  """angles.py:163-190: pixel centres of ``pixels`` (rows with their own ``nside_in``) re-indexed at ``nside_out`` (RING only)."""
  pixels = np.atleast_2d(pixels)
  nside_in = np.atleast_1d(nside_in)
  assert pixels.shape[0] == nside_in.shape[0], f"nside_in shape {nside_in.shape} does not match first dimension of pixels {pixels.shape}"
  results = []
  for i in range(pixels.shape[0]):
    theta, phi = pix2ang(int(nside_in[i]), pixels[i], nest=nest_in)
    results.append(ang2pix(nside_out, theta, phi, nest=nest_out))
  return np.stack(results)
