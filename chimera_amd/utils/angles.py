"""Angles and HEALPix indexing (reference: CHIMERA/utils/angles.py).

The reference delegates to ``healpy`` (absent here).  ``ang2pix`` / ``pix2ang`` below are an own NumPy implementation of the
published HEALPix schemes (Gorski et al. 2005, ApJ 622, 759): the RING index from eqs. 2-9 and the ring construction of the HEALPix
``ang2pix_ring`` / ``pix2ang_ring`` routines; NESTED (``nest=True``, nside a power of two) through the face / (x, y) decomposition
of the HEALPix library (``ring2xyf``, ``xyf2nest``, ``nest2xyf``, ``xyf2ring``: base-pixel tables ``jrll`` / ``jpll``, bit
interleaving inside a face).  Pinned by round trips over all pixels, equal-area counts, the nested hierarchy (the four children of
pixel p are 4p .. 4p+3 and lie inside it), the published nside = 2 RING -> NESTED table and hand-checked values
(tests/test_healpix_and_pixelization.py) -- not against healpy itself.
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


_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)     # ring of a base pixel's northernmost corner, in units of nside
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)     # its longitude, in units of pi/4


def _order(nside):
  nside = int(nside)
  if nside < 1 or nside & (nside - 1):
    raise ValueError(f"NESTED HEALPix ordering needs nside = 2^k, got {nside}")
  return nside.bit_length() - 1


def _spread_bits(v, order):
  """bit i of v -> bit 2i."""
  out = np.zeros_like(v)
  for i in range(order):
    out |= ((v >> i) & 1) << (2 * i)
  return out


def _compress_bits(v, order):
  """bit 2i of v -> bit i."""
  out = np.zeros_like(v)
  for i in range(order):
    out |= ((v >> (2 * i)) & 1) << i
  return out


def ring2nest(nside, pix):
  """NESTED index of RING pixel ``pix`` (HEALPix ``ring2xyf`` + ``xyf2nest``)."""
  order = _order(nside)
  nside = int(nside)
  pix = np.asarray(pix, dtype=np.int64)
  nl2, nl4 = 2 * nside, 4 * nside
  ncap, npix = 2 * nside * (nside - 1), 12 * nside * nside
  # north cap
  ir_n = (1 + np.floor(np.sqrt(1. + 2. * np.maximum(pix, 0))).astype(np.int64)) >> 1
  ir_n = np.where(2 * ir_n * (ir_n - 1) > pix, ir_n - 1, ir_n)
  ir_n = np.maximum(np.where(2 * ir_n * (ir_n + 1) <= pix, ir_n + 1, ir_n), 1)
  ip_n = (pix + 1) - 2 * ir_n * (ir_n - 1)
  face_n = (ip_n - 1) // ir_n
  # equatorial belt
  ipe = pix - ncap
  tmp = ipe // nl4
  ir_e = tmp + nside
  ip_e = ipe - nl4 * tmp + 1
  ks_e = (ir_e + nside) & 1
  ire = ir_e - nside + 1
  irm = nl2 + 2 - ire
  ifm = (ip_e - ire // 2 + nside - 1) // nside
  ifp = (ip_e - irm // 2 + nside - 1) // nside
  face_e = np.where(ifp == ifm, ifp | 4, np.where(ifp < ifm, ifp, ifm + 8))
  # south cap
  ips = npix - pix
  ir_s = (1 + np.floor(np.sqrt(np.maximum(2. * ips - 1., 0.))).astype(np.int64)) >> 1
  ir_s = np.where(2 * ir_s * (ir_s - 1) >= ips, ir_s - 1, ir_s)
  ir_s = np.maximum(np.where(2 * ir_s * (ir_s + 1) < ips, ir_s + 1, ir_s), 1)
  ip_s = 4 * ir_s + 1 - (ips - 2 * ir_s * (ir_s - 1))
  face_s = 8 + (ip_s - 1) // ir_s
  north, south = pix < ncap, pix >= npix - ncap
  iring = np.where(north, ir_n, np.where(south, 2 * nl2 - ir_s, ir_e))
  iphi = np.where(north, ip_n, np.where(south, ip_s, ip_e))
  nr = np.where(north, ir_n, np.where(south, ir_s, nside))
  kshift = np.where(north | south, 0, ks_e)
  face = np.clip(np.where(north, face_n, np.where(south, face_s, face_e)), 0, 11)
  irt = iring - _JRLL[face] * nside + 1
  ipt = 2 * iphi - _JPLL[face] * nr - kshift - 1
  ipt = np.where(ipt >= nl2, ipt - 8 * nside, ipt)
  ix = (ipt - irt) >> 1
  iy = (-ipt - irt) >> 1
  return face * (nside * nside) + _spread_bits(ix, order) + (_spread_bits(iy, order) << 1)


def nest2ring(nside, pix):
  """RING index of NESTED pixel ``pix`` (HEALPix ``nest2xyf`` + ``xyf2ring``)."""
  order = _order(nside)
  nside = int(nside)
  pix = np.asarray(pix, dtype=np.int64)
  nl4 = 4 * nside
  ncap, npix = 2 * nside * (nside - 1), 12 * nside * nside
  face = np.clip(pix >> (2 * order), 0, 11)
  pf = pix & (nside * nside - 1)
  ix, iy = _compress_bits(pf, order), _compress_bits(pf >> 1, order)
  jr = _JRLL[face] * nside - ix - iy - 1
  north, south = jr < nside, jr > 3 * nside
  nr = np.where(north, jr, np.where(south, nl4 - jr, nside))
  n_before = np.where(north, 2 * nr * (nr - 1), np.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - nside) * nl4))
  kshift = np.where(north | south, 0, (jr - nside) & 1)
  jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
  jp = np.where(jp > nl4, jp - nl4, np.where(jp < 1, jp + nl4, jp))
  return n_before + jp - 1


def ang2pix(nside, theta, phi, nest=False):
  """HEALPix pixel index of (theta, phi) [rad]; RING ordering, or NESTED with ``nest=True``."""
  if nest:
    return ring2nest(nside, ang2pix(nside, theta, phi, nest=False))
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
  """(theta, phi) [rad] of the centre of HEALPix pixel ``pix`` (RING, or NESTED with ``nest=True``)."""
  nside = int(nside)
  pix = np.asarray(pix, dtype=np.int64)
  if nest:
    pix = nest2ring(nside, pix)
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
  """angles.py:163-190: pixel centres of ``pixels`` (rows with their own ``nside_in``) re-indexed at ``nside_out``."""
  pixels = np.atleast_2d(pixels)
  nside_in = np.atleast_1d(nside_in)
  assert pixels.shape[0] == nside_in.shape[0], f"nside_in shape {nside_in.shape} does not match first dimension of pixels {pixels.shape}"
  results = []
  for i in range(pixels.shape[0]):
    theta, phi = pix2ang(int(nside_in[i]), pixels[i], nest=nest_in)
    results.append(ang2pix(nside_out, theta, phi, nest=nest_out))
  return np.stack(results)
