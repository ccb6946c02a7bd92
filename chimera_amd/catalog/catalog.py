"""Galaxy-catalogue redshift prior (reference: CHIMERA/catalog/catalog.py).

Runtime part of the plugin: ``p_gal(cosmo, z) = fR * p_cat + (1 - P_compl) * p_bkg`` with the -100 sentinel
(catalog.py:197-203).  Inside ``hyperlikelihood`` this expression is fused into the HIP integrand kernel; the
``p_gal`` method here is the standalone (array) form of the same plugin call.
``p_cat`` comes from arrays or an ``.npz`` cache (the reference's HDF5 cache, catalog.py:96-103, needs h5py).
"""
import numpy as np
from ..utils.config import logger
from ..population.cosmo import dVcdz_at_z
from ..data import theta_src


class empty_catalog(object):
  """catalog.py:19-43."""

  def __init__(self, p_bkg="dVdz"):
    self.p_cat = 0.
    self.N_gal = 0.
    self.P_compl = 0.
    if p_bkg != "dVdz":
      raise ValueError("only the 'dVdz' background is built into the HIP path")
    self.p_bkg = dVcdz_at_z
    self.max_npixels = None
    self.neff_pixels = None
    self.z_range = (0.073, 1.3)

  def p_gal(self, cosmo_lambdas, z):
    return self.p_bkg(cosmo_lambdas, z)


class pixelated_catalog(object):
  """catalog.py:51-203.

  Construct from a cache file (``gal_cat_file='...npz'`` holding max_npixels, neff_pixels, p_cat, N_gal, P_compl)
  or from arrays (``p_cat=`` (E,P,Z) padded with -100, ``z_grids=`` (E,Z), ``neff_pixels=`` (E,)).
  """

  def __init__(self, completeness, gal_cat_file=None, cosmo=None, z_grids=None, fname_data_gal=None,
               data_gw_pixelated=None, z_err=1, weights=None, mask_gal=None, sumgauss="dVdz", reshuffle=False,
               out_file=None, p_cat=None, neff_pixels=None, N_gal=None):
    self.completeness = completeness
    self.p_bkg = self.completeness.p_bkg
    self.fR = self.completeness.fR
    self.z_range = tuple(float(v) for v in completeness.z_range)
    self.attr_gal_cat = ['max_npixels', 'neff_pixels']
    self.data_gal_cat = ['p_cat', 'N_gal', 'P_compl']
    if gal_cat_file is not None:
      logger.info(f"Loading gal_cat object from {gal_cat_file}")
      with np.load(gal_cat_file) as d:
        self.p_cat = np.ascontiguousarray(d['p_cat'], dtype=np.float64)
        self.N_gal = d['N_gal']
        self.P_compl = np.ascontiguousarray(d['P_compl'], dtype=np.float64)
        self.neff_pixels = np.asarray(d['neff_pixels'])
        self.max_npixels = int(d['max_npixels'])
    elif p_cat is not None:
      if z_grids is None:
        raise ValueError("pixelated_catalog: `z_grids` is needed with `p_cat`")
      self.p_cat = np.ascontiguousarray(p_cat, dtype=np.float64)
      if self.p_cat.ndim != 3:
        raise ValueError("pixelated_catalog: `p_cat` must have shape (Nevents, max_npixels, z_int_res)")
      self.max_npixels = self.p_cat.shape[1]
      if neff_pixels is None:
        if data_gw_pixelated is None:
          raise ValueError("pixelated_catalog: give `neff_pixels` or `data_gw_pixelated`")
        ra_pix = np.asarray(data_gw_pixelated.ra_pix)
        neff_pixels = np.sum(ra_pix != -100., axis=1)                      # catalog.py:118
      self.neff_pixels = np.asarray(neff_pixels)
      self.N_gal = np.zeros(self.p_cat.shape[0]) if N_gal is None else np.asarray(N_gal)
      self.P_compl = self.completeness.P_compl(np.asarray(z_grids))[:, np.newaxis, :]   # catalog.py:195
      if out_file is not None:
        self.save(out_file)
    else:
      raise NotImplementedError("pixelated_catalog: building p_cat from a galaxy file (catalog.py:105-141) needs the "
                                "HEALPix/HDF5 preprocessing, which is outside the accelerated path; pass `p_cat=` or "
                                "`gal_cat_file=`")

  def save(self, fname):
    np.savez(fname, p_cat=self.p_cat, N_gal=self.N_gal, P_compl=self.P_compl, neff_pixels=self.neff_pixels,
             max_npixels=self.max_npixels)

  def p_gal(self, cosmo_lambdas, z):
    """catalog.py:197-203."""
    fR = np.atleast_3d(self.fR(cosmo_lambdas))
    p_bkg = self.p_bkg(cosmo_lambdas, np.asarray(z, dtype=np.float64))[:, np.newaxis, :]
    p_gal = fR * self.p_cat + (1. - self.P_compl) * p_bkg
    return np.where(self.p_cat != -100., p_gal, -100.)
