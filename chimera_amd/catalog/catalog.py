"""Galaxy-catalogue redshift prior (reference: CHIMERA/catalog/catalog.py).

Runtime part of the plugin: ``p_gal(cosmo, z) = fR * p_cat + (1 - P_compl) * p_bkg`` with the -100 sentinel
(catalog.py:197-203).  Inside ``hyperlikelihood`` this expression is fused into the HIP integrand kernel; the
``p_gal`` method here is the standalone (array) form of the same plugin call.
``p_cat`` comes from arrays or an ``.npz`` cache (the reference's HDF5 cache, catalog.py:96-103, needs h5py).
"""
import ctypes as C
import numpy as np
from ..utils.io import load_set, save_set
from .. import _lib
from ..utils.config import logger
from ..population.cosmo import dVcdz_at_z
from ..data import theta_src
from .completeness import dVdz_completeness


class empty_catalog(object):
  """catalog.py:19-43."""

  def __init__(self, p_bkg="dVdz"):
    self.p_cat = 0.
    self.N_gal = 0.
    self.P_compl = 0.
    if callable(p_bkg):                 # plug-in background p_bkg(cosmo, z): evaluated on the host (population/plugins.py)
      self.p_bkg = p_bkg
      self.p_bkg_is_plugin = True
    elif p_bkg == "dVdz":
      self.p_bkg = dVcdz_at_z
    else:
      raise ValueError("p_bkg must be 'dVdz' (built into the HIP path) or a callable p_bkg(cosmo, z)")
    self.max_npixels = None
    self.neff_pixels = None
    self.z_range = (0.073, 1.3)

  def p_gal(self, cosmo_lambdas, z):
    return self.p_bkg(cosmo_lambdas, z)


class pixelated_catalog(object):
  """catalog.py:51-203.

  Construct from a cache file (``gal_cat_file='...npz'`` holding max_npixels, neff_pixels, p_cat, N_gal, P_compl),
  from arrays (``p_cat=`` (E,P,Z) padded with -100, ``z_grids=`` (E,Z), ``neff_pixels=`` (E,)), or -- the reference's
  own path, catalog.py:105-141 -- from a galaxy sample (``cosmo=, z_grids=, data_gw_pixelated=, z_err=, weights=`` and
  ``data_gal=`` dict / ``fname_data_gal=`` ``.npz`` with ``z`` and the galaxies' HEALPix indices ``pix<nside>`` for every
  nside in ``data_gw_pixelated.opt_nsides``); ``p_cat`` is then computed on the GPU (``chm_pcat_compute``).
  The HEALPix indexing of galaxies itself (healpy ``ang2pix``, catalog.py:130-136) is preprocessing outside this package.
  """

  def __init__(self, completeness, gal_cat_file=None, cosmo=None, z_grids=None, fname_data_gal=None,
               data_gw_pixelated=None, z_err=1, weights=None, mask_gal=None, sumgauss="dVdz", reshuffle=False,
               out_file=None, p_cat=None, neff_pixels=None, N_gal=None, data_gal=None):
    self.completeness = completeness
    self.p_bkg = self.completeness.p_bkg
    self.fR = self.completeness.fR
    # (the built-in dVdz completeness carries z_range for the device-side fR; a plug-in completeness hands fR over per call and needs none)
    self.z_range = tuple(float(v) for v in getattr(completeness, 'z_range', (0.073, 1.3)))
    self.attr_gal_cat = ['max_npixels', 'neff_pixels']
    self.data_gal_cat = ['p_cat', 'N_gal', 'P_compl']
    if gal_cat_file is not None:
      logger.info(f"Loading gal_cat object from {gal_cat_file}")
      self._load(gal_cat_file)
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
    elif data_gal is not None or fname_data_gal is not None:
      if cosmo is None or z_grids is None or data_gw_pixelated is None:
        raise ValueError("pixelated_catalog: `cosmo`, `z_grids` and `data_gw_pixelated` are needed to compute p_cat")
      if sumgauss not in ("dVdz", "pbkg"):
        raise ValueError("sumgauss must be 'dVdz' or 'pbkg'")
      self.cosmo, self.z_grids, self.data_gw_pixelated = cosmo, np.ascontiguousarray(z_grids, dtype=np.float64), data_gw_pixelated
      self.z_err, self.sumgauss = z_err, sumgauss
      if data_gal is None:
        with np.load(fname_data_gal) as d:
          data_gal = {k: d[k] for k in d.files}
      self.data_gal = {k: np.asarray(v) for k, v in data_gal.items()}
      self.data_gal['w'] = np.asarray(weights, dtype=np.float64) if weights is not None else np.ones_like(self.data_gal['z'], dtype=np.float64)
      self.data_gal['z_err'] = self.z_err * (1. + self.data_gal['z'])                 # catalog.py:115
      self.nevents = len(self.data_gw_pixelated.dL)
      self.max_npixels = self.data_gw_pixelated.pixels_opt_nsides.shape[1]              # catalog.py:117
      self.neff_pixels = np.sum(np.asarray(self.data_gw_pixelated.ra_pix) != -100., axis=1)   # catalog.py:118
      if mask_gal is not None:
        mask_gal = np.asarray(mask_gal)
        self.data_gal = {k: v[mask_gal] for k, v in self.data_gal.items()}
      if reshuffle:
        self.data_gal['z'] = np.random.normal(self.data_gal['z'], self.data_gal['z_err'])
      for ns in np.unique(self.data_gw_pixelated.opt_nsides):
        if f"pix{ns}" not in self.data_gal:
          raise ValueError(f"pixelated_catalog: `data_gal` needs the galaxies' HEALPix indices 'pix{ns}' (computing them "
                           "needs healpy, catalog.py:130-136)")
      logger.info("Computing p_cat ...")
      self.precompute_p_cat(self.z_grids)
      if out_file is not None:
        self.save(out_file)
    else:
      raise ValueError("pixelated_catalog: pass `gal_cat_file=`, `p_cat=` or a galaxy sample (`data_gal=` / `fname_data_gal=`)")

  # -- catalog.py:143-195 ------------------------------------------------------------------------------------
  def _csr_of_event_pixels(self, zgrids):
    """Host part of `_select_galaxies_in_event_voxels` / `_compute_p_cat_event` (catalog.py:143-163): for every
    (event, pixel) the galaxies with that HEALPix index (at the event's nside) and z_grid[0] < z < z_grid[-1]."""
    E, P = self.nevents, self.max_npixels
    nsides = np.asarray(self.data_gw_pixelated.opt_nsides)
    pixels = np.asarray(self.data_gw_pixelated.pixels_opt_nsides)
    zgal = self.data_gal['z']
    order, sorted_pix = {}, {}
    for ns in np.unique(nsides):
      o = np.argsort(self.data_gal[f"pix{ns}"], kind='stable')
      order[ns], sorted_pix[ns] = o, self.data_gal[f"pix{ns}"][o]
    counts = np.zeros(E * P, dtype=np.int64)
    chunks = []
    for e in range(E):
      ns = nsides[e]
      zmin, zmax = zgrids[e, 0], zgrids[e, -1]
      for p in range(P):
        pid = pixels[e, p]
        if pid == -100:
          continue
        lo, hi = np.searchsorted(sorted_pix[ns], pid, side='left'), np.searchsorted(sorted_pix[ns], pid, side='right')
        idx = np.sort(order[ns][lo:hi])                    # keep catalogue order inside the pixel
        idx = idx[(zgal[idx] > zmin) & (zgal[idx] < zmax)]
        counts[e * P + p] = idx.size
        chunks.append(idx)
    idx = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)
    offsets = np.zeros(E * P + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return offsets, idx

  def precompute_p_cat(self, zgrids):
    """Store `p_cat`, `N_gal` and `P_compl` on the given redshift grids (catalog.py:180-195); the per-pixel sums of
    galaxy Gaussians run on the GPU."""
    from ..population._base import make_params
    zgrids = np.ascontiguousarray(zgrids, dtype=np.float64)
    E, P, Z = self.nevents, self.max_npixels, zgrids.shape[1]
    offsets, idx = self._csr_of_event_pixels(zgrids)
    gz, gs, gw = (_lib.as_f64(self.data_gal[k][idx]) for k in ('z', 'z_err', 'w'))
    d = _lib.chm_pcat_desc()
    d.E, d.P, d.Z, d.device = E, P, Z, _lib.default_device()
    d.z_grids, d.offsets = _lib.dptr(zgrids), offsets.ctypes.data_as(C.POINTER(C.c_int64))
    d.gal_z, d.gal_sig, d.gal_w = _lib.dptr(gz), _lib.dptr(gs), _lib.dptr(gw)
    wgrid = None
    if getattr(self, 'sumgauss', 'dVdz') == 'pbkg' and not isinstance(self.completeness, dVdz_completeness):
      # _sum_gaussians_pbkg (catalog.py:223-231): the Gaussians are weighted by the completeness model's own p_bkg(cosmo, z) -- evaluated
      # here on the event grids and handed to k_pcat (for the built-in dVdz completeness p_bkg IS dVc/dz, the kernel's default weight)
      wgrid = np.ascontiguousarray(np.broadcast_to(np.asarray(self.p_bkg(self.cosmo, zgrids), dtype=np.float64), (E, Z)))
      d.weight_grid = _lib.dptr(wgrid)
    p_cat = np.empty((E, P, Z))
    par = make_params(cosmo=self.cosmo)
    _lib.check(_lib.lib().chm_pcat_compute(C.byref(par), C.byref(d), _lib.dptr(p_cat)))
    pad = np.asarray(self.data_gw_pixelated.pixels_opt_nsides) == -100
    p_cat[pad] = -100.                                      # catalog.py:174-176
    self.p_cat = p_cat
    self.N_gal = (offsets[1:] - offsets[:-1]).reshape(E, P).sum(axis=1)
    self.P_compl = self.completeness.P_compl(zgrids)[:, np.newaxis, :]   # catalog.py:195

  def save(self, fname):
    """The reference's cache layout (catalog.py:96-103 with io.save_set): ``max_npixels`` / ``neff_pixels`` as attributes, ``p_cat``,
    ``N_gal``, ``P_compl`` as datasets -- HDF5 for ``.h5`` / ``.hdf5`` (interchangeable with the reference's and the Zenodo caches;
    needs h5py), ``.npz`` otherwise."""
    save_set(self, fname, self.attr_gal_cat, self.data_gal_cat)

  def _load(self, fname):
    if not str(fname).endswith(('.h5', '.hdf5')):
      with np.load(fname) as d:
        flat = 'attr/max_npixels' not in d.files and 'max_npixels' in d.files     # caches written before the attribute / dataset split
        if flat:
          for k in self.attr_gal_cat + self.data_gal_cat:
            setattr(self, k, d[k])
    else:
      flat = False
    if not flat:
      load_set(self, fname, self.attr_gal_cat, self.data_gal_cat)
    self.p_cat = np.ascontiguousarray(self.p_cat, dtype=np.float64)
    self.P_compl = np.ascontiguousarray(self.P_compl, dtype=np.float64)
    self.neff_pixels = np.asarray(self.neff_pixels)
    self.max_npixels = int(self.max_npixels)

  def p_gal(self, cosmo_lambdas, z):
    """catalog.py:197-203."""
    fR = np.atleast_3d(self.fR(cosmo_lambdas))
    p_bkg = self.p_bkg(cosmo_lambdas, np.asarray(z, dtype=np.float64))[:, np.newaxis, :]
    p_gal = fR * self.p_cat + (1. - self.P_compl) * p_bkg
    return np.where(self.p_cat != -100., p_gal, -100.)
