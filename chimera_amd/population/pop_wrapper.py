"""Population glue (reference: CHIMERA/population/pop_wrapper.py)."""
from numbers import Number
import numpy as np
from .. import _lib
from ..data import theta_src, theta_pe_det, theta_inj_det
from ..catalog.catalog import empty_catalog
from .cosmo import dVcdz_at_z, z_from_dGW, ddLdz_at_z
from .mass import p_m1m2
from .rate import merger_rate
from ._base import make_params


class population(object):
  """pop_wrapper.py:14-64: the bundle (cosmo, mass, rate, R0, gal_cat, Tobs, scale_free); ``update(**lambdas)``
  returns a new bundle with every model updated from the same keyword set (unknown keys ignored)."""

  def __init__(self, cosmo, mass, rate, R0=1., gal_cat=None, Tobs=1, scale_free=True):
    self.cosmo = cosmo
    self.mass = mass
    self.rate = rate
    self.R0 = R0
    if gal_cat is None:
      gal_cat = empty_catalog(p_bkg='dVdz')
    self.gal_cat = gal_cat
    self.Tobs = Tobs
    self.scale_free = scale_free

  def __repr__(self):
    return (f"cosmo = {self.cosmo},\nmass = {self.mass},\nrate = {self.rate},\nR0 = {self.R0},\n"
            f"galcat_obj = {self.gal_cat},\nTobs = {self.Tobs},\nscale_free = {self.scale_free}")

  def update(self, **hyper_lambdas):
    return self.__class__(self.cosmo.update(**hyper_lambdas), self.mass.update(**hyper_lambdas),
                          self.rate.update(**hyper_lambdas), hyper_lambdas.get('R0', self.R0),
                          self.gal_cat, self.Tobs, self.scale_free)

  def to_params(self):
    """The ``chm_params`` of this draw (include/chimera_hip.h)."""
    gc = self.gal_cat
    return make_params(self.cosmo, self.mass, self.rate, self.R0, self.Tobs, self.scale_free,
                       has_catalog=not isinstance(gc, empty_catalog),
                       z_range=getattr(gc, 'z_range', (0.073, 1.3)))


def theta_det2src(cosmo_lambdas, theta_det, include_original_distances=False):
  """pop_wrapper.py:67-75."""
  z = z_from_dGW(cosmo_lambdas, theta_det.dL)
  m1s, m2s = theta_det.m1det / (1. + z), theta_det.m2det / (1. + z)
  if include_original_distances:
    return theta_src(m1src=m1s, m2src=m2s, z=z, original_distances=theta_det.dL)
  return theta_src(m1src=m1s, m2src=m2s, z=z)


def get_theta_src_and_weights(pop_lambdas, theta_det):
  """pop_wrapper.py:77-80."""
  th_src = theta_det2src(pop_lambdas.cosmo, theta_det)
  with np.errstate(all='ignore'):
    weights = p_m1m2(pop_lambdas.mass, th_src) / theta_det.pe_prior
  return th_src, weights


def p_cbc(pop_lambdas, z):
  """pop_wrapper.py:82-90."""
  z = np.asarray(z, dtype=np.float64)
  p_gal = pop_lambdas.gal_cat.p_gal(pop_lambdas.cosmo, z)
  p_rate = merger_rate(pop_lambdas.rate, z) / (1 + z)
  if np.ndim(p_gal) > np.ndim(p_rate):
    return np.where(p_gal != -100, p_gal * p_rate[:, None, :], -100)
  return p_gal * p_rate


def pop_rate_det(pop_lambdas, th):
  """pop_wrapper.py:92-121 (the three plum overloads: theta_pe_det, theta_inj_det, theta_src)."""
  with np.errstate(all='ignore'):
    if isinstance(th, theta_pe_det):
      src = theta_det2src(pop_lambdas.cosmo, th)
      p_z = p_cbc(pop_lambdas, src.z)
    else:
      src = th if isinstance(th, theta_src) else theta_det2src(pop_lambdas.cosmo, th, include_original_distances=True)
      p_z = pop_lambdas.gal_cat.p_bkg(pop_lambdas.cosmo, src)
      p_z = p_z * (merger_rate(pop_lambdas.rate, src) / (1. + src.z))
    dNdtheta = pop_lambdas.R0 * p_m1m2(pop_lambdas.mass, src) * p_z
    jacobian = np.abs(ddLdz_at_z(pop_lambdas.cosmo, src)) * (1. + src.z)**2
    return dNdtheta / jacobian


def N_cbc_1yr(pop_lambdas):
  """pop_wrapper.py:123-129."""
  zz = np.linspace(0.001, pop_lambdas.cosmo.z_max, 10_000)
  dN_dz = merger_rate(pop_lambdas.rate, zz) / (1. + zz) * pop_lambdas.gal_cat.p_bkg(pop_lambdas.cosmo, zz)
  dN_dz = dN_dz * pop_lambdas.R0
  return 0.5 * np.sum(np.diff(zz) * (dN_dz[1:] + dN_dz[:-1]))


def _linspace_rows(start, stop, num):
  step = np.arange(num - 1, dtype=np.float64) / np.float64(num - 1)
  out = start[:, None] * (1. - step) + stop[:, None] * step
  return np.concatenate([out, stop[:, None]], axis=1)


def compute_z_grids(cosmo, theta_det, cosmo_prior=None, z_int_res=300, z_conf_range=None):
  """pop_wrapper.py:133-208: per-event redshift grids covering the samples for every cosmology in the prior.
  The two dL -> z inversions use the 10 000-point tables, built and interpolated on the GPU."""
  events_dL = np.asarray(theta_det.dL, dtype=np.float64)
  if isinstance(z_conf_range, list):
    dL_min, dL_max = np.percentile(events_dL, z_conf_range, axis=1)
  elif isinstance(z_conf_range, Number):
    mu, sig = np.mean(events_dL, axis=1), np.std(events_dL, axis=1)
    dL_min, dL_max = mu - z_conf_range * sig, mu + z_conf_range * sig
  else:
    dL_max = np.max(events_dL, axis=1) * 2
    dL_min = np.min(events_dL, axis=1) * 0.5
    dL_min = np.where(dL_min < 1.e-8, 1.e-8, dL_min)
  cp = {k: [v, v] for k, v in cosmo.as_dict.items()}
  if cosmo_prior is not None:
    cp.update(cosmo_prior)
  base = ['H0', 'Om0', 'Ok0', 'Or0', 'w0', 'wa']
  if not hasattr(cosmo, '_pack'):              # plug-in cosmology: whatever parameters it has (the caller orders the prior low -> nearest)
    base = [k for k in cosmo.keys if k not in ('z_max', 'z_grid_res')]
  lc_low = {k: cp[k][0] for k in base}
  lc_high = {k: cp[k][1] for k in base}
  if hasattr(cosmo, '_pack') and cosmo.name != 'flrw':
    lc_low.update(Xi0=cp['Xi0'][1], n=cp['n'][1])
    lc_high.update(Xi0=cp['Xi0'][0], n=cp['n'][1])
  cosmo1 = cosmo.update(**lc_low, z_grid_res=10_000)
  cosmo2 = cosmo.update(**lc_high, z_grid_res=10_000)
  z_min = z_from_dGW(cosmo1, dL_min)
  z_max = z_from_dGW(cosmo2, dL_max)
  return _linspace_rows(z_min, z_max, z_int_res)
