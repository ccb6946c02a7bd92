"""Cosmological models (reference: CHIMERA/population/cosmo.py).

``flrw`` / ``mg_flrw`` keep the reference's parameters, defaults and ``update`` semantics.  The interpolation table
(cosmo.py:43-46) and every distance function are evaluated on the GPU (k_tables / k_model_eval in
chimera_amd/csrc); ``z_grid_interp`` and ``integral_invE_interp`` are fetched from the device on first access.
"""
import numpy as np
from .. import _lib
from ..data import theta_src
from ._base import base_struct, make_params, model_eval, model_tables


class base_cosmology_struct(base_struct):
  default = {'z_max': 10., 'z_grid_res': 1000}
  name = 'base_cosmology_struct'

  def _tab(self):
    if self._tables is None:
      self._tables = model_tables(make_params(cosmo=self))
    return self._tables

  @property
  def z_grid_interp(self):
    return self._tab()['z_grid_interp']

  @property
  def integral_invE_interp(self):
    return self._tab()['integral_invE_interp']


class flrw(base_cosmology_struct):
  """cosmo.py:50-84.  H0 [km/s/Mpc]; distances are in Gpc (dH = 299792.458e-3 / H0)."""
  name = 'flrw'
  default = {**base_cosmology_struct.default, 'H0': 70., 'Om0': 0.25, 'Ok0': 0., 'Or0': 0., 'w0': -1., 'wa': 0.,
             'z_max': 10., 'z_grid_res': 1500}

  @property
  def Ode0(self):
    return 1.0 - self.Om0 - self.Or0 - self.Ok0

  @property
  def dH(self):
    return 299792.458e-3 / self.H0

  def _pack(self):
    return dict(model=0, vec=[self.H0, self.Om0, self.Ok0, self.Or0, self.w0, self.wa, 1., 0.],
                z_max=self.z_max, z_grid_res=self.z_grid_res)


class mg_flrw(flrw):
  """cosmo.py:86-115: modified GW propagation, Xi(z) = Xi0 + (1 - Xi0)/(1+z)^n."""
  name = 'mg_flrw'
  default = {**flrw.default, 'Xi0': 1., 'n': 0.}

  def _pack(self):
    return dict(model=1, vec=[self.H0, self.Om0, self.Ok0, self.Or0, self.w0, self.wa, self.Xi0, self.n],
                z_max=self.z_max, z_grid_res=self.z_grid_res)


class plugin_cosmology(base_struct):
  """Base of a user-written cosmology (the reference's open extension point: a new struct + plum overloads of the distance functions,
  cosmo.py:122-264).  It has no ``_pack``: nothing of it runs inside a kernel; the host evaluates what the path needs of a cosmology
  -- the (dL, z) table of ``z_from_dGW``, the Jacobian and ``p_bkg`` -- and hands the values over per draw (chm_tab, plugins.py).
  A subclass lists its parameters in ``default`` (with ``z_max`` and ``z_grid_res``) and provides, in Gpc:

    dL_at_z(z)                       GW luminosity distance                           (cosmo.py:205-210, 237-243)
    ddLdz_at_z(z, distances=None)    its z-derivative; `distances`: original dL       (cosmo.py:212-221, 245-257)
    dVcdz_at_z(z, distances=None)    differential comoving volume                     (cosmo.py:188-197)
    Vc_at_z(z, distances=None)       comoving volume                                  (cosmo.py:166-186)
  """
  default = {'z_max': 10., 'z_grid_res': 1500}
  name = 'plugin_cosmology'

  @property
  def z_grid_interp(self):
    """cosmo.py:43-46: [0] U logspace(-10, log10 z_max, z_grid_res - 1)."""
    return np.concatenate([[0.], np.logspace(-10., np.log10(self.z_max), int(self.z_grid_res) - 1)])


def is_plugin(cosmo):
  return cosmo is not None and not hasattr(cosmo, '_pack')


def _zd(z, distances):
  """plum-dispatch overloads on theta_src (cosmo.py:269-279)."""
  if isinstance(z, theta_src):
    return z.z, z.original_distances
  return z, distances


def _ev(cosmo, func, a, b=None):
  return model_eval(make_params(cosmo=cosmo), func, a, b)


def E_at_z(cosmo, z):
  """cosmo.py:122-130."""
  return _ev(cosmo, _lib.F_E, z)


def int_invE_at_z(cosmo, z):
  """cosmo.py:132-133."""
  return _ev(cosmo, _lib.F_INT_INVE, z)


def dCr_at_z(cosmo, z):
  """cosmo.py:135-139."""
  return _ev(cosmo, _lib.F_DCR, z)


def dCt_at_z(cosmo, z):
  """cosmo.py:141-153."""
  return _ev(cosmo, _lib.F_DCT, z)


def _dL2dCt(cosmo, distances, z):
  """cosmo.py:201-203, 230-235."""
  distances, z = np.asarray(distances, dtype=np.float64), np.asarray(z, dtype=np.float64)
  if isinstance(cosmo, mg_flrw):
    return (distances / Xi_at_z(cosmo, z)) / (1. + z)
  return distances / (1. + z)


def dA_at_z(cosmo, z, distances=None):
  """cosmo.py:155-162."""
  z = np.asarray(z, dtype=np.float64)
  dCt = _dL2dCt(cosmo, distances, z) if distances is not None else dCt_at_z(cosmo, z)
  return dCt / (1. + z)


def Vc_at_z(cosmo, z, distances=None):
  """cosmo.py:166-186, 273-275."""
  z, distances = _zd(z, distances)
  if is_plugin(cosmo):
    return np.asarray(cosmo.Vc_at_z(z, distances) if distances is not None else cosmo.Vc_at_z(z), dtype=np.float64)
  return _ev(cosmo, _lib.F_VC, z, distances)


def dVcdz_at_z(cosmo, z, distances=None):
  """cosmo.py:188-197, 269-271."""
  z, distances = _zd(z, distances)
  if is_plugin(cosmo):
    return np.asarray(cosmo.dVcdz_at_z(z, distances) if distances is not None else cosmo.dVcdz_at_z(z), dtype=np.float64)
  return _ev(cosmo, _lib.F_DVCDZ, z, distances)


def dL_at_z(cosmo, z):
  """cosmo.py:205-210, 237-243."""
  if is_plugin(cosmo):
    return np.asarray(cosmo.dL_at_z(np.asarray(z, dtype=np.float64)), dtype=np.float64)
  return _ev(cosmo, _lib.F_DL, z)


def ddLdz_at_z(cosmo, z, distances=None):
  """cosmo.py:212-221, 245-257, 277-279."""
  z, distances = _zd(z, distances)
  if is_plugin(cosmo):
    return np.asarray(cosmo.ddLdz_at_z(z, distances) if distances is not None else cosmo.ddLdz_at_z(z), dtype=np.float64)
  return _ev(cosmo, _lib.F_DDLDZ, z, distances)


def Xi_at_z(cosmo, z):
  """cosmo.py:225-228 (mg_flrw only)."""
  if not isinstance(cosmo, mg_flrw):
    raise TypeError("Xi_at_z is defined for mg_flrw only")
  return _ev(cosmo, _lib.F_XI, z)


def z_from_dGW(cosmo, dGWs):
  """cosmo.py:260-264."""
  if is_plugin(cosmo):                       # jnp.interp(dGWs, dL_at_z(z_grid_interp), z_grid_interp) on the host, slope form as the device's
    zt = np.asarray(cosmo.z_grid_interp, dtype=np.float64)
    return np.interp(np.asarray(dGWs, dtype=np.float64), dL_at_z(cosmo, zt), zt)
  return _ev(cosmo, _lib.F_Z_FROM_DGW, dGWs)
