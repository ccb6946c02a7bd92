"""Mass models (reference: CHIMERA/population/mass.py): ``tpl``, ``bpl``, ``plp`` with the reference's parameters and
defaults.  ``pl2p`` and ``pls`` are broken / unfinished in the reference (mass.py:307-314, 193-233) and are not provided.
Normalisation tables (mass.py:45-52) and pdfs are evaluated on the GPU.
"""
import numpy as np
from .. import _lib
from ..data import theta_src
from ._base import base_struct, make_params, model_eval, model_tables


class base_mass_paired_struct(base_struct):
  default = {'m_low': 5.1, 'm_high': 87., 'grid_res': 1000}
  name = 'base_mass_paired_struct'

  def _tab(self):
    if self._tables is None:
      self._tables = model_tables(make_params(mass=self))
    return self._tables

  @property
  def m_grid(self):
    return self._tab()['m_grid']

  @property
  def cdf_m2_conditioned(self):
    return self._tab()['cdf_m2_conditioned']

  @property
  def norm_p_m1(self):
    return self._tab()['norm_p_m1']


class tpl(base_mass_paired_struct):
  """mass.py:56-83."""
  default = {**base_mass_paired_struct.default, 'alpha': 2.5, 'beta': 1.1}
  name = 'truncated_power_law'

  def _pack(self):
    return dict(model=0, vec=[self.m_low, self.m_high, self.alpha, self.beta], grid_res=self.grid_res)


class bpl(base_mass_paired_struct):
  """mass.py:85-115."""
  default = {**base_mass_paired_struct.default, 'alpha_1': 1.6, 'alpha_2': 5.6, 'beta': 1.1, 'delta_m': 4.8,
             'break_fraction': 0.43}
  name = 'broken_power_law'

  def _pack(self):
    return dict(model=1, vec=[self.m_low, self.m_high, self.alpha_1, self.alpha_2, self.beta, self.delta_m,
                              self.break_fraction], grid_res=self.grid_res)


class plp(base_mass_paired_struct):
  """mass.py:117-149."""
  default = {**base_mass_paired_struct.default, 'lambda_peak': 0.039, 'alpha': 3.4, 'beta': 1.1, 'delta_m': 4.8,
             'mu_g': 34., 'sigma_g': 3.6}
  name = 'power_law_plus_peak'

  def _pack(self):
    return dict(model=2, vec=[self.m_low, self.m_high, self.lambda_peak, self.alpha, self.beta, self.delta_m,
                              self.mu_g, self.sigma_g], grid_res=self.grid_res)


def primary_mass_pdf_notnorm(mass, m):
  """mass.py:285-305."""
  return model_eval(make_params(mass=mass), _lib.F_PRIMARY, m)


def secondary_mass_conditioned_pdf_notnorm(mass, m2, m1):
  """mass.py:320-328."""
  return model_eval(make_params(mass=mass), _lib.F_SECONDARY, m2, m1)


def tpl_notnorm(m, alpha, m_low, m_high):
  """mass.py:240-245: m**alpha inside [m_low, m_high], 0 outside."""
  return model_eval(make_params(mass=tpl(alpha=-alpha, m_low=m_low, m_high=m_high)), _lib.F_PRIMARY, m)


def tpl_cdf(alpha, m_low, m):
  """mass.py:247-252."""
  return model_eval(make_params(mass=tpl(alpha=-alpha, m_low=m_low, m_high=max(2. * m_low, m_low + 1.))), _lib.F_TPL_CDF, m)


def gaussian(x, mu, sigma):
  """mass.py:267-269."""
  return model_eval(make_params(mass=plp(mu_g=mu, sigma_g=sigma)), _lib.F_GAUSSIAN, x)


def truncated_gaussian(x, mu, sigma, x_min, x_max):
  """mass.py:271-279."""
  return model_eval(make_params(mass=plp(mu_g=mu, sigma_g=sigma, m_low=x_min, m_high=x_max)), _lib.F_TRUNC_GAUSSIAN, x)


def smoothing(m, delta_m, m_low):
  """mass.py:255-264."""
  return model_eval(make_params(mass=plp(delta_m=delta_m, m_low=m_low)), _lib.F_SMOOTHING, m)


def p_m1m2(mass, m1, m2=None):
  """mass.py:334-345 (array and theta_src overloads)."""
  if isinstance(m1, theta_src):
    m1, m2 = m1.m1src, m1.m2src
  if not hasattr(mass, '_pack'):                 # plug-in mass model (population/plugins.py): its own host function
    return np.asarray(mass.p_m1m2(np.asarray(m1, dtype=np.float64), np.asarray(m2, dtype=np.float64)), dtype=np.float64)
  return model_eval(make_params(mass=mass), _lib.F_PM1M2, m1, m2)


def pdf_joint_and_marg(mass, res=(5000, 2500)):
  """mass.py:351-362 (the reference's plotting helper): the joint pdf on a (res[1], res[0]) mesh of [m_low, m_high]^2 and its two marginals, each
  normalised by its own trapezoid.  The joint pdf comes from the device (``chm_model_eval``: the arithmetic of the hot path's ``p_m1m2``), the four
  trapezoids from ``utils.math.trapz`` (``chm_trapz``)."""
  from ..utils.math import trapz
  m1 = np.linspace(mass.m_low, mass.m_high, int(res[0]))
  m2 = np.linspace(mass.m_low, mass.m_high, int(res[1]))
  m1mesh, m2mesh = np.meshgrid(m1, m2)
  p_joint = np.asarray(p_m1m2(mass, m1mesh, m2mesh)).reshape(m1mesh.shape)
  p1_marg = trapz(p_joint, x=m2, axis=0)
  p1_marg = p1_marg / trapz(p1_marg, x=m1)
  p2_marg = trapz(p_joint, x=m1, axis=1)
  p2_marg = p2_marg / trapz(p2_marg, x=m2)
  return {'m1': m1, 'm2': m2, 'm1mesh': m1mesh, 'm2mesh': m2mesh, 'p_joint': p_joint, 'p_m1_marg': p1_marg, 'p_m2_marg': p2_marg}
