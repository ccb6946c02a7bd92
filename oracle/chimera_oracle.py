"""
oracle/chimera_oracle.py -- CPU restatement (NumPy/SciPy) of CHIMERA's hyper-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``chimera_amd/`` may import this module.  It is the checker
for the HIP path (tests/, ``__graft_entry__.smoke()``) and the ``cpu_baseline`` leg of ``bench.py``.

PARITY UNPINNED: the reference (CHIMERA v2.0.0, pure Python on ``jax>=0.5`` + equinox + plum + numba,
``pyproject.toml:14-56``) has no tests, golden vectors or fixtures for this path, and it cannot be imported
in this image (jax, equinox, plum, numba, healpy, h5py are absent; no network).  The oracle is therefore a
line-by-line restatement of the reference *source*, each function citing the file:line it follows
(paths relative to the reference root, ``CHIMERA/...``), pinned by independent checks in
``tests/test_oracle_pins.py`` (scipy.stats.gaussian_kde, scipy.integrate.quad, analytic identities).

Third-party semantics restated here (the reference executes through ``jax.numpy``; no lock file, lower bound
``jax>=0.5``):  ``jnp.linspace`` (start*(1-i/div) + stop*(i/div), last point = stop), ``jnp.interp``
(``fp[i-1] + (delta/dx)*df`` with ``searchsorted(side='right')`` clipped to [1, n-1]), ``jnp.trapezoid``
(``0.5*sum(dx*(y[1:]+y[:-1]))``), ``jnp.std`` (two-pass, ddof=0), ``jnp.nan_to_num`` (-inf -> float64 min),
``jnp.logspace`` (``10**linspace``).  These differ from NumPy's own functions by <= a few ulp.
"""
import numpy as np
from math import erf as _erf_scalar
import math as _math

FLT_MIN_NEG = -np.finfo(np.float64).max     # jnp.nan_to_num default for -inf


# ----------------------------------------------------------------------------------------------------------
# jax.numpy semantics
# ----------------------------------------------------------------------------------------------------------

def jnp_linspace(start, stop, num):
  """jnp.linspace(start, stop, num) along a new LAST axis for array bounds (scalar bounds -> 1-D)."""
  start = np.asarray(start, dtype=np.float64)
  stop = np.asarray(stop, dtype=np.float64)
  div = num - 1
  step = np.arange(div, dtype=np.float64) / np.float64(div)
  out = start[..., None] * (1. - step) + stop[..., None] * step
  return np.concatenate([out, np.broadcast_to(stop[..., None], out.shape[:-1] + (1,))], axis=-1)


def jnp_logspace(a, b, num):
  return np.power(10., jnp_linspace(a, b, num))


def jnp_searchsorted_right(xp, x):
  """jnp.searchsorted(xp, x, side='right') (default method 'scan').  On a sorted, NaN-free table this is np.searchsorted.  On a
  NON-monotonic one (the dL table of an unphysical cosmology) the answer of a binary search depends on its probe sequence, so
  jax's own is followed: ceil(log2(n+1)) steps of ``mid = (low+high)//2; go_left = x < xp[mid] (NaN sorts last);
  high = mid if go_left else low = mid``, result ``high`` (jax/_src/numpy/lax_numpy.py:_searchsorted_via_scan).  np.searchsorted
  itself is not usable there: it narrows the window from the previous key when the keys are increasing."""
  xp = np.asarray(xp, dtype=np.float64)
  x = np.asarray(x, dtype=np.float64)
  with np.errstate(all='ignore'):
    if not (np.any(np.isnan(xp)) or np.any(np.diff(xp) < 0)):
      return np.searchsorted(xp, x, side='right')
    n = len(xp)
    lo = np.zeros(x.shape, dtype=np.int64)
    hi = np.full(x.shape, n, dtype=np.int64)
    for _ in range(int(np.ceil(np.log2(n + 1)))):
      mid = (lo + hi) // 2
      v = xp[mid]
      go_left = (x < v) | (np.isnan(v) & ~np.isnan(x))
      hi = np.where(go_left, mid, hi)
      lo = np.where(go_left, lo, mid)
    return hi


def jnp_interp(x, xp, fp, left=None, right=None):
  x = np.asarray(x, dtype=np.float64)
  i = np.clip(jnp_searchsorted_right(xp, x), 1, len(xp) - 1)
  df = fp[i] - fp[i - 1]
  dx = xp[i] - xp[i - 1]
  delta = x - xp[i - 1]
  epsilon = np.spacing(np.finfo(np.float64).eps)
  dx0 = np.abs(dx) <= epsilon
  with np.errstate(all='ignore'):
    f = np.where(dx0, fp[i - 1], fp[i - 1] + (delta / np.where(dx0, 1., dx)) * df)
  f = np.where(x < xp[0], fp[0] if left is None else left, f)
  f = np.where(x > xp[-1], fp[-1] if right is None else right, f)
  return f


def trapz(y, x, axis=-1):
  """CHIMERA/utils/math.py:10-16 (jnp.trapezoid)."""
  y = np.moveaxis(np.asarray(y), axis, -1)
  x = np.moveaxis(np.asarray(x), axis, -1) if np.ndim(x) == np.ndim(y) else np.asarray(x)
  dx = np.diff(x, axis=-1)
  return 0.5 * (dx * (y[..., 1:] + y[..., :-1])).sum(-1)


def cumtrapz(y, x):
  """CHIMERA/utils/math.py:22-26."""
  dx = np.diff(x)
  res = np.cumsum(0.5 * (y[:-1] + y[1:]) * dx)
  return np.concatenate([np.array([0.]), res])


def nan_to_num_neginf(x):
  """jnp.nan_to_num(x, nan=-inf): NaN -> -inf, +inf -> max, -inf -> -max (likelihood.py:297,330)."""
  x = np.array(x, dtype=np.float64, copy=True)
  neg = np.isneginf(x)
  pos = np.isposinf(x)
  nan = np.isnan(x)
  x[neg] = FLT_MIN_NEG
  x[pos] = -FLT_MIN_NEG
  x[nan] = -np.inf
  return x


# ----------------------------------------------------------------------------------------------------------
# models: cosmology  (CHIMERA/population/cosmo.py)
# ----------------------------------------------------------------------------------------------------------

class _Params(object):
  default = {}
  name = 'base'

  def __init__(self, **kwargs):
    self.keys = list(self.default.keys())
    for k in self.keys:
      setattr(self, k, kwargs.get(k, self.default[k]))
    self._setup()

  def _setup(self):
    pass

  @property
  def as_dict(self):
    return {k: getattr(self, k) for k in self.keys}

  def update(self, **kwargs):
    """cosmo.py:33-40, mass.py:35-42, rate.py:23-30 -- unknown keys are ignored."""
    upd = {k: v for k, v in kwargs.items() if k in self.keys}
    if upd == {}:
      return self
    fid = self.as_dict
    fid.update(upd)
    return self.__class__(**fid)


class flrw(_Params):
  """cosmo.py:50-84."""
  name = 'flrw'
  default = {'z_max': 10., 'z_grid_res': 1500, 'H0': 70., 'Om0': 0.25, 'Ok0': 0., 'Or0': 0., 'w0': -1., 'wa': 0.}

  @property
  def Ode0(self):
    return 1.0 - self.Om0 - self.Or0 - self.Ok0

  @property
  def dH(self):
    return 299792.458e-3 / self.H0

  def _setup(self):
    """setup_interp, cosmo.py:43-46."""
    self.z_grid_interp = np.concatenate([np.array([0.]), jnp_logspace(-10., np.log10(self.z_max), self.z_grid_res - 1)])
    Ez = E_at_z(self, self.z_grid_interp)
    self.integral_invE_interp = cumtrapz(1. / Ez, self.z_grid_interp)


class mg_flrw(flrw):
  """cosmo.py:86-115."""
  name = 'mg_flrw'
  default = {**flrw.default, 'Xi0': 1., 'n': 0.}


def E_at_z(cosmo, z):
  """cosmo.py:122-130."""
  z = np.asarray(z, dtype=np.float64)
  w_z = cosmo.w0 + cosmo.wa * z / (1 + z)
  return np.sqrt(cosmo.Om0 * (1. + z)**3 + cosmo.Or0 * (1. + z)**4 + cosmo.Ok0 * (1. + z)**2 +
                 cosmo.Ode0 * (1. + z)**(3. * (1. + w_z)))


def int_invE_at_z(cosmo, z):
  """cosmo.py:132-133."""
  return jnp_interp(z, cosmo.z_grid_interp, cosmo.integral_invE_interp)


def dCr_at_z(cosmo, z):
  """cosmo.py:135-139."""
  return cosmo.dH * int_invE_at_z(cosmo, z)


def dCt_at_z(cosmo, z):
  """cosmo.py:141-153."""
  dCr = dCr_at_z(cosmo, z)
  sqrtOk0 = np.sqrt(np.abs(cosmo.Ok0 + 1.e-10))
  dH = cosmo.dH
  if cosmo.Ok0 == 0.0:
    return dCr
  if cosmo.Ok0 > 0.0:
    return (dH / sqrtOk0) * np.sinh(sqrtOk0 * dCr / dH)
  return (dH / sqrtOk0) * np.sin(sqrtOk0 * dCr / dH)


def Xi_at_z(cosmo, z):
  """cosmo.py:225-228."""
  return cosmo.Xi0 + (1. - cosmo.Xi0) / ((1. + z)**cosmo.n)


def _dL2dCt(cosmo, distances, z):
  """cosmo.py:201-203 (flrw), 230-235 (mg_flrw)."""
  if isinstance(cosmo, mg_flrw):
    return (distances / Xi_at_z(cosmo, z)) / (1. + z)
  return distances / (1. + z)


def _dCt(cosmo, z, distances):
  return _dL2dCt(cosmo, distances, z) if distances is not None else dCt_at_z(cosmo, z)


def Vc_at_z(cosmo, z, distances=None):
  """cosmo.py:166-186."""
  z = np.asarray(z, dtype=np.float64)
  dCt = _dCt(cosmo, z, distances)
  regOk0 = cosmo.Ok0 + 1e-10
  sqrtOk0 = np.sqrt(np.abs(regOk0))
  dH = cosmo.dH
  if cosmo.Ok0 == 0.0:
    return 4. * np.pi * dCt**3 / 3.
  if cosmo.Ok0 > 0.0:
    return (4. * np.pi * dH**3 / (2. * regOk0)) * ((dCt / dH) * np.sqrt(1 + regOk0 * dCt**2 / dH**2)
                                                    - np.arcsinh(sqrtOk0 * dCt / dH) / sqrtOk0)
  with np.errstate(all='ignore'):
    return (4. * np.pi * dH**3 / (2. * regOk0)) * ((dCt / dH) * np.sqrt(1 + regOk0 * dCt**2 / dH**2)
                                                    - np.arcsin(sqrtOk0 * dCt / dH) / sqrtOk0)


def dVcdz_at_z(cosmo, z, distances=None):
  """cosmo.py:188-197."""
  z = np.asarray(z, dtype=np.float64)
  dCt = _dCt(cosmo, z, distances)
  return 4 * np.pi * cosmo.dH * dCt**2 / E_at_z(cosmo, z)


def dL_at_z(cosmo, z):
  """cosmo.py:205-210 (flrw), 237-243 (mg_flrw)."""
  z = np.asarray(z, dtype=np.float64)
  dL = dCt_at_z(cosmo, z) * (1. + z)
  if isinstance(cosmo, mg_flrw):
    return dL * Xi_at_z(cosmo, z)
  return dL


def ddLdz_at_z(cosmo, z, distances=None):
  """cosmo.py:212-221 (flrw), 245-257 (mg_flrw)."""
  z = np.asarray(z, dtype=np.float64)
  dCt = _dCt(cosmo, z, distances)
  Ez = E_at_z(cosmo, z)
  ddLflrw = dCt + (cosmo.dH / Ez) * (1. + z)
  if isinstance(cosmo, mg_flrw):
    dLflrw = dCt * (1. + z)
    Xiz = Xi_at_z(cosmo, z)
    dXiz = cosmo.n * (cosmo.Xi0 - 1.) / ((1. + z)**(cosmo.n + 1))
    return ddLflrw * Xiz + dLflrw * dXiz
  return ddLflrw


def z_from_dGW(cosmo, dGWs):
  """cosmo.py:260-264."""
  dGW_values = dL_at_z(cosmo, cosmo.z_grid_interp)
  return jnp_interp(dGWs, dGW_values, cosmo.z_grid_interp)


# ----------------------------------------------------------------------------------------------------------
# models: mass  (CHIMERA/population/mass.py)
# ----------------------------------------------------------------------------------------------------------

class _mass_base(_Params):
  default = {'m_low': 5.1, 'm_high': 87., 'grid_res': 1000}

  def _setup(self):
    """get_normalizations, mass.py:45-52."""
    self.m_grid = jnp_logspace(np.log10(self.m_low), np.log10(self.m_high), self.grid_res)
    # The end nodes 10**log10(m_low), 10**log10(m_high) are tested against `m_low <= m <= m_high` by tpl_notnorm (mass.py:240-245):
    # whether the first / last trapezoid node counts (an O(1e-3) effect on norm_p_m1 / cdf_m2) hangs on the last bit of the
    # platform's log10 and pow.  NumPy's AVX-512 loops and the C library disagree there for some masses (e.g. m_low = 3.760945123513589:
    # 3.7609451235135896 vs 3.7609451235135887), as XLA's CPU and GPU back-ends may.  The oracle pins the two nodes to the C
    # library's values (math.pow / math.log10 = glibc), which the C oracle and the product's host code use as well, so that all
    # three take the same branch on every host; the defaults (5.1, 87) are unaffected.
    self.m_grid[0] = _math.pow(10., _math.log10(self.m_low))
    self.m_grid[-1] = _math.pow(10., _math.log10(self.m_high))
    p_values = secondary_mass_conditioned_pdf_notnorm(self, self.m_grid, self.m_high)
    self.cdf_m2_conditioned = cumtrapz(p_values, self.m_grid)
    self.norm_p_m1 = trapz(primary_mass_pdf_notnorm(self, self.m_grid), self.m_grid)


class tpl(_mass_base):
  name = 'truncated_power_law'
  default = {**_mass_base.default, 'alpha': 2.5, 'beta': 1.1}


class bpl(_mass_base):
  name = 'broken_power_law'
  default = {**_mass_base.default, 'alpha_1': 1.6, 'alpha_2': 5.6, 'beta': 1.1, 'delta_m': 4.8, 'break_fraction': 0.43}


class plp(_mass_base):
  name = 'power_law_plus_peak'
  default = {**_mass_base.default, 'lambda_peak': 0.039, 'alpha': 3.4, 'beta': 1.1, 'delta_m': 4.8, 'mu_g': 34., 'sigma_g': 3.6}


def tpl_notnorm(m, alpha, m_low, m_high):
  """mass.py:240-245."""
  with np.errstate(all='ignore'):
    return np.where((m_low <= m) & (m <= m_high), np.power(m, alpha), 0.)


def tpl_cdf(alpha, m_low, m):
  """mass.py:247-252 (alpha == -1 branch sign as written)."""
  if alpha == -1:
    return np.log(m_low) - np.log(m)
  return (m**(1 + alpha) - m_low**(1 + alpha)) / (1 + alpha)


def smoothing(m, delta_m, m_low):
  """mass.py:255-264."""
  eps = 1.e-99
  m = np.asarray(m, dtype=np.float64)
  with np.errstate(all='ignore'):
    mid = -np.logaddexp(0.0, (delta_m / (m - m_low + eps) + delta_m / (m - m_low - delta_m + eps)))
    log_s = np.where(m < m_low, -np.inf, np.where(m > (m_low + delta_m), 0.0, mid))
    return np.exp(log_s)


def gaussian(x, mu, sigma):
  """mass.py:267-269."""
  log_G = -0.5 * np.log(2 * np.pi) - np.log(sigma) - (x - mu)**2 / (2. * sigma**2)
  return np.exp(log_G)


def truncated_gaussian(x, mu, sigma, x_min, x_max):
  """mass.py:271-279."""
  max_point = (x_max - mu) / (sigma * np.sqrt(2.))
  min_point = (x_min - mu) / (sigma * np.sqrt(2.))
  norm = 0.5 * _erf_scalar(max_point) - 0.5 * _erf_scalar(min_point)
  return np.where((x_min <= x) & (x <= x_max), gaussian(x, mu, sigma) / norm, 0.)


def primary_mass_pdf_notnorm(mass, m):
  """mass.py:285-305."""
  m = np.asarray(m, dtype=np.float64)
  if isinstance(mass, tpl):
    return tpl_notnorm(m, -mass.alpha, mass.m_low, mass.m_high)
  if isinstance(mass, bpl):
    m_break = mass.m_low + mass.break_fraction * (mass.m_high - mass.m_low)
    pl1_m_break = tpl_notnorm(m_break, -mass.alpha_1, mass.m_low, m_break)
    pl2_m_break = tpl_notnorm(m_break, -mass.alpha_2, m_break, mass.m_high)
    pdf = tpl_notnorm(m, -mass.alpha_1, mass.m_low, m_break)
    pdf = pdf + tpl_notnorm(m, -mass.alpha_2, m_break, mass.m_high) * pl1_m_break / pl2_m_break
    return pdf * smoothing(m, mass.delta_m, mass.m_low)
  if isinstance(mass, plp):
    P = tpl_notnorm(m, -mass.alpha, mass.m_low, mass.m_high) / tpl_cdf(-mass.alpha, mass.m_low, mass.m_high)
    G = truncated_gaussian(m, mass.mu_g, mass.sigma_g, mass.m_low, mass.mu_g + 5 * mass.sigma_g)
    pdf = (1 - mass.lambda_peak) * P + mass.lambda_peak * G
    return pdf * smoothing(m, mass.delta_m, mass.m_low)
  raise TypeError(mass)


def secondary_mass_conditioned_pdf_notnorm(mass, m2, m1):
  """mass.py:320-328."""
  m2 = np.asarray(m2, dtype=np.float64)
  pdf = tpl_notnorm(m2, mass.beta, mass.m_low, m1)
  if isinstance(mass, tpl):
    return pdf
  return pdf * smoothing(m2, mass.delta_m, mass.m_low)


def p_m1m2(mass, m1, m2):
  """mass.py:334-341."""
  m1 = np.asarray(m1, dtype=np.float64)
  m2 = np.asarray(m2, dtype=np.float64)
  p_m1 = primary_mass_pdf_notnorm(mass, m1) / mass.norm_p_m1
  with np.errstate(all='ignore'):
    p_m2m1 = secondary_mass_conditioned_pdf_notnorm(mass, m2, m1) / jnp_interp(m1, mass.m_grid, mass.cdf_m2_conditioned)
  p_m2m1 = np.where(np.isnan(p_m2m1), 0., p_m2m1)
  return p_m1 * p_m2m1


def pdf_joint_and_marg(mass, res=(5000, 2500)):
  """mass.py:351-362 (plotting helper on p_m1m2 + trapz)."""
  m1 = jnp_linspace(mass.m_low, mass.m_high, res[0])
  m2 = jnp_linspace(mass.m_low, mass.m_high, res[1])
  m1mesh, m2mesh = np.meshgrid(m1, m2)
  p_joint = p_m1m2(mass, m1mesh, m2mesh)
  p1_marg = trapz(p_joint, m2[:, None] * np.ones_like(p_joint), axis=0)
  p1_marg = p1_marg / trapz(p1_marg, m1)
  p2_marg = trapz(p_joint, m1, axis=1)
  p2_marg = p2_marg / trapz(p2_marg, m2)
  return {'m1': m1, 'm2': m2, 'm1mesh': m1mesh, 'm2mesh': m2mesh, 'p_joint': p_joint, 'p_m1_marg': p1_marg, 'p_m2_marg': p2_marg}


# ----------------------------------------------------------------------------------------------------------
# models: rate  (CHIMERA/population/rate.py)
# ----------------------------------------------------------------------------------------------------------

class power_law(_Params):
  name = 'power_law'
  default = {'gamma': 1.7}


class madau_dickinson(_Params):
  name = 'madau_dickinson'
  default = {'gamma': 2.7, 'kappa': 3.0, 'zp': 2.}


class trunc_madau_dickinson(_Params):
  name = 'trunc_madau_dickinson'
  default = {'gamma': 2.7, 'kappa': 3.0, 'zp': 2., 'zmax': 1.3}


class trunc_power_law(_Params):
  name = 'trunc_power_law'
  default = {'gamma': 1.9, 'zmax': 1.3}


def merger_rate(rate, z):
  """rate.py:96-122."""
  z = np.asarray(z, dtype=np.float64)
  if isinstance(rate, power_law):
    return (1. + z)**rate.gamma
  if isinstance(rate, trunc_power_law):
    pdf = (1. + z)**rate.gamma
    norm = ((1 + rate.zmax)**(rate.gamma + 1) - 1) / (rate.gamma + 1)
    return np.where(z < rate.zmax, pdf / norm, 0.)
  md = (1. + z)**rate.gamma / (1. + ((1. + z) / (1. + rate.zp))**(rate.gamma + rate.kappa))
  one_over_norm = 1. + (1. + rate.zp)**(-rate.gamma - rate.kappa)
  if isinstance(rate, madau_dickinson):
    return one_over_norm * md
  if isinstance(rate, trunc_madau_dickinson):
    return np.where(z < rate.zmax, one_over_norm * md, 0.)
  raise TypeError(rate)


# ----------------------------------------------------------------------------------------------------------
# catalogue / completeness  (CHIMERA/catalog/catalog.py, completeness.py)
# ----------------------------------------------------------------------------------------------------------

class dVdz_completeness(object):
  """completeness.py:22-67 (kind='step' only; 'step_smooth' is broken in the reference)."""

  def __init__(self, z_range=(0.073, 1.3)):
    self.z_range = np.asarray(z_range, dtype=np.float64)

  def P_compl(self, zgrids):
    return np.where(np.logical_and(zgrids > self.z_range[0], zgrids < self.z_range[1]), 1., 0.)

  def fR(self, cosmo):
    res = Vc_at_z(cosmo, self.z_range)
    return res[1] - res[0]

  def p_bkg(self, cosmo, z, distances=None):
    return dVcdz_at_z(cosmo, z, distances)


class empty_catalog(object):
  """catalog.py:19-43."""
  max_npixels = None
  neff_pixels = None

  def p_gal(self, cosmo, z):
    return dVcdz_at_z(cosmo, z)

  def p_bkg(self, cosmo, z, distances=None):
    return dVcdz_at_z(cosmo, z, distances)


class pixelated_catalog(object):
  """catalog.py:51-203 -- runtime part only: holds p_cat (E,P,Z; -100 padded), P_compl (E,1,Z), neff_pixels."""

  def __init__(self, completeness, p_cat, z_grids, neff_pixels):
    self.completeness = completeness
    self.p_cat = np.asarray(p_cat, dtype=np.float64)
    self.P_compl = completeness.P_compl(np.asarray(z_grids))[:, None, :]       # catalog.py:195
    self.max_npixels = self.p_cat.shape[1]
    self.neff_pixels = np.asarray(neff_pixels)

  def p_bkg(self, cosmo, z, distances=None):
    return self.completeness.p_bkg(cosmo, z, distances)

  def p_gal(self, cosmo, z):
    """catalog.py:197-203."""
    fR = self.completeness.fR(cosmo)
    p_bkg = self.completeness.p_bkg(cosmo, z)[:, None, :]
    p_gal = fR * self.p_cat + (1. - self.P_compl) * p_bkg
    return np.where(self.p_cat != -100., p_gal, -100.)


def _gaussian(x, mu, sigma):
  """catalog.py:209-210."""
  return np.power(2 * np.pi * (sigma ** 2), -0.5) * np.exp(-0.5 * np.power((x - mu) / sigma, 2.))


def sum_gaussians_ucv(z_grid, mu, sigma, cosmo, weights=None):
  """catalog.py:212-221."""
  if len(mu) == 0:
    return np.zeros_like(z_grid)
  if weights is None:
    weights = np.ones(len(mu))
  zgrid = z_grid[:, None]
  gauss = _gaussian(zgrid, mu, sigma)
  gauss = gauss * dVcdz_at_z(cosmo, zgrid)
  norm = trapz(gauss, zgrid, axis=0)
  with np.errstate(all='ignore'):
    return np.sum(weights * gauss / norm, axis=1) / np.sum(weights)


def sum_gaussians_pbkg(z_grid, mu, sigma, cosmo, p_bkg, weights=None):
  """catalog.py:223-231: as sum_gaussians_ucv with the completeness model's p_bkg(cosmo, z) in place of dVc/dz."""
  if len(mu) == 0:
    return np.zeros_like(z_grid)
  if weights is None:
    weights = np.ones(len(mu))
  zgrid = z_grid[:, None]
  gauss = _gaussian(zgrid, mu, sigma) * np.asarray(p_bkg(cosmo, zgrid))
  norm = trapz(gauss, zgrid, axis=0)
  with np.errstate(all='ignore'):
    return np.sum(weights * gauss / norm, axis=1) / np.sum(weights)


def compute_p_cat_event(z_grid, gal_z, gal_zerr, gal_w, gal_pix, good_pix, max_npixels, cosmo, p_bkg=None):
  """catalog.py:152-178: per-pixel sum of galaxy Gaussians; non-finite -> 0; padded with -100.  p_bkg: sumgauss='pbkg' (:164-171)."""
  sel = (gal_z > z_grid[0]) & (gal_z < z_grid[-1])                              # catalog.py:148
  gal_z, gal_zerr, gal_w, gal_pix = gal_z[sel], gal_zerr[sel], gal_w[sel], gal_pix[sel]
  if p_bkg is not None:
    p_cat = np.array([sum_gaussians_pbkg(z_grid, gal_z[gal_pix == p], gal_zerr[gal_pix == p], cosmo, p_bkg,
                                         weights=gal_w[gal_pix == p]) for p in good_pix])
  else:
    p_cat = np.array([sum_gaussians_ucv(z_grid, gal_z[gal_pix == p], gal_zerr[gal_pix == p], cosmo,
                                        weights=gal_w[gal_pix == p]) for p in good_pix])
  p_cat[~np.isfinite(p_cat)] = 0.
  if len(good_pix) < max_npixels:
    p_cat = np.concatenate([p_cat, np.full((max_npixels - len(good_pix), len(z_grid)), -100.)], axis=0)
  ngal = int(np.sum([np.sum(gal_pix == p) for p in good_pix]))
  return p_cat, ngal


# ----------------------------------------------------------------------------------------------------------
# data structs  (CHIMERA/data.py:15-59)
# ----------------------------------------------------------------------------------------------------------

class _Theta(object):
  _fields = ()

  def __init__(self, **kw):
    for f in self._fields:
      v = kw.get(f, None)
      setattr(self, f, None if v is None else np.asarray(v))

  def update(self, **kw):
    d = {f: getattr(self, f) for f in self._fields}
    d.update(kw)
    return self.__class__(**d)


class theta_pe_det(_Theta):
  _fields = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix',
             'gw_loc2d_pdf', 'pixels_pe_opt_nside')

  def __init__(self, **kw):
    super().__init__(**kw)
    if self.pe_prior is None and self.dL is not None:                            # data.py:45-47
      self.pe_prior = np.ones_like(self.dL)


class theta_inj_det(_Theta):
  _fields = ('m1det', 'm2det', 'dL', 'p_draw')


class theta_src(_Theta):
  _fields = ('m1src', 'm2src', 'z', 'original_distances')


# ----------------------------------------------------------------------------------------------------------
# population glue  (CHIMERA/population/pop_wrapper.py)
# ----------------------------------------------------------------------------------------------------------

class population(object):
  """pop_wrapper.py:14-64."""

  def __init__(self, cosmo, mass, rate, R0=1., gal_cat=None, Tobs=1, scale_free=True):
    self.cosmo, self.mass, self.rate, self.R0 = cosmo, mass, rate, R0
    self.gal_cat = empty_catalog() if gal_cat is None else gal_cat
    self.Tobs, self.scale_free = Tobs, scale_free

  def update(self, **lam):
    return self.__class__(self.cosmo.update(**lam), self.mass.update(**lam), self.rate.update(**lam),
                          lam.get('R0', self.R0), self.gal_cat, self.Tobs, self.scale_free)


def theta_det2src(cosmo, theta_det, include_original_distances=False):
  """pop_wrapper.py:67-75."""
  z = z_from_dGW(cosmo, theta_det.dL)
  m1s, m2s = theta_det.m1det / (1. + z), theta_det.m2det / (1. + z)
  if include_original_distances:
    return theta_src(m1src=m1s, m2src=m2s, z=z, original_distances=theta_det.dL)
  return theta_src(m1src=m1s, m2src=m2s, z=z)


def get_theta_src_and_weights(pop, theta_det):
  """pop_wrapper.py:77-80."""
  th_src = theta_det2src(pop.cosmo, theta_det)
  with np.errstate(all='ignore'):
    weights = p_m1m2(pop.mass, th_src.m1src, th_src.m2src) / theta_det.pe_prior
  return th_src, weights


def p_cbc(pop, z):
  """pop_wrapper.py:82-90."""
  p_gal = pop.gal_cat.p_gal(pop.cosmo, z)
  p_rate = merger_rate(pop.rate, z) / (1 + z)
  if p_gal.ndim > p_rate.ndim:
    return np.where(p_gal != -100, p_gal * p_rate[:, None, :], -100)
  return p_gal * p_rate


def pop_rate_det_inj(pop, th_det):
  """pop_wrapper.py:102-111 (theta_inj_det overload)."""
  th = theta_det2src(pop.cosmo, th_det, include_original_distances=True)
  with np.errstate(all='ignore'):
    p_z = pop.gal_cat.p_bkg(pop.cosmo, th.z, th.original_distances)
    p_z = p_z * (merger_rate(pop.rate, th.z) / (1. + th.z))
    dN = pop.R0 * p_m1m2(pop.mass, th.m1src, th.m2src) * p_z
    jac = np.abs(ddLdz_at_z(pop.cosmo, th.z, th.original_distances)) * (1. + th.z)**2
    return dN / jac


def compute_z_grids(cosmo, theta_det, cosmo_prior=None, z_int_res=300, z_conf_range=None):
  """pop_wrapper.py:133-208."""
  events_dL = theta_det.dL
  if isinstance(z_conf_range, list):
    dL_min, dL_max = np.percentile(events_dL, z_conf_range, axis=1)
  elif isinstance(z_conf_range, (int, float)):
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
  lc_low = {k: cp[k][0] for k in base}
  lc_high = {k: cp[k][1] for k in base}
  if cosmo.name != 'flrw':
    lc_low.update(Xi0=cp['Xi0'][1], n=cp['n'][1])
    lc_high.update(Xi0=cp['Xi0'][0], n=cp['n'][1])
  cosmo1 = cosmo.update(**lc_low, z_grid_res=10_000)
  cosmo2 = cosmo.update(**lc_high, z_grid_res=10_000)
  z_min = z_from_dGW(cosmo1, dL_min)
  z_max = z_from_dGW(cosmo2, dL_max)
  return jnp_linspace(z_min, z_max, z_int_res)


# ----------------------------------------------------------------------------------------------------------
# KDE numerics  (CHIMERA/utils/math.py)
# ----------------------------------------------------------------------------------------------------------

def binning1d(dataset, weights, num_bins=200):
  """math.py:32-46.  A NaN bin index (max==min) carries zero/NaN density downstream either way (std of the
  centres is 0 -> bandwidth 0 -> 0/0); it is mapped to bin 0 here to keep NumPy indexing defined."""
  min_val, max_val = np.min(dataset), np.max(dataset)
  bin_edges = jnp_linspace(min_val, max_val, num_bins + 1)
  bin_centers = (bin_edges[:-1] + bin_edges[1:]) / 2
  with np.errstate(all='ignore'):
    f = np.clip(np.floor((dataset - min_val) / (max_val - min_val) * num_bins), 0, num_bins - 1)
  idx = np.where(np.isnan(f), 0, f).astype(np.int64)
  counts = np.zeros(num_bins)
  np.add.at(counts, idx, weights)
  return bin_centers, counts


def _epan_kernel(u):
  """math.py:83-85."""
  return np.where(np.abs(u) <= 1, 3 / 4 * (1 - u**2), 0)


def _gaussian_kernel(u):
  """math.py:87-89."""
  return np.exp(-0.5 * u**2) / np.sqrt(2 * np.pi)


def kde1d(dataset, grid, weights=None, kernel='epan', bw_method=None):
  """math.py:52-81."""
  with np.errstate(all='ignore'):
    if weights is None:
      weights = np.ones_like(dataset) / dataset.size
    else:
      weights = weights / np.sum(weights)
    neff = 1.0 / np.sum(np.power(weights, 2))
    if bw_method == "scott" or bw_method is None:
      bandwidth = np.power(neff, -1. / (1 + 4)) * np.std(dataset)
    elif bw_method == "silverman":
      bandwidth = np.power(neff * (1 + 2) / 4.0, -1. / (1 + 4)) * np.std(dataset)
    elif np.isscalar(bw_method) and not isinstance(bw_method, str):
      bandwidth = bw_method * np.std(dataset)
    else:
      raise ValueError("bw_method should be 'scott', 'silverman', or a scalar")
    kernel_fn = _epan_kernel if kernel == 'epan' else _gaussian_kernel
    kernel_vals = kernel_fn((grid[:, None] - dataset) / bandwidth)
    return np.sum(weights * kernel_vals, axis=-1) / bandwidth


def gkde_nd(dataset, evaluation_grid, weights=None, bw_method=None):
  """math.py:154-229 (numba_gkde_nd + numba_gaussian_kernel, in_log=False, CPU branch)."""
  dataset = np.atleast_2d(dataset)
  d, n = dataset.shape
  points = np.atleast_2d(evaluation_grid)
  if weights is not None:
    _w = weights / np.sum(weights)
  else:
    _w = np.full(n, 1.0 / n)
  neff = 1.0 / np.sum(np.power(_w, 2))
  if bw_method == "scott" or bw_method is None:
    factor = np.power(neff, -1. / (d + 4))
  elif bw_method == "silverman":
    factor = np.power(neff * (d + 2) / 4.0, -1. / (d + 4))
  elif np.isscalar(bw_method) and not isinstance(bw_method, str):
    factor = bw_method
  else:
    raise ValueError("`bw_method` should be 'scott', 'silverman', a scalar")
  _mean = np.sum(_w * dataset, axis=1)
  _res = dataset - _mean[:, None]
  cov = np.atleast_2d(np.dot(_res * _w, _res.T))
  cov = cov / (1 - np.sum(_w ** 2))
  inv_cov = np.linalg.inv(cov) / factor**2
  L = np.linalg.cholesky(inv_cov)
  pw = np.dot(np.ascontiguousarray(points.T), L)
  dw = np.dot(np.ascontiguousarray(dataset.T), L)
  log_norm = np.sum(np.log(np.diag(L))) - 0.5 * d * np.log(2 * np.pi)
  out = np.zeros(pw.shape[0])
  CH = 256
  for s in range(0, pw.shape[0], CH):
    d2 = ((dw[None, :, :] - pw[s:s + CH, None, :])**2).sum(-1)
    out[s:s + CH] = (_w[None, :] * np.exp(log_norm - 0.5 * d2)).sum(-1)
  return out


# ----------------------------------------------------------------------------------------------------------
# hyper-likelihood  (CHIMERA/likelihood.py) and selection function (CHIMERA/selection_function.py)
# ----------------------------------------------------------------------------------------------------------

class selection_function(object):
  """selection_function.py:10-53."""

  def __init__(self, theta_inj_det, N_inj, N_eff=5.):
    self.theta_inj_det, self.N_inj, self.N_eff = theta_inj_det, N_inj, N_eff

  def dN(self, pop):
    with np.errstate(all='ignore'):
      return pop_rate_det_inj(pop, self.theta_inj_det) / self.theta_inj_det.p_draw

  def N_exp(self, pop):
    """selection_function.py:34-48 (nansum for xi, plain sum for the variance: SURVEY Q10)."""
    dN = self.dN(pop)
    xi = np.nansum(dN, axis=-1) / self.N_inj
    Nexp = pop.Tobs * xi
    if self.N_eff is not None:
      with np.errstate(all='ignore'):
        variance2 = np.sum(dN**2, axis=-1) / self.N_inj**2 - xi**2 / self.N_inj
        neff = xi**2 / variance2
      Nexp = np.where(neff < self.N_eff, 0.0, Nexp)
    return Nexp

  __call__ = N_exp


class hyperlikelihood(object):
  """likelihood.py:14-338."""

  def __init__(self, theta_gw_det, z_grids, population, selection_function=None, kind_p_gw3d=None,
               kernel='epan', bw_method=None, cut_grid=2.0, binning=True, num_bins=200, pe_neff=2.0):
    self.theta_gw_det, self.population, self.z_grids = theta_gw_det, population, np.asarray(z_grids)
    self.selection_function, self.kind_p_gw3d = selection_function, kind_p_gw3d
    self.kernel, self.bw_method, self.cut_grid = kernel, bw_method, cut_grid
    self.binning, self.num_bins, self.pe_neff = binning, num_bins, pe_neff
    self.pixelated = theta_gw_det.pixels_opt_nsides is not None
    self.nevents = len(theta_gw_det.dL)
    self.z_int_res = self.z_grids.shape[1]
    if self.pixelated:
      assert kind_p_gw3d in ['approximate', 'marginalized', 'full'], \
        "`kind_p_gw3d` must be one of `approximate`, `marginalized`, or `full`"
      self.max_npixels = population.gal_cat.max_npixels
      self.neff_pixels = population.gal_cat.neff_pixels
      self.p_gw3d = {'approximate': self.p_gw3dapprox, 'marginalized': self.p_gw3dmarg,
                     'full': self.p_gw3dfull}[kind_p_gw3d]
      self.compute_numlike_evs = self._compute_numlike_evs_pixelated
    else:
      self.compute_numlike_evs = self._compute_numlike_evs_no_pixels

  # --- p_gw1d, likelihood.py:105-144
  def p_gw1d(self, pop):
    th_src, weights = get_theta_src_and_weights(pop, self.theta_gw_det)
    with np.errstate(all='ignore'):
      norms = np.mean(weights, axis=-1)
      n_effs = np.sum(weights, axis=-1)**2 / np.sum(weights**2, axis=-1)
    Z = self.z_int_res
    out = np.zeros((self.nevents, Z))
    for ev in range(self.nevents):
      z, w = th_src.z[ev], weights[ev]
      if self.cut_grid is not None:
        data_min, data_max, sigma = np.min(z), np.max(z), np.std(z)
        lb = data_min - self.cut_grid * sigma if data_min - self.cut_grid * sigma > 0. else 1.e-8
        ub = data_max + self.cut_grid * sigma
        eff = jnp_linspace(lb, ub, Z // 2)
      else:
        eff = self.z_grids[ev]
      zs, ws = binning1d(z, w, self.num_bins) if self.binning else (z, w)
      if n_effs[ev] >= self.pe_neff:
        kde = kde1d(zs, eff, ws, self.kernel, self.bw_method) * norms[ev]
        out[ev] = jnp_interp(self.z_grids[ev], eff, kde, left=0., right=0.)
    return out

  # --- likelihood.py:150-154
  def p_gw3dapprox(self, pop):
    return self.p_gw1d(pop)[:, None, :] * self.theta_gw_det.gw_loc2d_pdf[:, :, None]

  # --- likelihood.py:160-205  (kde1d called WITHOUT kernel= -> always 'epan')
  def p_gw3dmarg(self, pop):
    th_src, weights = get_theta_src_and_weights(pop, self.theta_gw_det)
    with np.errstate(all='ignore'):
      norms = np.mean(weights, axis=-1)
      n_effs = np.sum(weights, axis=-1)**2 / np.sum(weights**2, axis=-1)
    P, Z = self.max_npixels, self.z_int_res
    out = np.zeros((self.nevents, P, Z))
    for ev in range(self.nevents):
      if not (n_effs[ev] >= self.pe_neff):
        continue
      z, w, zgrid = th_src.z[ev], weights[ev], self.z_grids[ev]
      pe_pix = self.theta_gw_det.pixels_pe_opt_nside[ev]
      pixels = self.theta_gw_det.pixels_opt_nsides[ev]
      gw_pdf = self.theta_gw_det.gw_loc2d_pdf[ev]
      if self.cut_grid is not None:
        zmin = np.maximum(np.min(z) - self.cut_grid * np.std(z), 1e-8)
        zmax = np.max(z) + self.cut_grid * np.std(z)
        eff = jnp_linspace(zmin, zmax, Z // 2)
      else:
        eff = zgrid
      for i in range(P):
        mask = pe_pix == pixels[i]
        z_m = np.where(mask, z, np.min(z))
        w_m = np.where(mask, w, 0.0)
        z_pix, w_pix = binning1d(z_m, w_m, self.num_bins) if self.binning else (z_m, w_m)
        kde_eff = kde1d(z_pix, eff, weights=w_pix, bw_method=self.bw_method)
        out[ev, i] = jnp_interp(zgrid, eff, kde_eff, left=0.0, right=0.0) * norms[ev] * gw_pdf[i]
    return out

  # --- likelihood.py:211-260
  def p_gw3dfull(self, pop):
    th_src, weights = get_theta_src_and_weights(pop, self.theta_gw_det)
    with np.errstate(all='ignore'):
      norms = np.mean(weights, axis=-1)
      n_effs = np.sum(weights, axis=-1)**2 / np.sum(weights**2, axis=-1)
    z_std = np.std(th_src.z, axis=1, keepdims=True)
    z_max = np.max(th_src.z, axis=1, keepdims=True)
    z_min = np.min(th_src.z, axis=1, keepdims=True)
    z_masks = (self.z_grids <= z_max + self.cut_grid * z_std) & (self.z_grids >= z_min - self.cut_grid * z_std)
    result = np.zeros((self.nevents, self.max_npixels, self.z_int_res))
    for ev in range(self.nevents):
      if n_effs[ev] < self.pe_neff:
        continue
      z_mask = z_masks[ev]
      z_eff = self.z_grids[ev][z_mask]
      npix = int(self.neff_pixels[ev])
      ra_pix = self.theta_gw_det.ra_pix[ev, :npix]
      dec_pix = self.theta_gw_det.dec_pix[ev, :npix]
      eff_grid = np.array([np.tile(z_eff, npix), np.repeat(ra_pix, len(z_eff)), np.repeat(dec_pix, len(z_eff))])
      eff_mask = np.tile(z_mask, npix)
      dat = np.array([th_src.z[ev], self.theta_gw_det.ra[ev], self.theta_gw_det.dec[ev]])
      kde_vals = np.zeros(npix * self.z_int_res)
      if eff_grid.shape[1] > 0:
        kde_vals[eff_mask] = gkde_nd(dat, eff_grid, weights=weights[ev], bw_method=self.bw_method)
      result[ev, :npix, :] = kde_vals.reshape(npix, self.z_int_res) * norms[ev]
    return result

  # --- likelihood.py:266-292
  def _compute_numlike_evs_pixelated(self, pop):
    p_gw3d = self.p_gw3d(pop)
    p_z = p_cbc(pop, self.z_grids)
    jacobian = ddLdz_at_z(pop.cosmo, self.z_grids) * (1. + self.z_grids)**2
    with np.errstate(all='ignore'):
      integrand = np.where(p_z != -100, p_gw3d * p_z / jacobian[:, None, :], 0.)
    like_evs_pixels = trapz(integrand, self.z_grids[:, None, :], axis=-1)
    return np.sum(like_evs_pixels, axis=-1)

  def _compute_numlike_evs_no_pixels(self, pop):
    p_gw = self.p_gw1d(pop)
    p_z = p_cbc(pop, self.z_grids)
    jacobian = ddLdz_at_z(pop.cosmo, self.z_grids) * (1. + self.z_grids)**2
    return trapz(p_gw * p_z / jacobian, self.z_grids, axis=-1)

  # --- likelihood.py:294-301
  def compute_log_likenum(self, pop):
    with np.errstate(all='ignore'):
      log_like_evs = nan_to_num_neginf(np.log(self.compute_numlike_evs(pop)))
      log_num = np.sum(log_like_evs, axis=-1)
    if not pop.scale_free:
      log_num = log_num + self.nevents * np.log(pop.R0 * pop.Tobs)
    return log_num

  # --- likelihood.py:307-320
  def compute_log_hyperlike(self, **lam):
    return self.compute_all(**lam)[3]

  __call__ = compute_log_hyperlike

  # --- likelihood.py:326-338
  def compute_all(self, **lam):
    pop = self.population.update(**lam)
    with np.errstate(all='ignore'):
      log_like_evs = nan_to_num_neginf(np.log(self.compute_numlike_evs(pop)))
      log_like_num = np.sum(log_like_evs, axis=-1)
      N_exp = self.selection_function.N_exp(pop)
      if not pop.scale_free:
        log_like_num = log_like_num + self.nevents * np.log(pop.R0 * pop.Tobs)
        log_hyper = log_like_num - N_exp
      else:
        log_hyper = log_like_num - self.nevents * np.log(N_exp)
      return log_like_evs, log_like_num, np.log(N_exp), log_hyper
