"""Shared machinery of the model parameter containers and the chm_params marshalling."""
import numpy as np
from .. import _lib

# defaults used to fill the parts of chm_params a call does not care about (e.g. a cosmology-only function)
_DEF_MASS = dict(model=2, vec=[5.1, 87., 0.039, 3.4, 1.1, 4.8, 34., 3.6], grid_res=1000)
_DEF_RATE = dict(model=1, vec=[2.7, 3.0, 2.0, 1.3])
_DEF_COSMO = dict(model=0, vec=[70., 0.25, 0., 0., -1., 0., 1., 0.], z_max=10., z_grid_res=1500)


class base_struct(object):
  """Reference: base_cosmology_struct / base_mass_paired_struct / base_rate_struct
  (cosmo.py:13-40, mass.py:13-42, rate.py:10-30): ``default`` dict, ``keys``, ``as_dict``, ``update(**kw)``."""
  default = {}
  name = 'base_struct'

  def __init__(self, **kwargs):
    self.keys = list(self.default.keys())
    for key in self.keys:
      setattr(self, key, kwargs.get(key, self.default[key]))
    self._tables = None

  @property
  def as_dict(self):
    return {k: getattr(self, k) for k in self.keys}

  def update(self, **kwargs):
    keys_to_update = {k: v for k, v in kwargs.items() if k in self.keys}
    if keys_to_update == {}:
      return self                      # no change: same object (cosmo.py:35-37)
    fiducials = self.as_dict
    fiducials.update(keys_to_update)
    return self.__class__(**fiducials)

  def __repr__(self):
    return f"{self.__class__.__name__}({', '.join(f'{k}={getattr(self, k)}' for k in self.keys)})"


def make_params(cosmo=None, mass=None, rate=None, R0=1., Tobs=1., scale_free=True, has_catalog=False,
                z_range=(0.073, 1.3)):
  """Fill one ``chm_params`` (include/chimera_hip.h) from model objects."""
  p = _lib.chm_params()
  if cosmo is not None and not hasattr(cosmo, '_pack'):                                    # plug-in cosmology: only the table size travels
    c = dict(_DEF_COSMO, z_max=float(cosmo.z_max), z_grid_res=len(cosmo.z_grid_interp))
  else:
    c = cosmo._pack() if cosmo is not None else _DEF_COSMO
  m = mass._pack() if (mass is not None and hasattr(mass, '_pack')) else _DEF_MASS        # plug-in models (population/plugins.py):
  r = rate._pack() if (rate is not None and hasattr(rate, '_pack')) else _DEF_RATE        # their values come from the host
  p.cosmo_model, p.mass_model, p.rate_model = c['model'], m['model'], r['model']
  p.z_grid_res, p.mass_grid_res = int(c['z_grid_res']), int(m['grid_res'])
  p.scale_free, p.has_catalog = int(bool(scale_free)), int(bool(has_catalog))
  p.z_max = float(c['z_max'])
  for i, v in enumerate(c['vec']):
    p.cosmo[i] = float(v)
  for i in range(_lib.NMASS):
    p.mass[i] = float(m['vec'][i]) if i < len(m['vec']) else 0.
  for i in range(_lib.NRATE):
    p.rate[i] = float(r['vec'][i]) if i < len(r['vec']) else 0.
  p.R0, p.Tobs = float(R0), float(Tobs)
  p.compl_z0, p.compl_z1 = float(z_range[0]), float(z_range[1])
  return p


def param_slots(cosmo, mass, rate):
  """hyper-parameter name -> [(chm_params field, index or None, is_int)] for the models of a population: lets
  ``hyperlikelihood.batch`` patch a copy of the base ``chm_params`` instead of rebuilding the model objects per draw."""
  slots = {}

  def add(key, field, idx=None, is_int=False):
    slots.setdefault(key, []).append((field, idx, is_int))
  cnames = ['H0', 'Om0', 'Ok0', 'Or0', 'w0', 'wa', 'Xi0', 'n']
  for k in cosmo.keys:
    if k in cnames:
      add(k, 'cosmo', cnames.index(k))
    elif k == 'z_max':
      add(k, 'z_max')
    elif k == 'z_grid_res':
      add(k, 'z_grid_res', None, True)
  mnames = {0: ['m_low', 'm_high', 'alpha', 'beta'],
            1: ['m_low', 'm_high', 'alpha_1', 'alpha_2', 'beta', 'delta_m', 'break_fraction'],
            2: ['m_low', 'm_high', 'lambda_peak', 'alpha', 'beta', 'delta_m', 'mu_g', 'sigma_g']}[mass._pack()['model']]
  for k in mass.keys:
    if k in mnames:
      add(k, 'mass', mnames.index(k))
    elif k == 'grid_res':
      add(k, 'mass_grid_res', None, True)
  rnames = {0: ['gamma'], 1: ['gamma', 'kappa', 'zp'], 2: ['gamma', None, None, 'zmax'], 3: ['gamma', 'kappa', 'zp', 'zmax']}[rate._pack()['model']]
  for k in rate.keys:
    if k in rnames:
      add(k, 'rate', rnames.index(k))
  add('R0', 'R0')
  return slots


def model_eval(params, func, a, b=None, device=None):
  """out = f(a[, b]) elementwise on the device (chm_model_eval); keeps the input shape."""
  L = _lib.lib()
  a_arr = np.asarray(a, dtype=np.float64)
  shape = a_arr.shape
  if b is not None:
    a_arr, b_arr = np.broadcast_arrays(a_arr, np.asarray(b, dtype=np.float64))
    shape = a_arr.shape
    b_flat = _lib.as_f64(b_arr).ravel()
  else:
    b_flat = None
  a_flat = _lib.as_f64(a_arr).ravel()
  out = np.empty_like(a_flat)
  dev = _lib.default_device() if device is None else device
  _lib.check(L.chm_model_eval(params, func, _lib.dptr(a_flat), _lib.dptr(b_flat), a_flat.size, _lib.dptr(out), dev))
  return out.reshape(shape) if shape else out.reshape(())[()]


def model_tables(params, device=None):
  L = _lib.lib()
  Tc, Tm = params.z_grid_res, params.mass_grid_res
  zt, It, dLt = np.empty(Tc), np.empty(Tc), np.empty(Tc)
  mg, cdf, sc = np.empty(Tm), np.empty(Tm), np.empty(2)
  dev = _lib.default_device() if device is None else device
  _lib.check(L.chm_model_tables(params, _lib.dptr(zt), _lib.dptr(It), _lib.dptr(dLt), _lib.dptr(mg), _lib.dptr(cdf),
                                _lib.dptr(sc), dev))
  return dict(z_grid_interp=zt, integral_invE_interp=It, dL_interp=dLt, m_grid=mg, cdf_m2_conditioned=cdf,
              norm_p_m1=sc[0], fR=sc[1])
