"""Merger-rate models (reference: CHIMERA/population/rate.py): four closed-form models evaluated on the GPU."""
import numpy as np
from .. import _lib
from ..data import theta_src
from ._base import base_struct, make_params, model_eval


class base_rate_struct(base_struct):
  name = 'base_rate_struct'


class power_law(base_rate_struct):
  """rate.py:32-49."""
  name = 'power_law'
  default = {'gamma': 1.7}

  def _pack(self):
    return dict(model=0, vec=[self.gamma, 0., 0., 0.])


class madau_dickinson(base_rate_struct):
  """rate.py:51-72."""
  name = 'madau_dickinson'
  default = {'gamma': 2.7, 'kappa': 3.0, 'zp': 2.}

  def _pack(self):
    return dict(model=1, vec=[self.gamma, self.kappa, self.zp, 0.])


class trunc_madau_dickinson(base_rate_struct):
  """rate.py:74-81."""
  name = 'trunc_madau_dickinson'
  default = {'gamma': 2.7, 'kappa': 3.0, 'zp': 2., 'zmax': 1.3}

  def _pack(self):
    return dict(model=3, vec=[self.gamma, self.kappa, self.zp, self.zmax])


class trunc_power_law(base_rate_struct):
  """rate.py:83-88."""
  name = 'trunc_power_law'
  default = {'gamma': 1.9, 'zmax': 1.3}

  def _pack(self):
    return dict(model=2, vec=[self.gamma, 0., 0., self.zmax])


def merger_rate(rate, z):
  """rate.py:96-129."""
  if isinstance(z, theta_src):
    z = z.z
  if not hasattr(rate, '_pack'):                 # plug-in rate model (population/plugins.py): its own host function
    return np.asarray(rate.merger_rate(np.asarray(z, dtype=np.float64)), dtype=np.float64)
  return model_eval(make_params(rate=rate), _lib.F_RATE, z)
