"""Catalogue completeness (reference: CHIMERA/catalog/completeness.py:22-67).

Only ``dVdz_completeness(kind='step')`` is a working model in the reference ('step_smooth' raises a shape error,
completeness.py:48; ``homogeneous_completeness`` is unfinished, :73-216), so that is what is provided.
"""
import numpy as np
from ..population.cosmo import dVcdz_at_z, Vc_at_z
from ..data import theta_src


class dVdz_completeness(object):
  builtin = True                  # evaluated inside the kernels; any other completeness object is a host plug-in (population/plugins.py)

  def __init__(self, z_range=(0.073, 1.3), kind="step", z_sig=None):
    self.z_range = np.asarray(z_range, dtype=np.float64)
    if kind != "step":
      raise ValueError("kind must be 'step' ('step_smooth' is not usable in the reference either: completeness.py:48)")
    self.kind = kind
    self.z_sig = z_sig

  def P_compl(self, zgrids):
    """completeness.py:43-52 -- cosmology independent step function, shape of ``zgrids``."""
    zgrids = np.asarray(zgrids, dtype=np.float64)
    return np.where(np.logical_and(zgrids > self.z_range[0], zgrids < self.z_range[1]), 1., 0.)

  def fR(self, cosmo_lambdas, normalized=False):
    """completeness.py:54-58."""
    res = Vc_at_z(cosmo_lambdas, self.z_range)
    return res[1] - res[0]

  def p_bkg(self, cosmo_lambdas, z):
    """completeness.py:60-67."""
    if isinstance(z, theta_src):
      return dVcdz_at_z(cosmo_lambdas, z)
    return dVcdz_at_z(cosmo_lambdas, np.asarray(z, dtype=np.float64))
