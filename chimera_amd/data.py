"""Data containers of the hyper-likelihood path (reference: CHIMERA/data.py:15-59).

Plain attribute containers with the reference's field names and ``.update(**kw)`` (which returns a new object, as
``eqx.tree_at`` does in data.py:16-25).  Arrays are kept as NumPy arrays on the host; the HIP handle copies them to
HBM once (see likelihood.py).  File loaders / HEALPix pixelisation (data.py:70-484) are outside the hot path.
"""
import numpy as np


class theta_generic(object):
  _fields = ()

  def __init__(self, **kwargs):
    unknown = set(kwargs) - set(self._fields)
    if unknown:
      raise TypeError(f"{self.__class__.__name__}: unexpected field(s) {sorted(unknown)}")
    for f in self._fields:
      v = kwargs.get(f, None)
      if v is not None and not isinstance(v, dict):
        v = np.asarray(v)
      setattr(self, f, v)
    self.__post_init__()

  def __post_init__(self):
    pass

  def update(self, **kwargs):
    d = {f: getattr(self, f) for f in self._fields}
    for k in kwargs:
      if k not in self._fields:
        raise AttributeError(f"{self.__class__.__name__} has no field '{k}'")
    d.update(kwargs)
    return self.__class__(**d)

  def __repr__(self):
    parts = []
    for f in self._fields:
      v = getattr(self, f)
      if v is not None:
        parts.append(f"{f}={getattr(v, 'shape', v)}")
    return f"{self.__class__.__name__}({', '.join(parts)})"


class theta_pe_det(theta_generic):
  """data.py:27-47."""
  _fields = ('m1det', 'm2det', 'dL', 'phi', 'theta', 'ra', 'dec', 'pe_prior', 'pixels_pe_all_nsides', 'opt_nsides',
             'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')

  def __post_init__(self):
    if self.pe_prior is None and self.dL is not None:
      self.pe_prior = np.ones_like(self.dL, dtype=np.float64)


class theta_inj_det(theta_generic):
  """data.py:49-53."""
  _fields = ('m1det', 'm2det', 'dL', 'p_draw')


class theta_src(theta_generic):
  """data.py:55-59."""
  _fields = ('m1src', 'm2src', 'z', 'original_distances')


theta_pe_datasets = ['m1det', 'm2det', 'dL', 'pe_prior']
theta_pe_pixelated_datasets = ['m1det', 'm2det', 'dL', 'pe_prior', 'ra', 'dec', 'theta', 'phi', 'opt_nsides',
                               'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside']


def save_npz(fname, obj):
  """Store a theta_* container as ``.npz`` (stand-in for the reference's HDF5 layout, data.py:61-64, io.py:7-66)."""
  np.savez(fname, **{f: getattr(obj, f) for f in obj._fields
                     if getattr(obj, f) is not None and not isinstance(getattr(obj, f), dict)})


def load_npz(fname, cls=theta_pe_det):
  with np.load(fname) as d:
    return cls(**{k: d[k] for k in d.files if k in cls._fields})
