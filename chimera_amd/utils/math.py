"""Numerical building blocks with the reference's names and signatures (CHIMERA/utils/math.py), evaluated on the GPU through
the C ABI (``chm_kde1d``, ``chm_binning1d``, ``chm_gkde_nd``, ``chm_trapz``, ``chm_cumtrapz``).  The hyper-likelihood itself does
not call these -- its kernels fuse the same arithmetic (chm_kernels.h) -- they serve callers that use the pieces directly
(diagnostic plots, the sky-localisation density of ``pixelize_gw_catalog``).  No CPU fallback.
"""
import numpy as np
from .. import _lib


def _bw(bw_method):
  """math.py:65-75 / 116-124 -> (code, scalar)."""
  if bw_method == "scott" or bw_method is None:
    return 0, 0.
  if bw_method == "silverman":
    return 1, 0.
  if np.isscalar(bw_method) and not isinstance(bw_method, str):
    return 2, float(bw_method)
  raise ValueError("bw_method should be 'scott', 'silverman', or a scalar")


def trapz(y, x=None, dx=1.0, axis=-1):
  """math.py:10-16 (``jnp.trapezoid``): integral along ``axis``; ``x`` one-dimensional or of ``y``'s shape."""
  y = np.asarray(y, dtype=np.float64)
  ym = np.moveaxis(y, axis, -1)
  n = ym.shape[-1]
  if x is None:
    x = np.arange(n, dtype=np.float64) * dx
  x = np.asarray(x, dtype=np.float64)
  per_row = x.ndim > 1
  if per_row:
    x = np.moveaxis(np.broadcast_to(x, y.shape), axis, -1)
  elif x.shape[0] != n:
    raise ValueError("trapz: x and y differ in length along the axis")
  rows = int(np.prod(ym.shape[:-1])) if ym.ndim > 1 else 1
  out = np.empty(rows)
  if rows == 0 or n == 0:
    return np.zeros(ym.shape[:-1])
  yc, xc = _lib.as_f64(ym).reshape(rows, n), _lib.as_f64(x).reshape(-1)
  _lib.check(_lib.lib().chm_trapz(_lib.dptr(yc), _lib.dptr(xc), rows, n, 1 if per_row else 0, _lib.dptr(out), _lib.default_device()))
  return out.reshape(ym.shape[:-1]) if ym.ndim > 1 else out[0]


def cumtrapz(y, x):
  """math.py:22-26: [0, cumsum(0.5 (y[:-1] + y[1:]) diff(x))]."""
  y, x = _lib.as_f64(y).reshape(-1), _lib.as_f64(x).reshape(-1)
  if y.shape != x.shape:
    raise ValueError("cumtrapz: x and y differ in length")
  out = np.empty_like(y)
  _lib.check(_lib.lib().chm_cumtrapz(_lib.dptr(y), _lib.dptr(x), y.size, _lib.dptr(out), _lib.default_device()))
  return out


def binning1d(dataset, weights, num_bins=200):
  """math.py:32-46 -> (bin_centers, bin_counts)."""
  x, w = _lib.as_f64(dataset).reshape(-1), _lib.as_f64(weights).reshape(-1)
  if x.shape != w.shape:
    raise ValueError("binning1d: dataset and weights differ in length")
  centers, counts = np.empty(int(num_bins)), np.empty(int(num_bins))
  _lib.check(_lib.lib().chm_binning1d(_lib.dptr(x), _lib.dptr(w), x.size, int(num_bins), _lib.dptr(centers), _lib.dptr(counts),
                                      _lib.default_device()))
  return centers, counts


def kde1d(dataset, grid, weights=None, kernel='epan', bw_method=None):
  """math.py:52-81: weighted 1-D KDE (Epanechnikov unless kernel != 'epan', then Gaussian) on ``grid``."""
  code, scalar = _bw(bw_method)
  x, g = _lib.as_f64(dataset).reshape(-1), _lib.as_f64(grid).reshape(-1)
  w = None if weights is None else _lib.as_f64(weights).reshape(-1)
  if w is not None and w.shape != x.shape:
    raise ValueError("kde1d: dataset and weights differ in length")
  out = np.empty_like(g)
  _lib.check(_lib.lib().chm_kde1d(_lib.dptr(x), _lib.dptr(w), x.size, _lib.dptr(g), g.size, 0 if kernel == 'epan' else 1, code, scalar,
                                  _lib.dptr(out), _lib.default_device()))
  return out.reshape(np.shape(grid))


def gkde_nd(dataset, evaluation_grid, weights=None, bw_method=None, in_log=False):
  """math.py:95-148 (``jax_gkde_nd``) / 154-229 (``numba_gkde_nd``): n-dimensional weighted Gaussian KDE with covariance
  whitening ("same as jax.scipy.stats.gaussian_kde").  ``dataset`` (d, N), ``evaluation_grid`` (d, M); d <= 4.
  ``in_log=True`` (math.py:223-226, the numba kernel): the logarithm of the density, accumulated as the reference does
  (``np.logaddexp`` over the dataset from -inf) -- finite where the density itself underflows."""
  dataset = np.atleast_2d(np.asarray(dataset, dtype=np.float64))
  d, n = dataset.shape
  points = np.atleast_2d(np.asarray(evaluation_grid, dtype=np.float64))
  dp, m = points.shape
  if dp != d:
    if dp == 1 and m == d:
      points = points.T
      m = points.shape[1]
    else:
      raise ValueError("points have dimension " + str(dp) + ", dataset has dimension " + str(d))
  w = None
  if weights is not None:
    w = np.asarray(weights, dtype=np.float64)
    if w.ndim != 1:
      raise ValueError("`weights` input should be one-dimensional.")
    if len(w) != n:
      raise ValueError("`weights` input should be of length n_dataset")
    w = _lib.as_f64(w)
  if bw_method == "scott" or bw_method is None:
    code, scalar = 0, 0.
  elif bw_method == "silverman":
    code, scalar = 1, 0.
  elif np.isscalar(bw_method) and not isinstance(bw_method, str):
    code, scalar = 2, float(bw_method)
  else:
    raise ValueError("`bw_method` should be 'scott', 'silverman', a scalar")
  ds, pts = _lib.as_f64(dataset), _lib.as_f64(points)
  out = np.empty(m)
  fn = _lib.lib().chm_gkde_nd_log if in_log else _lib.lib().chm_gkde_nd
  _lib.check(fn(_lib.dptr(ds), _lib.dptr(w), d, n, _lib.dptr(pts), m, code, scalar, _lib.dptr(out), _lib.default_device()))
  return out


jax_gkde_nd = gkde_nd
numba_gkde_nd = gkde_nd
