"""Sampler glue (reference: CHIMERA/utils/emcee_utils.py:54-64, 281-288): turn sampler positions into the keyword
arguments of ``hyperlikelihood`` and evaluate an ensemble of walkers in one batched call (``emcee`` ``vectorize=True``).
The emcee driver itself (chain files, restart, moves) is outside the accelerated path."""
import numpy as np


def generate_dict(params, params_keys, to_calc=None):
  """emcee_utils.py:54-64: (nwalkers, ndim) or (ndim,) positions -> dict of arrays / scalars."""
  params = np.asarray(params)
  if params.ndim > 1:
    if to_calc is None:
      return {k: params[:, i] for i, k in enumerate(params_keys)}
    return {k: params[to_calc, i] for i, k in enumerate(params_keys)}
  return {k: params[i] for i, k in enumerate(params_keys)}


def make_log_prob(like, params_keys, priors=None):
  """log-posterior for ``emcee.EnsembleSampler(..., vectorize=True)``: flat priors inside ``priors`` (ndim, 2), -inf
  outside; only the walkers inside the prior are evaluated (the reference's ``to_calc`` mask)."""
  priors = None if priors is None else np.asarray(priors, dtype=np.float64)

  def log_prob(theta):
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    ok = np.ones(len(theta), dtype=bool) if priors is None else np.all((theta >= priors[:, 0]) & (theta <= priors[:, 1]), axis=1)
    out = np.full(len(theta), -np.inf)
    if ok.any():
      lp = np.atleast_1d(like(**generate_dict(theta, params_keys, to_calc=ok)))
      out[ok] = np.where(np.isnan(lp), -np.inf, lp)
    return out
  return log_prob
