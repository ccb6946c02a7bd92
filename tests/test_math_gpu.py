"""GPU parity of the stand-alone building blocks (chimera_amd/utils/math.py, the generic helpers of mass.py) against the oracle's
restatement of CHIMERA/utils/math.py and CHIMERA/population/mass.py:240-279 -- and, for the n-d KDE, against scipy directly."""
import numpy as np
import pytest

from oracle import chimera_oracle as O
import chimera_amd as CH
from chimera_amd.utils import math as M

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kernel', ['epan', 'gauss'])
@pytest.mark.parametrize('bw', [None, 'scott', 'silverman', 0.37])
@pytest.mark.parametrize('weighted', [True, False])
def test_kde1d(kernel, bw, weighted):
  rng = np.random.default_rng(5)
  x = np.concatenate([rng.normal(0.3, 0.05, 700), rng.normal(0.6, 0.02, 333)])
  w = rng.uniform(0., 2., x.size) if weighted else None
  g = np.linspace(0.0, 0.9, 517)
  np.testing.assert_allclose(M.kde1d(x, g, w, kernel=kernel, bw_method=bw), O.kde1d(x, g, w, kernel, bw), rtol=1e-11, atol=1e-13)
  with pytest.raises(ValueError):
    M.kde1d(x, g, w, bw_method='nope')


@pytest.mark.parametrize('B', [200, 7, 1100])
def test_binning1d_then_kde(B):
  rng = np.random.default_rng(6)
  x, w = rng.gamma(3., 0.1, 4096), rng.uniform(0., 1., 4096)
  c, n = M.binning1d(x, w, num_bins=B)
  co, no = O.binning1d(x, w, B)
  np.testing.assert_array_equal(c, co)
  np.testing.assert_allclose(n, no, rtol=1e-13, atol=0)
  assert abs(n.sum() - w.sum()) < 1e-10
  g = np.linspace(0., 1.5, 250)
  np.testing.assert_allclose(M.kde1d(c, g, n), O.kde1d(co, g, no), rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize('d', [1, 2, 3])
@pytest.mark.parametrize('bw', [None, 'silverman', 0.5])
def test_gkde_nd(d, bw):
  from scipy.stats import gaussian_kde
  rng = np.random.default_rng(7 + d)
  A = rng.normal(size=(d, d))
  data = A @ rng.normal(size=(d, 900)) + rng.normal(size=(d, 1))
  w = rng.uniform(0.1, 1., 900)
  pts = A @ rng.normal(size=(d, 301))
  got = M.gkde_nd(data, pts, weights=w, bw_method=bw)
  np.testing.assert_allclose(got, O.gkde_nd(data, pts, weights=w, bw_method=bw), rtol=1e-10)
  np.testing.assert_allclose(got, gaussian_kde(data, weights=w, bw_method=bw)(pts), rtol=1e-10)       # math.py:96 "same as gaussian_kde"
  np.testing.assert_allclose(M.numba_gkde_nd(data, pts), O.gkde_nd(data, pts), rtol=1e-10)             # unweighted
  with pytest.raises(ValueError):
    M.jax_gkde_nd(data, np.zeros((d + 1, 5)))
  # in_log=True (math.py:223-226): log of the density where it is representable, and finite far out in the tail where the density
  # itself underflows -- against scipy's logsumexp of the same terms (and gaussian_kde.logpdf)
  from scipy.special import logsumexp
  near = data[:, ::3] + 0.2 * (A @ rng.normal(size=(d, 300)))                                        # points inside the cloud (in its own geometry)
  dens = M.gkde_nd(data, near, weights=w, bw_method=bw)
  assert np.all(dens > 1e-30)
  np.testing.assert_allclose(M.gkde_nd(data, near, weights=w, bw_method=bw, in_log=True), np.log(dens), rtol=0, atol=1e-10)
  assert np.all(np.isfinite(M.gkde_nd(data, pts, weights=w, bw_method=bw, in_log=True)))
  far = near[:, :7] + 60. * np.std(data, axis=1, keepdims=True)
  lf = M.numba_gkde_nd(data, far, weights=w, bw_method=bw, in_log=True)
  assert np.all(np.isfinite(lf)) and np.all(lf < -700.) and np.all(M.gkde_nd(data, far, weights=w, bw_method=bw) == 0.)
  kde = gaussian_kde(data, weights=w, bw_method=bw)
  np.testing.assert_allclose(lf, kde.logpdf(far), rtol=1e-10)
  W = w / w.sum()
  Lc = np.linalg.cholesky(np.linalg.inv(np.atleast_2d(kde.covariance)))
  dw, fw = data.T @ Lc, far.T @ Lc
  terms = np.log(W)[None, :] + (np.sum(np.log(np.diag(Lc))) - 0.5 * d * np.log(2 * np.pi)) - 0.5 * np.sum((dw[None, :, :] - fw[:, None, :])**2, axis=2)
  np.testing.assert_allclose(lf, logsumexp(terms, axis=1), rtol=1e-10)


def test_trapz_and_cumtrapz():
  rng = np.random.default_rng(8)
  x = np.sort(rng.uniform(0., 3., 1500)); y = np.sin(x) + 2.
  np.testing.assert_allclose(M.trapz(y, x), O.trapz(y, x), rtol=1e-14)
  np.testing.assert_allclose(M.cumtrapz(y, x), O.cumtrapz(y, x), rtol=1e-13, atol=1e-15)
  Y = rng.uniform(size=(6, 5, 333)); X = np.sort(rng.uniform(size=(6, 1, 333)), axis=-1)
  np.testing.assert_allclose(M.trapz(Y, X, axis=-1), O.trapz(Y, X, axis=-1), rtol=1e-13)
  np.testing.assert_allclose(M.trapz(Y[0], X[0, 0]), O.trapz(Y[0], X[0, 0]), rtol=1e-13)
  assert M.cumtrapz(y, x)[0] == 0.


def test_generic_mass_helpers():
  m = np.concatenate([np.linspace(1., 120., 400), [5.1, 87.]])
  for alpha in (-2.3, 1.1, 0.):
    np.testing.assert_allclose(CH.mass.tpl_notnorm(m, alpha, 5.1, 87.), O.tpl_notnorm(m, alpha, 5.1, 87.), rtol=1e-12)
    np.testing.assert_allclose(CH.mass.tpl_cdf(alpha, 5.1, m), O.tpl_cdf(alpha, 5.1, m), rtol=1e-12, atol=1e-13)
  np.testing.assert_allclose(CH.mass.tpl_cdf(-1., 5.1, m), O.tpl_cdf(-1., 5.1, m), rtol=1e-12, atol=1e-14)
  np.testing.assert_allclose(CH.mass.gaussian(m, 34., 3.6), O.gaussian(m, 34., 3.6), rtol=1e-12, atol=1e-300)
  np.testing.assert_allclose(CH.mass.truncated_gaussian(m, 34., 3.6, 5.1, 52.), O.truncated_gaussian(m, 34., 3.6, 5.1, 52.), rtol=1e-12, atol=1e-300)


def test_localization_volumes():
  """data.compute_localization_volumes (CHIMERA/data.py:452-484: the body the reference spells out with undefined names) -- the sky area in
  steradians times the comoving shell between the 5 % distance under one cosmology and the 95 % distance under another, per unit solid angle --
  against the same expression on the oracle's z_from_dGW / Vc_at_z."""
  import chimera_amd as CH
  from chimera_amd import data as D
  rng = np.random.default_rng(11)
  E, S = 4, 3000
  theta = rng.normal(1.1, 0.03, size=(E, S)) + rng.normal(0., 0.2, size=(E, 1))
  phi = rng.normal(2.0, 0.05, size=(E, S)) + 0.4 * (theta - theta.mean(axis=1, keepdims=True))
  dL = rng.lognormal(np.log([[0.8], [1.5], [3.0], [6.0]]), 0.2, size=(E, S))
  kw_lo, kw_hi = dict(H0=60., Om0=0.3, z_max=5.), dict(H0=80., Om0=0.28, z_max=5.)
  got = D.compute_localization_volumes(theta, phi, dL, CH.cosmo.flrw(**kw_lo), CH.cosmo.flrw(**kw_hi), percentile=90)
  ster = D.compute_localization_areas(theta, phi, 90) / (180. / np.pi)**2
  d5, d95 = np.percentile(dL, 5., axis=1), np.percentile(dL, 95., axis=1)
  olo, ohi = O.flrw(**kw_lo), O.flrw(**kw_hi)
  want = ster * (O.Vc_at_z(ohi, O.z_from_dGW(ohi, d95)) - O.Vc_at_z(olo, O.z_from_dGW(olo, d5))) / (4. * np.pi)
  np.testing.assert_allclose(got, want, rtol=1e-10)
  assert got.shape == (E,) and np.all(got > 0.) and np.all(np.diff(got) > 0.)       # farther events: larger shells
