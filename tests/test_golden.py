"""Golden vectors (tests/golden/*.npz): the oracle must keep reproducing them (CPU), the HIP path must match them (GPU)."""
import numpy as np
import pytest

from tests import helpers as H
from tests.golden.make_golden import CASES, LAMBDAS


def _valid(go, gp, neff):
  if go.ndim == 3:
    valid = np.arange(go.shape[1])[None, :] < np.asarray(neff)[:, None]
    return go[valid], gp[valid]
  return go, gp


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_reproduces_golden(name):
  pixelated, kind, models, like_kw = CASES[name]
  ev, inj, exp = H.load_golden(name)
  like, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind or 'marginalized', models=models, like_kw=like_kw)
  for i, lam in enumerate(LAMBDAS):
    r = like.compute_all(**lam)
    H.assert_loglike_close(r[0], exp['log_like_evs'][i], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r[3], exp['log_hyper'][i], rtol=1e-12)
    np.testing.assert_allclose(r[2], exp['log_Nexp'][i], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_hip_matches_golden(name):
  pixelated, kind, models, like_kw = CASES[name]
  ev, inj, exp = H.load_golden(name)
  like, pop, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind or 'marginalized', models=models, like_kw=like_kw)
  E = len(ev['dL'])
  for i, lam in enumerate(LAMBDAS):
    r = like.compute_all(**lam)
    H.assert_loglike_close(r[0], exp['log_like_evs'][i], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r[1], exp['log_num'][i], rtol=0, atol=1e-7 * np.sqrt(E))
    np.testing.assert_allclose(r[2], exp['log_Nexp'][i], rtol=1e-10)
    np.testing.assert_allclose(r[3], exp['log_hyper'][i], rtol=0, atol=1e-7 * np.sqrt(E))
  p0 = like.population.update(**LAMBDAS[0])
  gp = like.p_gw3d(p0) if pixelated else like.p_gw1d(p0)
  go, gp = _valid(exp['p_gw'], gp, ev.get('neff_pixels'))
  fin = np.isfinite(go)
  np.testing.assert_allclose(gp[fin], go[fin], rtol=1e-9, atol=1e-9 * np.max(np.abs(go[fin])))
