"""Oracle-free checks of the hyper-likelihood path: exact relations any correct implementation of the reference's formulas satisfies,
run on the CPU restatements at small size (`-m "not gpu"`) and on the HIP path at the full size of BASELINE.json's headline
configuration C3 (`-m gpu`).  They need no reference output, so they hold whatever one reads into the reference's lines:

  distance scaling   H0 -> H0/a with every luminosity distance -> a dL (events and injections): the redshifts, source masses and
                     weights do not change, dVc/dz and fR = Vc(z1) - Vc(z0) grow by a^3, ddL/dz by a  =>  every L_i and N_exp grow by a^2
                     (likelihood.py:262-281, pop_wrapper.py:102-111, completeness.py:54-67) and the scale-free log-hyperlikelihood
                     (likelihood.py:313-316) does not move.  a = 2 keeps every floating-point operation exact up to the final logs.
  prior scaling      pe_prior -> c pe_prior divides every L_i by c (pop_wrapper.py:79); p_draw -> c p_draw divides N_exp by c
                     (selection_function.py:38).
  permutation        reordering the events reorders the per-event likelihoods and changes nothing else (likelihood.py:296-300).
"""
import numpy as np
import pytest

from tests import helpers as H

LN4 = 2. * np.log(2.)


def _scaled(ev, inj, a):
  ev2, inj2 = dict(ev), dict(inj)
  ev2['dL'] = ev['dL'] * a
  inj2['dL'] = inj['dL'] * a
  return ev2, inj2


def _perm(ev, order):
  out = dict(ev)
  E = len(ev['dL'])
  for k, v in ev.items():
    if isinstance(v, np.ndarray) and v.shape[:1] == (E,):
      out[k] = v[order]
  return out


def _relations(build, ev, inj, lam, models, kind, pixelated, tol):
  """build(ev, inj, ...) -> likelihood with compute_all; returns nothing, asserts the relations."""
  kw = dict(pixelated=pixelated, kind=kind, models=models)
  base = build(ev, inj, **kw)[0].compute_all(**lam)
  fin = ~H.neginf_class(base[0])
  assert fin.sum() >= max(1, len(fin) // 2)
  # distance scaling, a = 2
  ev2, inj2 = _scaled(ev, inj, 2.)
  lam2 = dict(lam, H0=lam['H0'] / 2.)
  r = build(ev2, inj2, **kw)[0].compute_all(**lam2)
  assert np.array_equal(H.neginf_class(r[0]), ~fin)
  np.testing.assert_allclose(r[0][fin] - base[0][fin], LN4, rtol=0, atol=tol)
  np.testing.assert_allclose(r[2] - base[2], LN4, rtol=0, atol=tol)                   # log N_exp
  np.testing.assert_allclose(r[3], base[3], rtol=0, atol=tol * len(fin))              # scale-free log-hyperlikelihood: unchanged
  # prior scaling
  ev3 = dict(ev); ev3['pe_prior'] = ev['pe_prior'] * 2.
  inj3 = dict(inj); inj3['p_draw'] = inj['p_draw'] * 4.
  r = build(ev3, inj3, **kw)[0].compute_all(**lam)
  np.testing.assert_allclose(r[0][fin] - base[0][fin], -np.log(2.), rtol=0, atol=tol)
  np.testing.assert_allclose(r[2] - base[2], -np.log(4.), rtol=0, atol=tol)
  # permutation of the events
  order = np.random.default_rng(5).permutation(len(fin))
  r = build(_perm(ev, order), inj, **kw)[0].compute_all(**lam)
  a, b = r[0], base[0][order]
  assert np.array_equal(H.neginf_class(a), H.neginf_class(b))
  np.testing.assert_allclose(a[fin[order]], b[fin[order]], rtol=0, atol=tol)
  np.testing.assert_allclose(r[2], base[2], rtol=0, atol=1e-13)
  np.testing.assert_allclose(r[3], base[3], rtol=0, atol=tol * len(fin))


CASES = [
  ('marginalized', True, dict(), dict(H0=70.)),
  ('approximate', True, dict(), dict(H0=64., alpha=3.1)),
  (None, False, dict(), dict(H0=76.)),
  ('marginalized', True, dict(cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.7, n=1.9)), dict(H0=70., Xi0=2.2)),
  ('marginalized', True, dict(mass='bpl', rate='trunc_madau_dickinson'), dict(H0=70.)),
  ('full', True, dict(), dict(H0=70.)),
]


@pytest.mark.parametrize('kind,pixelated,models,lam', CASES)
def test_exact_relations_hold_in_the_numpy_restatement(kind, pixelated, models, lam):
  cfg, ev, inj = H.small_config(E=5, S=192, P=3, Z=48, I=1200, seed=11, pixelated=pixelated)
  with np.errstate(all='ignore'):
    _relations(H.build_oracle, ev, inj, lam, models, kind, pixelated, tol=2e-12)


def test_exact_relations_hold_in_the_c_restatement():
  from oracle import oracle_c as OC
  cfg, ev, inj = H.small_config(E=24, S=512, P=4, Z=96, I=4000, seed=12)

  class _C:                                                 # compute_all of the C restatement behind the builder interface
    def __init__(self, like): self.like = like
    def compute_all(self, **lam): return OC.compute_all(self.like, lam, nthreads=4)

  def build(ev_, inj_, **kw):
    return (_C(H.build_oracle(ev_, inj_, **kw)[0]),)
  for kind in ('marginalized', 'approximate', 'full'):
    _relations(build, ev, inj, dict(H0=70.), dict(), kind, True, tol=2e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('kind,pixelated,models,lam', CASES)
def test_exact_relations_hold_on_the_gpu(kind, pixelated, models, lam):
  cfg, ev, inj = H.small_config(E=12, S=1024, P=4, Z=128, I=5000, seed=13, pixelated=pixelated)
  _relations(H.build_product, ev, inj, lam, models, kind, pixelated, tol=2e-12)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_exact_relations_hold_on_the_gpu_at_the_full_size_of_c3():
  """1000 events x 32 pixels x 1000 z-bins x 4096 samples, 1e5 injections: the distance scaling with a = 2 moves every one of the
  1000 per-event log-likelihoods by 2 ln 2 and leaves the log-hyperlikelihood where it was; the batched call sees the same."""
  import gc
  from chimera_amd import synth
  cfg, ev, inj = synth.make_config('C3')
  assert (cfg['E'], cfg['P'], cfg['Z'], cfg['S']) == (1000, 32, 1000, 4096)
  lams = [dict(H0=67.), dict(H0=88., lambda_peak=0.08, gamma=2.0)]
  like = H.build_product(ev, inj)[0]
  base = [like.compute_all(**l) for l in lams]
  like.close(); del like; gc.collect()
  ev2, inj2 = _scaled(ev, inj, 2.)
  like2 = H.build_product(ev2, inj2)[0]
  lams2 = [dict(l, H0=l['H0'] / 2.) for l in lams]
  for b, l2 in zip(base, lams2):
    r = like2.compute_all(**l2)
    fin = ~H.neginf_class(b[0])
    assert fin.mean() > 0.9 and np.array_equal(H.neginf_class(r[0]), ~fin)
    np.testing.assert_allclose(r[0][fin] - b[0][fin], LN4, rtol=0, atol=2e-12)
    np.testing.assert_allclose(r[2] - b[2], LN4, rtol=0, atol=2e-12)
    np.testing.assert_allclose(r[3], b[3], rtol=0, atol=2e-9)
  np.testing.assert_allclose(like2.batch(lams2), [b[3] for b in base], rtol=0, atol=2e-9)
  like2.close()
