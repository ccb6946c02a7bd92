"""One-off: the smallest shapes (S, Z, P, E, I, num_bins down to 1-4), HIP vs the NumPy oracle, all modes."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import helpers as H

bad = tot = 0
for kind in ('marginalized', 'approximate', 'full', None):
  pixelated = kind is not None
  for S, Z, P, nbins in itertools.product((2, 3, 5, 64), (4, 5, 9), (1, 2), (1, 2, 7)):
    if kind == 'full' and nbins != 1:
      continue
    for E, I in ((1, 1), (2, 5)):
      cfg, ev, inj = H.small_config(E=E, S=S, P=P, Z=Z, I=I, seed=S * 100 + Z * 10 + P, ragged=False, pixelated=pixelated)
      like_kw = {} if kind == 'full' else dict(num_bins=nbins)
      tot += 1
      try:
        like_o, _, _ = H.build_oracle(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw)
        like_p, _, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind, like_kw=like_kw)
        with np.errstate(all='ignore'):
          try:
            ro = like_o.compute_all(H0=70.)
          except np.linalg.LinAlgError:
            like_p.compute_all(H0=70.)          # must not crash; the oracle's Cholesky raises on degenerate covariances
            continue
          rp = like_p.compute_all(H0=70.)
        H.assert_loglike_close(rp[0], ro[0], rtol=1e-9, atol=1e-9)
        if np.isfinite(ro[2]):
          np.testing.assert_allclose(rp[2], ro[2], rtol=1e-10)
        like_p.close()
      except AssertionError as e:
        bad += 1
        print(f"MISMATCH kind={kind} S={S} Z={Z} P={P} bins={nbins} E={E} I={I}: {str(e)[:300]}", flush=True)
print('done;', bad, 'mismatches of', tot)
