"""Development check of the fused event kernel (chm_fused.h): the same calls with CHM_FUSED=0 (separate kernels) and CHM_FUSED=2 (fused for every
call size), against each other and against the oracle, on a handful of small shapes; then timing at C3 for 1 and 128 draws per call.
    python3 scripts/try_fused.py [--time]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as H


def both(like, lams):
  out = {}
  for mode in ('0', '2'):
    like.set_option('fused', int(mode))
    out[mode] = like._eval(like._params_array(lams), want=('log_like_evs',))
  return out


def check(name, E, S, P, Z, seed, ragged=True, models=None, draws=(60., 70., 85.), like_kw=None):
  cfg, ev, inj = H.small_config(E=E, S=S, P=P, Z=Z, I=1500, seed=seed, ragged=ragged)
  like_o, _, _ = H.build_oracle(ev, inj, models=models, like_kw=like_kw)
  like_p, _, _ = H.build_product(ev, inj, models=models, like_kw=like_kw)
  lams = [dict(H0=h) for h in draws]
  r = both(like_p, lams)
  a, f = r['0'], r['2']
  ref = np.array([like_o.compute_all(**l)[0] for l in lams])
  H.assert_loglike_close(f['log_like_evs'], ref, rtol=1e-9, atol=1e-9)
  fin = ~H.neginf_class(a['log_like_evs'])
  d = np.max(np.abs(a['log_like_evs'][fin] - f['log_like_evs'][fin])) if fin.any() else 0.
  dh = np.max(np.abs(a['log_hyper'] - f['log_hyper']))
  # scalar calls go through the few-draw instantiation
  like_p.set_option('fused', 1)
  sc = np.array([like_p(**l) for l in lams])
  ds = np.max(np.abs(sc - f['log_hyper']))
  print(f"{name}: E={E} S={S} P={P} Z={Z}: fused vs oracle ok; |fused - separate| per event {d:.2e}, log_hyper {dh:.2e}; scalar vs batch {ds:.2e}")
  assert d < 1e-11 and dh < 1e-10 * np.sqrt(E) and ds == 0.


def main():
  check('ragged', 6, 256, 4, 64, 7)
  check('odd-P', 5, 384, 5, 48, 3)
  check('one-pixel', 3, 128, 1, 32, 5)
  check('short', 4, 100, 3, 40, 11)
  check('many-pixels', 4, 2048, 32, 200, 2, ragged=False)
  check('long', 3, 6000, 16, 100, 9)
  check('bpl', 6, 512, 6, 64, 1, models=dict(mass='bpl'))
  check('tpl-mg', 6, 512, 6, 64, 4, models=dict(mass='tpl', cosmo='mg_flrw', cosmo_kw=dict(Xi0=1.6, n=1.5)))
  check('silverman', 5, 512, 8, 64, 8, like_kw=dict(bw_method='silverman'))
  print('small shapes ok')
  if '--time' not in sys.argv:
    return
  from chimera_amd import synth
  cfg, ev, inj = synth.make_config('C3', seed=20250926)
  like, _, _ = H.build_product(ev, inj)
  for nbatch in (1, 128):
    lams = [dict(H0=60. + 20. * i / max(nbatch - 1, 1)) for i in range(nbatch)]
    for mode in ('0', '2'):
      like.set_option('fused', int(mode))
      for _ in range(4):
        r = like.batch(lams)
      t0 = time.perf_counter()
      n = 40 if nbatch == 1 else 10
      for _ in range(n):
        r = like.batch(lams)
      dt = (time.perf_counter() - t0) / n
      print(f"C3 nbatch={nbatch} CHM_FUSED={mode}: {dt * 1e3:.3f} ms per call, log_hyper[0] = {r[0]:.12f}")


if __name__ == '__main__':
  main()
