#!/usr/bin/env python3
"""CPU: the generator of scripts/fuzz_parity.py with the C restatement (oracle/chimera_oracle_c.c through oracle/oracle_c.py) in the place of the
product -- the two restatements of the reference algorithm against each other on random configurations with hostile inputs and extreme
hyper-parameters (check (1) of fuzz_parity only):  python3 scripts/fuzz_oracles_cpu.py N seed0 [hostile share]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts')); sys.path.insert(0, ROOT)
import numpy as np
from tests import helpers as H
from oracle import oracle_c as OC
import fuzz_parity as F
F.HOSTILE_SHARE, F.EXTREME_SHARE = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6, 0.3
class Skip(Exception): pass
cap = {}
class CLike:
  """stands in for the product: the C restatement evaluated through oracle_c on the oracle object captured at build time"""
  def __init__(self, like_o): self.o = like_o
  def compute_all(self, **lam): return OC.compute_all(self.o, lam, nthreads=8)
  def __call__(self, **lam): return 0.
  def batch(self, lams): return np.zeros(len(lams))
  def set_option(self, *a): raise AssertionError('skip fused')
  def close(self): pass
orig_o = H.build_oracle
def bo(ev, inj, **kw):
  r = orig_o(ev, inj, **kw); cap['o'] = r[0]; return r
H.build_oracle = bo
H.build_product = lambda ev, inj, **kw: (CLike(cap['o']), None, CLike(cap['o']))
n, seed0 = int(sys.argv[1]), int(sys.argv[2])
fails, by = 0, {}
t0 = time.time()
for i in range(n):
  try:
    ok, desc, checks = F.one(np.random.default_rng(77000 + seed0 + i))
  except Exception as ex:
    ok, desc = False, f'EXC {type(ex).__name__}: {ex}'
  if not ok and 'skip fused' not in desc:
    fails += 1
    k = desc.split(',')[0][:24] if desc.startswith('HOSTILE') else desc[:12]
    by[k] = by.get(k, 0) + 1
    if by[k] <= 1: print('FAIL', seed0 + i, desc.split(' kind=')[0][:40], '|', ' '.join(desc.split('\n')[1:6])[:400], flush=True)
print('C oracle vs NumPy oracle:', n, 'configurations', fails, 'failures', by, round(time.time() - t0), 's')
