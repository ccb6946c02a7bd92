"""One-off wider fuzz: the random-configuration and random-draw parity tests of tests/test_gpu_parity.py over many more seeds."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
  try:
    T.test_random_configurations_against_the_numpy_oracle(seed)
  except AssertionError as e:
    bad += 1
    print('SEED', seed, 'FAILED:', str(e)[:1500], flush=True)
  except Exception:
    bad += 1
    print('SEED', seed, 'ERROR'); traceback.print_exc()
print('done', hi - lo, 'seeds,', bad, 'failures')
