"""Phase timing of k_marg_fused (diagnostic build -DCHM_PHASE_PROF, scripts/build_variant.sh phase -DCHM_PHASE_PROF): shader-clock cycles between the
phase marks, summed over the first wave of every 64th block.
    CHIMERA_LIB=chimera_amd/lib/variants/libchimera_hip_phase.so python3 scripts/phase_fused.py [nbatch ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chimera_amd import _lib, synth
from tests import helpers as H

NAMES = ['stage+zero', 'bounds', 'sample pass', 'merge+stats', 'per-z factors', 'pixel pass', 'pixel sum']


def main():
  cfg, ev, inj = synth.make_config('C3', seed=20250926)
  like, _, _ = H.build_product(ev, inj)
  L = _lib.lib()
  out = (C.c_double * 8)()
  like.set_option("fused", 2)
  for nbatch in [int(a) for a in sys.argv[1:]] or [1, 128]:
    lams = [dict(H0=60. + 20. * i / max(nbatch - 1, 1)) for i in range(nbatch)]
    for _ in range(3):
      like.batch(lams)
    L.chm_debug_phase(out)
    like.batch(lams)
    L.chm_debug_phase(out)
    v = np.array(out[:])
    n = max(v[7], 1)
    print(f"nbatch {nbatch}: {int(v[7])} sampled blocks; cycles per block (100 MHz clock x ?): " +
          ', '.join(f"{NAMES[i]} {v[i] / n:.0f}" for i in range(7)) + f"; total {v[:7].sum() / n:.0f}")


if __name__ == '__main__':
  main()
