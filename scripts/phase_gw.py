"""Phase timing of the standard GW kernel k_kde_marg_sub2 (diagnostic build: scripts/build_variant.sh phasegw -DCHM_PHASE_PROF): shader-clock ticks between
the phase marks of kde_sub_item (every mark waits for the wave's outstanding memory and LDS operations first), summed over lane 0 of the waves of draws 0 and 64
of every 16th event -- where a wave's life goes.
    CHIMERA_LIB=chimera_amd/lib/variants/libchimera_hip_phasegw.so python3 scripts/phase_gw.py [nbatch ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chimera_amd import _lib, synth
from tests import helpers as H

NAMES = ['samples arrive (exposed load latency)', 'max z + zeroing + histogram (+ wait for the first pass)', 'bin sums + scans + prefix stores', 'bandwidth + constants',
         'grid loop', 'final scans + stores']


def main():
  cfg, ev, inj = synth.make_config('C3', seed=20250926)
  like, _, _ = H.build_product(ev, inj)
  like.set_option('groups', 1)
  L = _lib.lib()
  out = (C.c_double * 8)()
  for nbatch in [int(a) for a in sys.argv[1:]] or [128, 1]:
    lams = [dict(H0=60. + 20. * i / max(nbatch - 1, 1)) for i in range(nbatch)]
    for _ in range(3):
      like.batch(lams)
    L.chm_debug_phase(out)
    for _ in range(4):
      like.batch(lams)
    L.chm_debug_phase(out)
    v = np.array(out[:])
    n = max(v[7], 1)
    tot = v[:6].sum()
    print(f"nbatch {nbatch}: {int(v[7])} sampled waves (4 items each at nbatch > 8); ticks per wave and share: " +
          '; '.join(f"{NAMES[i]} {v[i] / n:.0f} ({100 * v[i] / tot:.1f} %)" for i in range(6)) + f"; total {tot / n:.0f} ticks per wave")


if __name__ == '__main__':
  main()
