"""Phase timing of the sample-stationary 3-D KDE kernel k_full_kde_chain (diagnostic build: scripts/build_variant.sh phasefull -DCHM_PHASE_PROF): shader-clock
ticks between the phase marks of the kernel (every mark waits for the wave's outstanding memory and LDS operations first), summed over wave 0 of every 64th
block -- where a wave's life goes -- at the full-mode bench workload (bench.py --mode full --nbatch 4).
    CHIMERA_LIB=chimera_amd/lib/variants/libchimera_hip_phasefull.so python3 scripts/phase_full.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chimera_amd import _lib, synth
from tests import helpers as H

NAMES = ["the event's record + the pixel's constants", 'march (all chunks)', 'sums across the lanes (all chunks)', 'waiting for the other waves of the block',
         'integrand + trapezoid + block sum + store', "the owned samples' values arrive", 'their starting values (one exp each)']


def main():
  cfg, ev, inj = synth.make_config('C3', seed=20250926)
  like, _, _ = H.build_product(ev, inj, kind='full')
  like.set_option('groups', 1)
  L = _lib.lib()
  out = (C.c_double * 8)()
  nbatch = 4
  lams = [dict(H0=60. + 20. * i / max(nbatch - 1, 1)) for i in range(nbatch)]
  for _ in range(2):
    like.batch(lams)
  L.chm_debug_phase(out)
  for _ in range(3):
    like.batch(lams)
  L.chm_debug_phase(out)
  v = np.array(out[:])
  n = max(v[7], 1)
  tot = v[:7].sum()
  print(f"nbatch {nbatch}: {int(v[7])} sampled waves; ticks per wave and share: " +
        '; '.join(f"{NAMES[i]} {v[i] / n:.0f} ({100 * v[i] / tot:.1f} %)" for i in range(7)) + f"; total {tot / n:.0f} ticks per wave")


if __name__ == '__main__':
  main()
