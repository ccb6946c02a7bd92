#!/usr/bin/env python3
"""Measured ceilings of the two hot kernels (run on the GPU box with the -DCHM_PROBE build):

  CHIMERA_LIB=chimera_amd/lib/variants/libchimera_hip_probe.so python3 scripts/run_probes.py [--events 4] [--draws 4] [--seconds 1.2] [--out FILE.json]
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d DIR -- python3 scripts/run_probes.py ...      (instruction counts of the same launches)

A small C3-shaped workload (E events x 32 pixels x 1000 z-bins x 4096 samples) is evaluated once for `draws` draws -- that fills the (z, w) workspaces, the
per-z factors and the event statistics -- and then the PRODUCTION BODIES of k_kde_marg_sub2<32, 4, 200, false> and k_samples_fast<2, false, false> are
replayed over that cache-resident data by scripts/gw_loop_probe.hip / scripts/sample_body_probe.hip for >= `seconds` each, in launches of ~0.15 s.  Printed /
stored: pairs of pixels per second and samples per second the bodies sustain without HBM round trips -- the ceiling bench.py's
roofline.frac_of_sustained is measured against (profiles/r06/probe_ceilings.json)."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--events', type=int, default=4)
  ap.add_argument('--draws', type=int, default=4)
  ap.add_argument('--seconds', type=float, default=1.2)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  import chimera_amd as CH
  from chimera_amd import synth, _lib
  from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
  L = _lib.lib()
  if not hasattr(L, 'chm_debug_probe'):
    raise SystemExit('run_probes.py: the loaded library has no chm_debug_probe (build it with scripts/build_variant.sh probe -DCHM_PROBE and select it with CHIMERA_LIB)')
  L.chm_debug_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
  L.chm_debug_probe.restype = C.c_int
  cfg, ev, inj = synth.make_config('C3', E=args.events, I=4000)
  E, S, P, Z = cfg['E'], cfg['S'], cfg['P'], cfg['Z']
  pe_fields = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')
  th = CH.data.theta_pe_det(**{k: ev[k] for k in pe_fields})
  gal_cat = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
  pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gal_cat, scale_free=True)
  sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), N_inj=inj['N_inj'], N_eff=5.)
  like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', kernel='epan', bw_method=None, cut_grid=2, binning=True, num_bins=200)
  nb = args.draws
  # 16 draws: the batched launch sequence (k_event_stats, ranged per-z factors); the probes use the first `nb` of them
  v = like.batch(dict(H0=np.linspace(62., 78., max(16, nb))))
  h = like._handle()
  out = {"workload": {"E": E, "P": P, "Z": Z, "S": S, "draws": nb, "resident_MB": (nb * E * S * 16 + E * P * Z * 8 + E * S * 48 + nb * E * Z * 16) / 1e6},
         "log_hyper_check": float(v[0])}
  for which, name, per_call in ((0, 'k_kde_marg_sub2<32, 4, 200, false>', 4.), (1, 'k_samples_fast<2, false, false>', float(E * S))):
    nblocks = 256 * 16 * 4 if which == 0 else 256 * 4 * 2
    ms = (C.c_double * 64)()
    reps = 40 if which == 0 else 4
    _lib.check(L.chm_debug_probe(h, which, nb, nblocks, reps, 1, ms))               # calibration launch (also the first touch of the data)
    reps = min(65535, max(1, int(reps * 150. / ms[0])))                                          # ~0.15 s per launch
    nl = min(64, max(4, int(np.ceil(args.seconds / 0.15))))
    _lib.check(L.chm_debug_probe(h, which, nb, nblocks, reps, nl, ms))
    t = np.array(ms[:nl])
    units = nblocks * reps * per_call
    tail = t[nl // 2:]                                                               # the second half: the clock has settled
    rate = units / (np.median(tail) * 1e-3)
    out[name] = {"unit": "pairs of pixels" if which == 0 else "samples", "units_per_launch": units, "blocks": nblocks * reps, "body_calls_per_block": 1,
                 "launches": nl, "launch_ms": [float(x) for x in t], "launch_ms_median_second_half": float(np.median(tail)),
                 "units_per_s": rate, "total_s": float(t.sum() * 1e-3)}
    print(f"{name}: {nl} launches x {np.median(tail):.1f} ms, {rate / 1e6:.1f} M {out[name]['unit']}/s sustained ({t.sum() * 1e-3:.2f} s in all)", flush=True)
  # what the SAME small workload gives through the production path would be launch-bound; the reference points are the C3 numbers of bench.py:
  out["production_reference"] = {"note": "C3, 128 draws per call: 2.048e6 pairs of pixels per k_kde_marg_sub2 launch, 5.243e8 samples per k_samples_fast launch"}
  like2 = float(like(H0=70.))
  out["log_hyper_after"] = like2
  if args.out:
    with open(os.path.join(ROOT, args.out), 'w') as f:
      json.dump(out, f, indent=1)
  print(json.dumps(out))
  like.close(); sel.close()


if __name__ == '__main__':
  main()
